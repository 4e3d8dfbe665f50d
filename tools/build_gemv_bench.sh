#!/bin/bash
# builds tools/gemv_bench from the library's kernel sources (decode projections + the LayerNorm they call)
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Imanual-whisper_amd/csrc "$@" tools/gemv_bench.hip manual-whisper_amd/csrc/kernels_dec.hip manual-whisper_amd/csrc/kernels_enc.hip -o tools/gemv_bench
