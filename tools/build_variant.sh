#!/bin/bash
# A/B builds: tools/build_variant.sh <name> [-DFLAG=1 ...]  ->  tools/ab/<name>/libmwhip.so (select with MW_LIB_PATH)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/tools/ab/$name
mkdir -p $out
cd $root/manual-whisper_amd/csrc
pids=()
for f in engine kernels_mel kernels_gemm kernels_enc kernels_dec kernels_vad kernels_w2v aligner; do
  extra=""; [ $f = kernels_enc ] && extra="-fno-honor-nans"       # as manual_whisper_amd/build.py
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $extra "$@" -c $f.hip -o $out/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libmwhip.so $out/*.o
rm -f $out/*.o
echo $out/libmwhip.so
