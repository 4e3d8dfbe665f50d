#!/bin/bash
# The four products of an encoder layer at 16 chunks (M = 24000), each with its own epilogue, and the cross K/V shape.
set -e
B=${1:-tools/gemm_bench}
timeout -k 5 60 $B 24000 3840 1280 30 qkv
timeout -k 5 60 $B 24000 1280 1280 30 resid
timeout -k 5 60 $B 24000 5120 1280 30 gelu
timeout -k 5 60 $B 24000 1280 5120 30 resid
timeout -k 5 60 $B 24000 3840 1280 30
timeout -k 5 60 $B 24000 5120 1280 30
timeout -k 5 60 $B 24000 3840 1280 30 f8
timeout -k 5 60 $B 24000 2560 1280 30
timeout -k 5 60 $B 3000 5120 1280 30 gelu
