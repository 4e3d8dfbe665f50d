#!/bin/bash
# This build's encoder products (with and without their epilogues) and the vendor BLAS on the same shapes, same data
# distribution, same box, sustained (400 launches each): gpurun_out/gemm_vs_blas.log
set -e
mkdir -p gpurun_out
B=tools/gemm_bench
{
echo "== vendor BLAS (torch.matmul), no epilogue"; timeout -k 10 300 python tools/blas_probe.py 400
echo "== this build, plain fp16 output"
for s in "3840 1280" "5120 1280" "1280 5120" "1280 1280" "2560 1280"; do timeout -k 5 60 $B 24000 $s 400 | tail -1; done
echo "== this build, the epilogues the encoder uses"
timeout -k 5 60 $B 24000 3840 1280 400 qkv | tail -1
timeout -k 5 60 $B 24000 1280 1280 400 resid | tail -1
timeout -k 5 60 $B 24000 5120 1280 400 gelu | tail -1
timeout -k 5 60 $B 24000 1280 5120 400 resid | tail -1
} > gpurun_out/gemm_vs_blas.log 2>&1
