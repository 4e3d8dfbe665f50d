#!/bin/bash
# tools/coreside_bench with the CU-sharing tile kernel (three-buffer ring); logs under gpurun_out/
set -e
mkdir -p gpurun_out
B=tools/coreside_bench
{
echo "== check"; MW_GEMM_CS=1 timeout -k 10 120 tools/gemm_bench 24000 5120 1280 50; MW_GEMM_CS=1 timeout -k 10 120 tools/gemm_bench 24000 1280 5120 50 resid; MW_GEMM_CS=1 timeout -k 10 120 tools/gemm_bench 24000 3840 1280 50 qkv
echo "== 256x128 4-wave ring-3 kernel N=5120"; MW_GEMM_CS=1 timeout -k 10 120 $B 24000 5120 1280
echo "== same, stream LDS 6000 (2 stream WG beside a product WG)"; MW_GEMM_CS=1 MW_XSTREAM_LDS=6000 timeout -k 10 120 $B 24000 5120 1280
echo "== same, stream LDS 12000 (1 stream WG beside a product WG)"; MW_GEMM_CS=1 MW_XSTREAM_LDS=12000 timeout -k 10 120 $B 24000 5120 1280
echo "== ring-3 kernel N=1280 K=5120 resid"; MW_GEMM_CS=1 timeout -k 10 120 $B 24000 1280 5120 45 100 resid
} > gpurun_out/coreside_d.log 2>&1
