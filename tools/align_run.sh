#!/bin/bash
# alignment: GPU tests, then align() throughput by pack size and passes in flight; log under gpurun_out/
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_align.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/pytest_align.log 2>&1 || { tail -30 gpurun_out/pytest_align.log; exit 1; }
tail -3 gpurun_out/pytest_align.log
{
for cfg in "30 8" "240 2" "240 3" "240 4" "120 4" "360 3"; do
  set -- $cfg
  echo "== pack $1 s, $2 passes in flight"; MW_ALIGN_PACK=$1 MW_ALIGN_STREAMS=$2 timeout -k 10 300 python tools/align_bench.py
done
} > gpurun_out/align_bench.log 2>&1
cat gpurun_out/align_bench.log
