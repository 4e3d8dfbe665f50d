#!/bin/bash
# the six decode projections of large-v3 at B sequences (default 128) through tools/gemv_bench
B=${1:-128}
for cfg in "3840 1280 0" "1280 1280 1" "1280 1280 0" "5120 1280 0" "1280 5120 1"; do
  timeout -k 5 60 tools/gemv_bench $cfg $B || exit 1
done
