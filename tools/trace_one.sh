#!/bin/bash
# kernel trace of one command summarised per (kernel, grid) by tools/trace_by_grid.py: tools/trace_one.sh <tag> <program> [args...]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/trace_one_$tag
rm -rf $O && mkdir -p $O
cd $R
timeout -k 10 400 rocprofv3 --kernel-trace -d $O -o t --output-format csv -- "$@" > $R/gpurun_out/trace_$tag.log 2>&1
python3 tools/trace_by_grid.py $(find $O -name "*kernel_trace.csv" | head -1) > $R/gpurun_out/trace_$tag.txt
rm -rf $O
cat $R/gpurun_out/trace_$tag.txt
