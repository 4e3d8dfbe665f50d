#!/bin/bash
# kernel trace of the default arrangement and its summary by tools/timeline_summary.py: tools/timeline_run.sh [bench args]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/timeline
rm -rf $O && mkdir -p $O
cd $R
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/t -o t --output-format csv -- python3 bench.py --only-main --steps 21 --warmup 14 "$@" > $O/run.log 2>&1
f=$(find $O/t -name "*kernel_trace.csv" | head -1)
head -2 $f > $O/trace_head.csv
python3 tools/timeline_summary.py $f > $O/summary.txt
rm -rf $O/t
grep '^{' $O/run.log | cut -c1-200
cat $O/summary.txt
