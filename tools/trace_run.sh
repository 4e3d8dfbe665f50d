# kernel trace of the default bench arrangement -> class overlap summary (run on the GPU box from the repo root)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace
rm -rf $O && mkdir -p $O
cd $R
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/t -o t --output-format csv -- python3 bench.py --no-cpu-baseline --no-fp8-b32 --no-ungrouped --no-align --steps 14 --warmup 7 "$@" > $O/bench.log 2>&1
F=$(find $O/t -name "*kernel_trace.csv" | head -1)
python3 tools/overlap_classes.py $F > $O/classes.txt
rm -rf $O/t
cat $O/classes.txt
grep -o "\"ms_per_step\": [0-9.]*" $O/bench.log || true
