#!/bin/bash
# kernel stats of one engine at 128 chunks per pass: tools/prof_b128.sh <tag> [env assignments are inherited]
set -e
tag=$1
R=$PWD
O=$R/gpurun_out/prof_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/t -o t --output-format csv -- python3 bench.py --no-cpu-baseline --also-batch 0 --no-fp8-b32 --no-ungrouped --no-align --group 1 --engines 1 --batch 128 --steps 2 --warmup 1 > $O/run.log 2>&1
cp $(find $O/t -name "*_kernel_stats.csv") $O/kernel_stats.csv
rm -rf $O/t
python3 - $O/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>7s} avg {float(r["AverageNs"])/1e3:8.2f} us {100*float(r["TotalDurationNs"])/tot:5.1f}%')
PY
tail -c 300 $O/run.log
