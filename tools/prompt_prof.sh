#!/bin/bash
# kernel totals of the headline's arrangement with a 75-id prompt against a 4-id prompt
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/promptprof
rm -rf $O && mkdir -p $O
cd $R
for P in 75 4; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/t$P -o t --output-format csv -- python3 tools/prompt_probe.py $P 14 > $O/run$P.log 2>&1
  cp $(find $O/t$P -name "*_kernel_stats.csv") $O/kernel_stats_$P.csv
  rm -rf $O/t$P
  echo "== prompt $P"; tail -1 $O/run$P.log; cut -d, -f1-5 $O/kernel_stats_$P.csv | cut -c1-170 | head -16
done
