#!/bin/bash
# rocprofv3 kernel stats of one command: tools/prof_one.sh <tag> <program> [args...]  -> gpurun_out/prof_<tag>.csv
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/prof_one_$tag
rm -rf $O && mkdir -p $O
cd $R
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O -o t --output-format csv -- "$@" > $R/gpurun_out/prof_$tag.log 2>&1
cp $(find $O -name "*_kernel_stats.csv") $R/gpurun_out/prof_$tag.csv
rm -rf $O
cut -d, -f1-4 $R/gpurun_out/prof_$tag.csv | cut -c1-150 | head -14
