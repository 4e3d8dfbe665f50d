# kernel stats of the forced-alignment acoustic model alone (run on the GPU box from the repo root)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/alignprof
rm -rf $O && mkdir -p $O
cd $R
MW_ALIGN_STREAMS=${MW_ALIGN_STREAMS:-1} timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/t -o t --output-format csv -- python3 tools/align_bench.py > $O/run.log 2>&1
cp $(find $O/t -name "*_kernel_stats.csv") $O/kernel_stats.csv
rm -rf $O/t
cut -d, -f1-5 $O/kernel_stats.csv | cut -c1-150 | head -14
tail -1 $O/run.log | cut -c1-400
