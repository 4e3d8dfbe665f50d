# Development aid (run on the GPU box from the repo root): kernel stats of one engine decoding 112 sequences, and the
# FETCH_SIZE pass of tools/pmc_probe.py at 112 sequences.   tools/prof_b112.sh <tag>
set -e
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/b112_$tag
mkdir -p $O
cd $R
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/ks -o ks --output-format csv -- python3 bench.py --no-cpu-baseline --no-fp8-b32 --no-ungrouped --no-align --group 1 --engines 1 --batch 112 --steps 2 --warmup 1 > $O/ks.log 2>&1
cp $(find $O/ks -name "*_kernel_stats.csv") $O/kernel_stats.csv
rm -rf $O/ks
MW_B=112 timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $O/pf -o p --output-format csv -- python3 tools/pmc_probe.py > $O/pmc_fetch.log 2>&1
MW_B=112 timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $O/pw -o p --output-format csv -- python3 tools/pmc_probe.py > $O/pmc_write.log 2>&1
python3 tools/pmc_summary.py $O/pmc_counters.csv $O/pf $O/pw
rm -rf $O/pf $O/pw
echo PROF_DONE
