# HBM counters of the fp16 engine's kernels (FETCH_SIZE and WRITE_SIZE, one pass each) from tools/pmc_probe.py.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r1d_f16
mkdir -p $O
cd $R
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/p_fetch -o p --output-format csv -- python3 tools/pmc_probe.py > $O/p_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/p_write -o p --output-format csv -- python3 tools/pmc_probe.py > $O/p_write.log 2>&1
python3 tools/pmc_summary.py $O/pmc_f16.csv $O/p_fetch $O/p_write
rm -rf $O/p_fetch $O/p_write
echo PMC_DONE
