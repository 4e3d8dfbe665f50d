# LDS counters of the matrix-product kernels (run on the GPU box from the repo root): bank-conflict cycles against
# all LDS-array cycles, one pass per counter set
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/gemmpmc
rm -rf $O && mkdir -p $O
cd $R
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES -d $O/p1 -o p --output-format csv -- tools/gemm_bench ${1:-24000} ${2:-3840} ${3:-1280} 3 ${4:-} > $O/run1.log 2>&1
python3 tools/pmc_summary.py $O/lds.csv $O/p1
rm -rf $O/p1
cat $O/lds.csv
