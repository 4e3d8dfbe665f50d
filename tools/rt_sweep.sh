# decode projections of a 112-sequence group with K ranges of 640 columns (two ranges for K = 1280) in the benchmark
set -e
cd $GRAFT_REPO_ROOT
val() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for v in 0 1 2 0 1 2; do
echo "== grouped, MW_DEC_GEMM_MT7=$v"; MW_DEC_GEMM_MT7=$v timeout -k 10 200 python bench.py --only-main --steps 40 --warmup 5 2>/dev/null | val
done
