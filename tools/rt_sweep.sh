# decode projections as fewer, larger workgroups (option dec_gemm_fat, set by the pools): the benchmark's arrangement,
# one decode group alone, beam search on three engines
set -e
cd $GRAFT_REPO_ROOT
val() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for fat in 1 0 1 0; do
echo "== grouped, MW_DEC_GEMM_FAT=$fat"; MW_DEC_GEMM_FAT=$fat timeout -k 10 200 python bench.py --only-main --steps 40 --warmup 5 2>/dev/null | val
done
echo "== one group alone"; timeout -k 10 200 python bench.py --only-main --group 1 --engines 1 --batch 112 --steps 3 --warmup 1 2>/dev/null | val
for fat in 1 0; do
echo "== beam, MW_DEC_GEMM_FAT=$fat"; MW_DEC_GEMM_FAT=$fat timeout -k 10 120 python3 tools/beam_probe.py 5 48 3 3 | cut -c1-110
done
