set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_beam.py -m gpu -x -q --timeout 120 --timeout-method thread -k "beyond or beam or ended" 2>&1 | tail -4
val() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for v in 1 0 1 0; do
echo "== grouped, MW_LOGITS_WIDE=$v"; MW_LOGITS_WIDE=$v timeout -k 10 200 python bench.py --only-main --steps 42 --warmup 7 2>/dev/null | val
done
for v in 1 0; do
echo "== one group alone, MW_LOGITS_WIDE=$v"; MW_LOGITS_WIDE=$v timeout -k 10 200 python bench.py --only-main --group 1 --engines 1 --batch 112 --steps 3 --warmup 1 2>/dev/null | val
echo "== beam, MW_LOGITS_WIDE=$v"; MW_LOGITS_WIDE=$v timeout -k 10 120 python3 tools/beam_probe.py 5 48 3 3 | cut -c1-110
done
