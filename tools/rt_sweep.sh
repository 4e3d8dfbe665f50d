set -e
cd $GRAFT_REPO_ROOT
val() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for v in 1 2 0 1 2; do
echo "== grouped, MW_XSTREAM_PER_CU=$v"; MW_XSTREAM_PER_CU=$v timeout -k 10 200 python bench.py --only-main --steps 42 --warmup 7 2>/dev/null | val
done
