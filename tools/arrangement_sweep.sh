# the headline arrangement with other group sizes / engine counts: tools/arrangement_sweep.sh "<bench args>" "<bench args>" ...
for args in "$@"; do
  echo "== $args"
  timeout -k 10 400 python bench.py --only-main $args 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'])" || exit 1
done
