#!/bin/bash
# the default arrangement with the decode cross-attention stream kernel held to fewer workgroups per CU by an LDS request
for v in ${SWEEP:-0 54000 80000 0 54000}; do
  echo "== MW_XSTREAM_LDS=$v"
  MW_XSTREAM_LDS=$v timeout -k 10 300 python bench.py --only-main --steps 42 --warmup 21 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])" || exit 1
done
