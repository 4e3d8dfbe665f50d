#!/bin/bash
# one environment knob over the default arrangement: tools/knob_sweep.sh NAME "v1 v2 ..." [bench args]
name=$1; vals=$2; shift 2
for v in $vals; do
  echo "== $name=$v $*"
  env $name=$v timeout -k 10 300 python bench.py --only-main --steps 42 --warmup 21 "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'])" || exit 1
done
