#!/bin/bash
# tools/grp_run.sh <out-prefix> "<group> <encoders> <decoders>" ...: bench.py --steps 20 --warmup 5 per configuration
out=$1; shift
for cfg in "$@"; do
  set -- $cfg
  tag=g$1e$2d$3
  timeout -k 10 300 python bench.py --no-cpu-baseline --also-batch 0 --no-fp8-b32 --no-ungrouped --no-align --steps 20 --warmup 5 --group $1 --encoders $2 --decoders $3 ${EXTRA} > ${out}_$tag.log 2>&1 || { echo "$tag failed"; tail -5 ${out}_$tag.log; exit 1; }
  python - "$tag" ${out}_$tag.log <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"{sys.argv[1]:>10}: value {d['value']:8.1f}  ms/step {d['ms_per_step']:7.2f}  stage {d['stage_ms_per_step']}", flush=True)
PY
done
