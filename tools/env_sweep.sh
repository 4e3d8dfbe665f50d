#!/bin/bash
# Development aid: the benchmark step under sets of environment knobs.
# usage: tools/env_sweep.sh out.log steps warmup "VAR=val VAR2=val" ...   (an empty string = no knobs)
out=$1; steps=$2; warm=$3; shift 3
: > "$out"
for c in "$@"; do
  echo "== [$c] steps $steps ${BENCH_ARGS}" >> "$out"
  env $c timeout -k 10 240 python bench.py --no-cpu-baseline --no-fp8-b32 --no-ungrouped --no-align --steps "$steps" --warmup "$warm" ${BENCH_ARGS} 2>&1 | python -c '
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print(d["value"], d["ms_per_step"], d["stage_roofline"].get("decode_group", {}).get("loop_ms"), d["roofline"]["avg_launch_us"])
    elif l.strip() and "amdgpu.ids" not in l: print(l.rstrip()[:300])
' >> "$out" || { echo "FAILED rc=$?" >> "$out"; exit 1; }
done
