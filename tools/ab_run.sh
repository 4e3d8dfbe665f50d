#!/bin/bash
# tools/ab_run.sh <out-prefix> <bench args...> -- <variant names...>: bench.py once per tools/ab/<variant>/libmwhip.so
out=$1; shift
args=()
while [ "$1" != "--" ]; do args+=("$1"); shift; done
shift
for v in "$@"; do
  MW_LIB_PATH=$PWD/tools/ab/$v/libmwhip.so timeout -k 10 300 python bench.py --no-cpu-baseline --also-batch 0 --no-fp8-b32 --no-ungrouped --no-align "${args[@]}" > ${out}_$v.log 2>&1 || { echo "variant $v failed"; tail -5 ${out}_$v.log; exit 1; }
  python - "$v" ${out}_$v.log <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
sr = d["stage_roofline"]
print(f"{sys.argv[1]:>10}: value {d['value']:8.1f}  ms/step {d['ms_per_step']:7.2f}  solo {sr['solo_step_ms']:7.2f}  xattn {d['roofline']['avg_launch_us']:6.2f} us  "
      f"gemv/layer {d.get('roofline_gemv', {}).get('avg_layer_us', 0):6.2f} us  decode frac {sr['decode']['frac']:.4f} enc {sr['encoder']['achieved']}", flush=True)
PY
done
