#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv: per (kernel, grid) count / mean / min duration in us."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "mw" not in n:
        continue
    m = re.search(r"(gemv_kernel<[^>]*>|logits_kernel<\d>|dec_attn_kernel\w*|gemm_kernel|enc_attention_kernel|layernorm_kernel|dec_finalize_kernel|dec_init_kernel|mel_\w+_kernel|xattn\w*|\w+_kernel)", n)
    key = (m.group(1) if m else n[:40], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Workgroup_Size_X"]))
    agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{k[0]:34s} blocks {k[1]:6d} x{k[2]:5d} n {len(v):6d} mean {sum(v)/len(v)/1e3:8.2f} us  p10 {v[len(v)//10]/1e3:8.2f}  share {100*sum(v)/tot:5.1f}%")
