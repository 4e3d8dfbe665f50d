#!/usr/bin/env python3
"""Development aid: what runs beside what in a rocprofv3 kernel trace (kernel_trace.csv) of the grouped arrangement.
Kernels are put in classes (cross-attention stream, encoder products, encoder attention, the rest of the decode step,
other); over the middle half of the trace the script prints, per class, the time at least one of its kernels was in
flight, and the share of the wall time spent in each combination of classes (how many cross-attention launches at
once, with or without an encoder product beside them, nothing at all).
Usage: python tools/timeline_summary.py kernel_trace.csv"""
import csv
import sys
from collections import defaultdict


def cls(name):
    if "cross_attn_stream" in name or "cross_attn8_stream" in name:
        return "xstream"
    if "gemm256" in name or "gemm_kernel" in name:
        return "encgemm"
    if "enc_attention" in name:
        return "encattn"
    if "gemm_dec" in name or "layernorm" in name or "dec_attn" in name or "logits" in name or "gemv" in name or "finalize" in name:
        return "chain"
    return "other"


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), cls(r["Kernel_Name"])))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    lo, hi = t0 + (t1 - t0) // 4, t0 + 3 * (t1 - t0) // 4
    ev = []
    for s, e, c in rows:
        s, e = max(s, lo), min(e, hi)
        if e > s:
            ev.append((s, 1, c))
            ev.append((e, -1, c))
    ev.sort()
    active = defaultdict(int)
    combo = defaultdict(int)
    per_class = defaultdict(int)
    prev = lo
    for t, d, c in ev:
        if t > prev:
            key = (min(active["xstream"], 3), active["encgemm"] > 0, active["encattn"] > 0, active["chain"] > 0, active["other"] > 0)
            combo[key] += t - prev
            for k in ("xstream", "encgemm", "encattn", "chain", "other"):
                if active[k] > 0:
                    per_class[k] += t - prev
            prev = t
        active[c] += d
    wall = hi - lo
    print(f"window {wall / 1e6:.1f} ms")
    for k, v in sorted(per_class.items(), key=lambda kv: -kv[1]):
        print(f"  {k:8s} in flight {100.0 * v / wall:5.1f} % of the window")
    dur = defaultdict(int)
    cnt = defaultdict(int)
    for s, e, c in rows:
        if s >= lo and e <= hi:
            dur[c] += e - s
            cnt[c] += 1
    for k in dur:
        print(f"  {k:8s} {cnt[k]:7d} launches, summed durations {100.0 * dur[k] / wall:6.1f} % of the window, mean {dur[k] / cnt[k] / 1e3:7.1f} us")
    print("  combinations (cross-attention launches in flight, encoder product, encoder attention, decode chain, other):")
    for k, v in sorted(combo.items(), key=lambda kv: -kv[1])[:24]:
        print(f"    x{k[0]} gemm={int(k[1])} attn={int(k[2])} chain={int(k[3])} other={int(k[4])}: {100.0 * v / wall:5.1f} %")
    none = sum(v for k, v in combo.items() if k[0] == 0 and not k[1] and not k[2])
    print(f"  neither cross-attention nor encoder kernels in flight: {100.0 * none / wall:5.1f} %")


if __name__ == "__main__":
    main()
