#!/usr/bin/env python3
"""Development aid: from a rocprofv3 kernel trace (csv), how the busiest second of the run divides into time with
the decode cross-attention (HBM-bound) running, with encoder products / attention (MFMA-bound) running, with
both, with only the short decode kernels, and with nothing.  Usage: python tools/overlap_classes.py kernel_trace.csv"""
import csv
import sys

rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if "mw" not in name:
            continue
        if "dec_attn_kernelILi256ELi12ELi4" in name or "dec_attn_kernel<256, 12, 4" in name:
            cls = "X"
        elif "gemm256" in name or "gemm_kernel" in name or "enc_attention" in name or "mel_" in name:
            cls = "G"
        else:
            cls = "S"
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), cls, r.get("Queue_Id", r.get("Stream_Id", "0"))))
rows.sort()
# the stretch in which several engines run at once: 10 ms bins with kernels of at least three queues
BIN = 10_000_000
t0 = rows[0][0]
active = {}
for s_, e_, c_, q_ in rows:
    for b in range((s_ - t0) // BIN, (e_ - t0) // BIN + 1):
        active.setdefault(b, set()).add(q_)
multi = sorted(b for b, v in active.items() if len(v) >= 3)
# the longest run of such bins (gaps of up to 5 bins allowed) is the warm-up + timed steps; take its last third
runs, cur = [], [multi[0]]
for b in multi[1:]:
    if b - cur[-1] <= 5:
        cur.append(b)
    else:
        runs.append(cur)
        cur = [b]
runs.append(cur)
best = max(runs, key=lambda r: r[-1] - r[0])
lo_b, hi_b = best[0], best[-1]
lo, hi = t0 + (lo_b + (hi_b - lo_b) * 2 // 3) * BIN, t0 + hi_b * BIN      # set-up, warm-up, timed: the last third
rows = [(s_, e_, c_) for s_, e_, c_, _ in rows]
ev = []
for s, e, c in rows:
    s, e = max(s, lo), min(e, hi)
    if s < e:
        ev.append((s, 1, c))
        ev.append((e, -1, c))
ev.sort()
cnt = {"X": 0, "G": 0, "S": 0}
acc = {}
last = lo
for t, d, c in ev:
    key = ("X" if cnt["X"] else "-") + ("G" if cnt["G"] else "-") + ("S" if cnt["S"] else "-")
    acc[key] = acc.get(key, 0) + (t - last)
    last = t
    cnt[c] += d
acc["---"] = acc.get("---", 0) + (hi - last)
span = hi - lo
print(f"span {span / 1e6:.0f} ms; X = decode cross-attention, G = encoder products / attention / log-mel, S = other decode kernels")
for k in sorted(acc, key=lambda k: -acc[k]):
    print(f"  {k}: {acc[k] / 1e6:8.1f} ms  {100.0 * acc[k] / span:5.1f} %")
tot = {}
for s, e, c in rows:
    s, e = max(s, lo), min(e, hi)
    if s < e:
        tot[c] = tot.get(c, 0) + (e - s)
for c, v in tot.items():
    print(f"  summed kernel time {c}: {v / 1e6:.1f} ms ({v / span:.2f} of the span)")
