#!/usr/bin/env python3
"""Development aid: from a rocprofv3 results database (kernel trace), report how much of the traced span
had 0, 1, 2, 3+ kernels executing, overall and for the span in which several engines run at once (10 ms bins
with kernels of at least three streams: the warm-up and timed steps of bench.py).
Usage: python tools/overlap_stats.py results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start, end, stream_id, name from kernels where name like '%mw::%' or name like '%_ZN2mw%' order by start").fetchall()
ev = []
for s, e, _, _ in rows:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
hist = {}
cur, last = 0, ev[0][0]
for t, d in ev:
    hist[min(cur, 4)] = hist.get(min(cur, 4), 0) + (t - last)
    cur += d
    last = t
span = ev[-1][0] - ev[0][0]
print(f"engine kernels: {len(rows)}, span {span / 1e6:.1f} ms")
for k in sorted(hist):
    print(f"  {k}{'+' if k == 4 else ' '} kernels in flight: {hist[k] / 1e6:9.1f} ms  {100.0 * hist[k] / span:5.1f} %")
streams = {}
for s, e, sid, _ in rows:
    streams.setdefault(sid, []).append((s, e))
for sid, iv in streams.items():
    busy = sum(e - s for s, e in iv)
    print(f"  stream {sid}: {len(iv)} kernels, busy {busy / 1e6:.1f} ms over {(iv[-1][1] - iv[0][0]) / 1e6:.1f} ms")

# ---- the multi-engine span
BIN = 10_000_000
t0 = ev[0][0]
active = {}
for s, e, sid, _ in rows:
    for b in range((s - t0) // BIN, (e - t0) // BIN + 1):
        active.setdefault(b, set()).add(sid)
multi = sorted(b for b, v in active.items() if len(v) >= 3)
if multi:
    lo, hi = t0 + multi[0] * BIN, t0 + (multi[-1] + 1) * BIN
    hist, cur, last = {}, 0, lo
    for t, d in ev:
        if t > hi:
            break
        if t >= lo:
            hist[min(cur, 4)] = hist.get(min(cur, 4), 0) + (t - last)
            last = t
        cur += d
    span = hi - lo
    print(f"multi-engine span: {span / 1e6:.1f} ms")
    for k in sorted(hist):
        print(f"  {k}{'+' if k == 4 else ' '} kernels in flight: {hist[k] / 1e6:9.1f} ms  {100.0 * hist[k] / span:5.1f} %")
    busy = {}
    for s, e, sid, name in rows:
        if s >= lo and e <= hi:
            key = "gemm/attention (encoder)" if ("gemm" in name or "enc_attention" in name or "layernorm" in name and False) else "other"
            busy[key] = busy.get(key, 0) + (e - s)
    for k, v in busy.items():
        print(f"  summed kernel time, {k}: {v / 1e6:.1f} ms ({v / span:.2f} of the span)")
