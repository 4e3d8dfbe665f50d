#!/usr/bin/env python3
"""Development aid: how long do N decode loops take when they run concurrently with NO encoder in
flight, against N whole steps with phases left to overlap freely?"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import manual_whisper_amd as mw
from manual_whisper_amd import synth
from manual_whisper_amd.decode_options import synthetic_options
from manual_whisper_amd.engine import Engine

B = 16
dims = mw.get_dims("large-v3")
dev = torch.device("cuda", 0)
w = synth.make_weights_torch(dims, 1234, dev)
opts = synthetic_options(dims, max_new=128, suppress_eot=True)
NE = int(os.environ.get("NE", "4"))
root = Engine(dims, 0, B); root.load_weights(w); root.set_decode_options(opts)
del w
engs = [root] + [root.fork() for _ in range(NE - 1)]
for e in engs[1:]:
    e.set_decode_options(opts)
pcm = synth.make_pcm_torch(B, 5, dev)
for e in engs:
    e.transcribe_batch(pcm)

def par(fn, n):
    th = [threading.Thread(target=fn, args=(engs[i],)) for i in range(n)]
    [t.start() for t in th]; [t.join() for t in th]

for n in range(1, NE + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for e in engs[:n]:
        e.logmel(pcm, want_output=False); e.encode(B, want_output=False)
    t1 = time.perf_counter()
    par(lambda e: e.decode_greedy(B), n)
    t2 = time.perf_counter()
    print(f"{n} engines: encoders one after another {1e3*(t1-t0):.1f} ms, {n} decodes together {1e3*(t2-t1):.1f} ms "
          f"-> {1e3*(t2-t0)/n:.1f} ms per step, {n*B*30/(t2-t0):.0f} audio-s/s", flush=True)
