#!/usr/bin/env python3
"""Development aid: throughput of 1 vs 2 vs 3 independent engines driven from host threads."""
import os, sys, time, threading
import numpy as np
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
NE = int(os.environ.get("NE", "3"))
engs = []
for i in range(NE):
    e = Engine(dims, 0, B); e.load_weights(w); e.set_decode_options(opts); engs.append(e)
del w
pcm = synth.make_pcm_torch(B, 5, dev)
for e in engs:
    e.transcribe_batch(pcm)
for n in range(1, NE + 1):
    reps = 3
    def work(e):
        for _ in range(reps):
            e.transcribe_batch(pcm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(engs[i],)) for i in range(n)]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    print(f"{n} engines: {n*reps} batches in {dt*1e3:.1f} ms -> {n*reps*B*30/dt:.0f} audio-s/s ({dt/(n*reps)*1e3:.1f} ms per batch)", flush=True)
