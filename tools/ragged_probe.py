#!/usr/bin/env python3
"""Development aid: greedy passes in which part of the sequences end at once (at the first sampled position only
end-of-text and `k` other ids are allowed, so end-of-text wins there for about one sequence in k + 1; afterwards
it is suppressed and the others run to the token limit), with and without skipping the cross-attention of ended
sequences.  Usage: python tools/ragged_probe.py [batch] [k] [engines] [passes]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("MW_HW_QUEUES", "8")
import numpy as np
import torch

import manual_whisper_amd as mw
from manual_whisper_amd import synth
from manual_whisper_amd.decode_options import DecodeOptions, synthetic_options
from manual_whisper_amd.engine import Engine
from manual_whisper_amd.scheduler import EnginePool

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
k = int(sys.argv[2]) if len(sys.argv) > 2 else 40
engines = int(sys.argv[3]) if len(sys.argv) > 3 else 1
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda", 0)
dims = mw.get_dims("large-v3")
base = synthetic_options(dims, max_new=128)
keep = set(np.random.default_rng(7).choice(50000, size=k, replace=False).tolist()) | {base.eot}
opts = DecodeOptions(base.prompt, base.eot, (), tuple(i for i in range(dims.vocab) if i not in keep), 128)
eng = Engine(dims, device=0, max_batch=batch, weight_format=os.environ.get("MW_WEIGHTS", "f16"), kv_format=os.environ.get("MW_KV", "f16"))
eng.load_weights(synth.make_weights_torch(dims, seed=1234, device=dev))
eng.set_decode_options(opts)
pool = EnginePool(eng, engines)
pcms = [synth.make_pcm_torch(batch, seed=i, device=dev) for i in range(3)]
ref = None
for skip in (0, 1, 0, 1):
    for e in pool.engines:
        e.set_option("skip_done", skip)
    pool.map([pcms[i % 3] for i in range(engines)])
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = pool.map([pcms[i % 3] for i in range(engines * passes)])
    dt = time.perf_counter() - t
    lens = np.concatenate([l for _, l in out])
    ids = np.concatenate([i for i, _ in out])
    same = "" if ref is None else f", ids equal to the first run: {bool((ids == ref).all())}"
    ref = ids if ref is None else ref
    print(f"skip_done={skip}: batch {batch} x {engines} engines: {engines * passes * batch * 30 / dt:.0f} audio-s/s; lengths mean {lens.mean():.1f} "
          f"max {lens.max()} (steps run {eng.stage_ms()['decode_steps']}){same}", flush=True)
pool.close()
