#!/usr/bin/env python3
"""Development aid: beam-search throughput (large-v3, 128 tokens, EOT suppressed) against greedy.
Usage: python tools/beam_probe.py [beam] [recordings per pass] [engines] [passes per engine]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("MW_HW_QUEUES", "8")
import torch

import manual_whisper_amd as mw
from manual_whisper_amd import synth
from manual_whisper_amd.decode_options import synthetic_options
from manual_whisper_amd.engine import Engine
from manual_whisper_amd.scheduler import EnginePool

beam = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n_rec = int(sys.argv[2]) if len(sys.argv) > 2 else 16
engines = int(sys.argv[3]) if len(sys.argv) > 3 else 1
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda", 0)
dims = mw.get_dims("large-v3")
cached = os.environ.get("MW_BEAM_CACHED", "1") == "1"      # forks whose K/V cache holds the recordings once (what load_model builds)
eng = Engine(dims, device=0, max_batch=16 if cached else n_rec * beam, weight_format=os.environ.get("MW_WEIGHTS", "f16"),
             kv_format=os.environ.get("MW_KV", "f16"))
eng.load_weights(synth.make_weights_torch(dims, seed=1234, device=dev))
eng.set_decode_options(synthetic_options(dims, max_new=128, suppress_eot=True))
pool = EnginePool(eng, engines, beam_size=beam, fork_batch=n_rec * beam, fork_cache=n_rec) if cached else EnginePool(eng, engines, beam_size=beam)
pool.set_decode_options(synthetic_options(dims, max_new=128, suppress_eot=True))
pcms = [synth.make_pcm_torch(n_rec, seed=i, device=dev) for i in range(3)]
pool.map([pcms[i % 3] for i in range(engines)])           # graphs captured
torch.cuda.synchronize()
t = time.perf_counter()
pool.map([pcms[i % 3] for i in range(engines * passes)])
dt = time.perf_counter() - t
print(f"beam {beam}, {n_rec} recordings per pass ({n_rec * beam} sequences), {engines} engines: "
      f"{engines * passes * n_rec * 30 / dt:.0f} audio-s/s, {1e3 * dt / (engines * passes):.1f} ms per pass; stages {eng.stage_ms()}", flush=True)
pool.close()
