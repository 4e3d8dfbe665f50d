#!/usr/bin/env python3
"""Development aid: whole passes of B chunks (B = 4..16) on 1, 2 and 4 engines in flight, every engine always given the
same batch size (so that no decode graph is captured again inside the timed region).
Usage: python tools/small_batch_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import manual_whisper_amd as mw  # noqa: E402  (before torch: raises GPU_MAX_HW_QUEUES)
import torch  # noqa: E402
from manual_whisper_amd import synth  # noqa: E402
from manual_whisper_amd.decode_options import synthetic_options  # noqa: E402
from manual_whisper_amd.engine import Engine  # noqa: E402
from manual_whisper_amd.scheduler import EnginePool  # noqa: E402

dims = mw.get_dims("large-v3")
dev = torch.device("cuda", 0)
opts = synthetic_options(dims, max_new=128, suppress_eot=True)
eng = Engine(dims, device=0, max_batch=16)
eng.load_weights(synth.make_weights_torch(dims, seed=1234, device=dev))
eng.set_decode_options(opts)
pcm = synth.make_pcm_torch(16, seed=7, device=dev)
torch.cuda.synchronize()
for ne in (1, 2, 4):
    pool = EnginePool(eng, ne)
    for e in pool.engines:
        e.set_decode_options(opts)
    for B in (16, 12, 8, 4):
        batches = [pcm[:B]] * (2 * ne)
        pool.map(batches)
        torch.cuda.synchronize()
        t = time.perf_counter()
        pool.map(batches)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        st = pool.engines[-1].stage_ms()
        print(f"{ne} engines x passes of {B:2d}: {dt / 2 * 1e3:7.1f} ms per round of {ne} passes, {2 * ne * B * 30 / dt:7.0f} audio-s/s; "
              f"last engine's stages: encoder {st['encoder']:.1f} ms, decode {st['decode']:.1f} ms", flush=True)
    for e in reversed(pool.engines):
        if e is not eng:
            e.close()
