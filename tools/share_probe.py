#!/usr/bin/env python3
"""Development aid: a short job (the share of one rank when one hour of audio is split over 8, 4 or 2 GPUs: 15, 30 or 60
chunks) through GroupedPool under different splits into decode groups, and through EnginePool with the batch cut up.
Usage: python tools/share_probe.py [tokens]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import manual_whisper_amd as mw  # noqa: E402  (before torch: raises GPU_MAX_HW_QUEUES)
import torch  # noqa: E402
from manual_whisper_amd import synth  # noqa: E402
from manual_whisper_amd.decode_options import synthetic_options  # noqa: E402
from manual_whisper_amd.engine import Engine  # noqa: E402
from manual_whisper_amd.scheduler import EnginePool, GroupedPool, batches_of  # noqa: E402

tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dims = mw.get_dims("large-v3")
dev = torch.device("cuda", 0)
opts = synthetic_options(dims, max_new=tokens, suppress_eot=True)
eng = Engine(dims, device=0, max_batch=16)
eng.load_weights(synth.make_weights_torch(dims, seed=1234, device=dev))
eng.set_decode_options(opts)
pcm = synth.make_pcm_torch(64, seed=7, device=dev)
torch.cuda.synchronize()


def timed(pool, batches, sizes=None, reps=3):
    kw = {"sizes": sizes} if sizes is not None else {}
    pool.map(batches, **kw)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter()
        pool.map(batches, **kw)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return best


for n in (15, 30, 60):
    print(f"== {n} chunks", flush=True)
    chunks = pcm[:n]
    for bs, group, n_enc, n_dec, sizes in ((16, 7, 2, 1, None), (16, 2, 2, 2, None), (16, 1, 2, 2, None), (16, 1, 2, 3, None), (16, 1, 2, 4, None),
                                           (8, 2, 2, 2, None), (8, 4, 2, 2, None), (8, 1, 2, 4, None), (8, 3, 2, 3, None)):
        nb = -(-n // bs)
        if group > nb and (group, n_dec) != (7, 1):
            continue
        g = min(group, nb)
        pool = GroupedPool(eng, bs, g, min(n_enc, nb), min(n_dec, -(-nb // g)))
        pool.set_decode_options(opts)
        dt = timed(pool, [chunks[a:b] for a, b in batches_of(n, bs)])
        print(f"  grouped  batch {bs:2d} group {g} encoders {min(n_enc, nb)} decoders {len(pool.decoders)}: {dt * 1e3:7.1f} ms  {n * 30 / dt:7.0f} audio-s/s", flush=True)
        for e in reversed(pool.engines):
            if e is not eng:
                e.close()
    for bs, ne in ((16, 1), (16, 2), (8, 2), (8, 4), (5, 3), (4, 4)):
        nb = -(-n // bs)
        pool = EnginePool(eng, min(ne, nb))
        for e in pool.engines:
            e.set_decode_options(opts)
        dt = timed(pool, [chunks[a:b] for a, b in batches_of(n, bs)])
        print(f"  whole passes of {bs:2d} on {min(ne, nb)} engines: {dt * 1e3:7.1f} ms  {n * 30 / dt:7.0f} audio-s/s", flush=True)
        for e in reversed(pool.engines):
            if e is not eng:
                e.close()
