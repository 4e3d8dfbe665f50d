#!/usr/bin/env python3
"""Development aid: explicit splits of a short job's batches of 16 into decode groups (GroupedPool.map(sizes=...)),
2 encoder engines, one decode engine per group up to 3.  Usage: python tools/share_probe2.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import manual_whisper_amd as mw  # noqa: E402  (before torch: raises GPU_MAX_HW_QUEUES)
import torch  # noqa: E402
from manual_whisper_amd import synth  # noqa: E402
from manual_whisper_amd.decode_options import synthetic_options  # noqa: E402
from manual_whisper_amd.engine import Engine  # noqa: E402
from manual_whisper_amd.scheduler import GroupedPool, batches_of  # noqa: E402

dims = mw.get_dims("large-v3")
dev = torch.device("cuda", 0)
opts = synthetic_options(dims, max_new=128, suppress_eot=True)
eng = Engine(dims, device=0, max_batch=16)
eng.load_weights(synth.make_weights_torch(dims, seed=1234, device=dev))
eng.set_decode_options(opts)
pcm = synth.make_pcm_torch(16, seed=7, device=dev)
torch.cuda.synchronize()
cases = {3: ([3], [2, 1], [1, 1, 1]), 4: ([4], [2, 2], [2, 1, 1]), 5: ([5], [3, 2], [2, 2, 1]), 6: ([6], [3, 3], [2, 2, 2]),
         8: ([4, 4], [3, 3, 2], [7, 1]), 12: ([6, 6], [4, 4, 4]), 14: ([7, 7], [5, 5, 4])}
for nb, splits in cases.items():
    print(f"== {nb} batches of 16 ({nb * 16} chunks)", flush=True)
    for sizes in splits:
        pool = GroupedPool(eng, 16, max(sizes), 2, min(3, len(sizes)))
        pool.set_decode_options(opts)
        batches = [pcm] * nb
        pool.map(batches, sizes=sizes)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(2):
            t = time.perf_counter()
            pool.map(batches, sizes=sizes)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t)
        print(f"  groups {sizes}: {best * 1e3:7.1f} ms  {nb * 16 * 30 / best:7.0f} audio-s/s", flush=True)
        for e in reversed(pool.engines):
            if e is not eng:
                e.close()
