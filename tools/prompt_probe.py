#!/usr/bin/env python3
"""Development aid: the headline's arrangement with a forced prompt of PROMPT ids per chunk (the reference's initial_prompt
makes it about 75), `steps` batches of 16 through GroupedPool; prints ms per step.  Under rocprofv3 --kernel-trace --stats it
shows what the prompt prefill of a decode group costs.
Usage: python tools/prompt_probe.py [prompt ids, default 75] [steps, default 14] [tokens, default 128]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import manual_whisper_amd as mw  # noqa: E402  (before torch: raises GPU_MAX_HW_QUEUES)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from manual_whisper_amd import synth  # noqa: E402
from manual_whisper_amd.decode_options import DecodeOptions, synthetic_options  # noqa: E402
from manual_whisper_amd.engine import Engine  # noqa: E402
from manual_whisper_amd.scheduler import GroupedPool  # noqa: E402

n_prompt = int(sys.argv[1]) if len(sys.argv) > 1 else 75
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 14
tokens = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dims = mw.get_dims("large-v3")
dev = torch.device("cuda", 0)
base = synthetic_options(dims, max_new=tokens, suppress_eot=True)
rng = np.random.default_rng(0)
prompt = tuple(int(t) for t in rng.integers(0, 50000, size=max(0, n_prompt - len(base.prompt)))) + tuple(base.prompt)
opts = DecodeOptions(prompt, base.eot, base.suppress, base.begin_suppress, tokens)
eng = Engine(dims, device=0, max_batch=16)
eng.load_weights(synth.make_weights_torch(dims, seed=1234, device=dev))
eng.set_decode_options(opts)
pool = GroupedPool(eng, 16, 7, 2, 3)
pool.set_decode_options(opts)
pcms = [synth.make_pcm_torch(16, seed=7000 + i, device=dev) for i in range(2)]
torch.cuda.synchronize()
pool.map([pcms[0]] * 21)
torch.cuda.synchronize()
t0 = time.perf_counter()
pool.map([pcms[i % 2] for i in range(steps)])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"prompt {len(prompt)} ids + {tokens} sampled: {dt / steps * 1e3:.1f} ms per step of 16 chunks, {steps * 16 * 30 / dt:.0f} audio-s/s; "
      f"a decode engine's last loop {pool.decoders[0].stage_ms()['decode']:.1f} ms", flush=True)
pool.close()
