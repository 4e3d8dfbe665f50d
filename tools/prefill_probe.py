#!/usr/bin/env python3
"""Decode time of one large-v3 batch with the prompt stepped through or prefilled in one pass.
    python tools/prefill_probe.py [batch] [new_tokens]
Prompt lengths: 4 (task tokens only) and 75 (about what the reference's initial_prompt tokenises to)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import manual_whisper_amd as mw
from manual_whisper_amd import synth
from manual_whisper_amd.decode_options import DecodeOptions, synthetic_options
from manual_whisper_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
NEW = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dims = mw.get_dims("large-v3")
dev = torch.device("cuda", 0)
eng = Engine(dims, 0, B)
eng.load_weights(synth.make_weights_torch(dims, 1234, dev))
pcm = synth.make_pcm_torch(B, 7, dev)
base = synthetic_options(dims, max_new=NEW, suppress_eot=True)
rng = np.random.default_rng(0)
for n_prompt in (4, 75):
    prompt = tuple(int(t) for t in rng.integers(0, 50000, size=n_prompt - 1)) + (base.prompt[-1],)
    eng.set_decode_options(DecodeOptions(prompt, base.eot, base.suppress, base.begin_suppress, NEW))
    res = {}
    for prefill in (0, 1):
        eng.set_option("prefill", prefill)
        eng.transcribe_batch(pcm)
        ids, _ = eng.transcribe_batch(pcm)
        res[prefill] = (eng.stage_ms()["decode"], ids)
    same = float((res[0][1] == res[1][1]).all(axis=1).mean())
    print(f"batch {B}, prompt {n_prompt:3d} + {NEW} sampled: decode {res[0][0]:7.1f} ms stepping, {res[1][0]:7.1f} ms with prefill "
          f"({res[0][0] - res[1][0]:+.1f} ms); sequences with identical ids: {same:.2f}", flush=True)
eng.close()
