#!/usr/bin/env python3
"""Development aid: run a few decode steps of large-v3 (batch 16) so that a rocprofv3 kernel trace
of this process shows one duration per decode kernel.  Prints nothing but a completion line."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import manual_whisper_amd as mw
from manual_whisper_amd import synth
from manual_whisper_amd.decode_options import synthetic_options
from manual_whisper_amd.engine import Engine

B = int(os.environ.get("MW_B", "16"))
dims = mw.get_dims(os.environ.get("MW_MODEL", "large-v3"))
dev = torch.device("cuda", 0)
w = synth.make_weights_torch(dims, 1234, dev)
eng = Engine(dims, 0, B)
eng.load_weights(w)
del w
eng.set_decode_options(synthetic_options(dims, max_new=int(os.environ.get("MW_NEW", "40")), suppress_eot=True))
eng.encode(B, mel=np.zeros((B, dims.n_mels, 3000), np.float32), want_output=False)
for _ in range(2):
    eng.decode_greedy(B)
print("done", eng.stage_ms())
