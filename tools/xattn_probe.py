#!/usr/bin/env python3
"""Development aid: decode cross-attention kernel bandwidth versus batch size (workgroups = batch*heads)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import manual_whisper_amd as mw
from manual_whisper_amd import synth
from manual_whisper_amd.engine import Engine

dims = mw.get_dims("large-v3")
dev = torch.device("cuda", 0)
w = synth.make_weights_torch(dims, 1234, dev)
eng = Engine(dims, 0, 32)
eng.load_weights(w)
del w
for B in (4, 8, 12, 13, 16, 19, 24, 25, 32):
    mel = np.zeros((B, dims.n_mels, 3000), np.float32)
    eng.encode(B, mel=mel, want_output=False)
    ms, nbytes = eng.profile_cross_attention(B, 128)
    print(f"B={B:2d} blocks={B*dims.n_heads:4d} {ms*1e3:7.2f} us  {nbytes/ms/1e6:8.1f} GB/s", flush=True)
