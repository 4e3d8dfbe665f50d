#!/usr/bin/env python3
"""Development aid for hardware-counter runs: a torch-free process (MW_SYSTEM_HIP=1, system ROCm
runtime) that runs log-mel, the encoder, cross K/V and a few large-v3 decode steps at batch 16 on
random weights (MW_WEIGHTS=fp8 / MW_KV=fp8 select the e4m3 formats).  Used as `rocprofv3 --pmc FETCH_SIZE -- python3 tools/pmc_probe.py`, with
WRITE_SIZE and the SQ / GRBM counters each in a pass of their own; tools/pmc_summary.py turns
the per-dispatch CSVs into one table."""
import os, sys
os.environ["MW_SYSTEM_HIP"] = "1"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manual_whisper_amd as mw
from manual_whisper_amd import synth
from manual_whisper_amd.decode_options import synthetic_options
from manual_whisper_amd.engine import Engine

B = int(os.environ.get("MW_B", "16"))      # MW_B=112: the shape of a decode group
dims = mw.get_dims("large-v3")
rng = np.random.default_rng(0)
eng = Engine(dims, 0, B, weight_format=os.environ.get("MW_WEIGHTS", "f16"), kv_format=os.environ.get("MW_KV", "f16"))
w = {}
for name, shape, kind, scale in synth.tensor_specs(dims):
    n = int(np.prod(shape))
    a = rng.standard_normal(n, dtype=np.float32) * np.float32(scale / 3.46)
    if kind == "gamma":
        a += 1
    w[name] = (a.astype(np.float16) if kind in ("mat", "sin") else a).reshape(shape)
eng.load_weights(w)
del w
eng.set_decode_options(synthetic_options(dims, max_new=int(os.environ.get("MW_NEW", "3")), suppress_eot=True))
pcm = synth.make_pcm_np(B, seed=7)
eng.logmel(pcm, want_output=False)
eng.encode(B, want_output=False)
eng.set_option("graph", 0)
if os.environ.get("MW_SHARED", "0") == "1":      # the kernel forms the pools select for engines that share a GPU
    eng.set_option("xattn_stream_per_cu", 1)
    eng.set_option("dec_gemm_fat", 1)
eng.decode_greedy(B)
print("done", eng.stage_ms())
