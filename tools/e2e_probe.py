#!/usr/bin/env python3
"""Development aid: where the time of Model.transcribe goes on a long recording (windows, engine passes, text),
first call (creates the engines, captures the decode graphs) and later calls.
Usage: python tools/e2e_probe.py [minutes] [engines] [coalesce]; E2E_ARRANGE=grouped (default) | coalesced"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# (the package raises GPU_MAX_HW_QUEUES itself when it is imported, before torch: _lib.configure_hw_queues)
import numpy as np

import manual_whisper_amd as mw

minutes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
engines = int(sys.argv[2]) if len(sys.argv) > 2 else 2
coalesce = int(sys.argv[3]) if len(sys.argv) > 3 else 4
vad = {"vad_onset": 0.5, "vad_offset": 0.363} if os.environ.get("E2E_VAD", "1") == "1" else None
CHURN = os.environ.get("E2E_CHURN", "")
# what a long-lived process has behind it when the model under test is built: 1 = a pool created, run and closed, its
# root engine still alive; 2 = the root closed too; 3 = only a root engine, alive; 4 = as 1 without running the pool;
# 5 = as 1, the pool closed only after the model has been loaded; 7 = as 4, but all of it AFTER the model has been loaded
_state = {}


def churn_open():
    import torch
    from manual_whisper_amd import synth
    from manual_whisper_amd.decode_options import synthetic_options
    from manual_whisper_amd.engine import Engine
    from manual_whisper_amd.scheduler import GroupedPool
    dims0 = mw.get_dims("large-v3")
    dev0 = torch.device("cuda", 0)
    e0 = Engine(dims0, device=0, max_batch=16)
    e0.load_weights(synth.make_weights_torch(dims0, seed=1234, device=dev0))
    _state["e0"] = e0
    if CHURN == "3":
        return
    p0 = GroupedPool(e0, 16, 7, 2, 3)
    p0.set_decode_options(synthetic_options(dims0, max_new=8, suppress_eot=True))
    if CHURN in ("1", "2", "5"):
        p0.map([synth.make_pcm_torch(16, seed=1, device=dev0)] * 7)
    _state["p0"] = p0


def churn_close():
    import torch
    p0, e0 = _state.get("p0"), _state["e0"]
    if p0 is not None:
        for e in reversed(p0.engines):
            if e is not e0:
                e.close()
    if CHURN == "2":
        e0.close()
    torch.cuda.empty_cache()


if CHURN and CHURN != "7":
    churn_open()
    if CHURN != "5":
        churn_close()
rng = np.random.default_rng(0)
audio = (0.1 * rng.standard_normal(minutes * 60 * 16000)).astype(np.float32)
t = time.perf_counter()
model = mw.load_model("synthetic:large-v3", "cuda", compute_type=os.environ.get("MW_DTYPE", "float16"), language="zh",
                      asr_options={"max_new_tokens": 128}, vad_options=vad,
                      max_batch=16 if os.environ.get("E2E_ARRANGE", "grouped") == "grouped" else min(128, 16 * coalesce),
                      n_engines=engines, coalesce=coalesce, arrangement=os.environ.get("E2E_ARRANGE", "grouped"))
print(f"load {time.perf_counter() - t:.2f} s", flush=True)
if CHURN == "5":
    churn_close()
if CHURN == "7":
    churn_open()
    churn_close()
for call in (1, 2, 3):
    t0 = time.perf_counter()
    chunks, spans = model.windows(audio)
    t1 = time.perf_counter()
    ids = model.transcribe_ids_chunks(chunks, 16, "zh")
    t2 = time.perf_counter()
    text = [model.detokenize(i) for i in ids]
    t3 = time.perf_counter()
    n_tok = sum(len(i) for i in ids)
    print(f"call {call}: {len(chunks)} windows; windows {t1 - t0:.3f} s, passes {t2 - t1:.3f} s, text {t3 - t2:.3f} s; "
          f"{n_tok} tokens; {minutes * 60 / (t3 - t0):.0f} audio-s/s ({minutes * 60 / (t2 - t1):.0f} in the passes)", flush=True)
for call in (1, 2):
    t3 = time.perf_counter()
    res = model.transcribe(audio, batch_size=16, language="zh")
    t4 = time.perf_counter()
    print(f"transcribe() call {call}: {t4 - t3:.3f} s for {len(res['segments'])} segments", flush=True)
if getattr(model.pool, "trace", None):
    tr = model.pool.trace
    for kind in ("encode", "decode"):
        d = [e - s_ for k, _, s_, e in tr if k == kind]
        if d:
            print(f"last map: {len(d)} x {kind}: mean {1e3 * sum(d) / len(d):.1f} ms, max {1e3 * max(d):.1f} ms", flush=True)
model.close()
