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
rng = np.random.default_rng(0)
audio = (0.1 * rng.standard_normal(minutes * 60 * 16000)).astype(np.float32)
t = time.perf_counter()
model = mw.load_model("synthetic:large-v3", "cuda", compute_type=os.environ.get("MW_DTYPE", "float16"), language="zh",
                      asr_options={"max_new_tokens": 128}, vad_options=vad,
                      max_batch=16 if os.environ.get("E2E_ARRANGE", "grouped") == "grouped" else min(128, 16 * coalesce),
                      n_engines=engines, coalesce=coalesce, arrangement=os.environ.get("E2E_ARRANGE", "grouped"))
print(f"load {time.perf_counter() - t:.2f} s", flush=True)
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
model.close()
