#!/usr/bin/env python3
"""Time the forced-alignment acoustic model (wav2vec2-large-xlsr-53 geometry, synthetic weights) on one GPU:
ms per mw_align_emissions call for segments of 5, 15 and 30 s, and the host alignment (trellis + walk) beside it."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manual_whisper_amd  # noqa: F401,E402
from manual_whisper_amd import alignment as al, localckpt, synth  # noqa: E402


def main():
    cfg = localckpt.wav2vec2_config("w2v-large")
    w = localckpt.make_wav2vec2_weights_np(cfg, 77)
    a = al.Aligner(cfg, device=0, max_seconds=30)
    a.load_weights(w)
    out = {}
    for sec in (5, 15, 30):
        wave = synth.hash_uniform_np(sec * 16000, "wave.bench", 3) * np.float32(0.6)
        a.emissions(wave)
        t = time.perf_counter()
        n = 5
        for _ in range(n):
            em = a.emissions(wave)
        ms = (time.perf_counter() - t) / n * 1e3
        tokens = list(np.random.default_rng(0).integers(5, 3000, size=sec * 5))     # about 5 characters per second
        t = time.perf_counter()
        tr = al.get_trellis(em, tokens, 0)
        path = al.backtrack(tr, em, tokens, 0)
        host_ms = (time.perf_counter() - t) * 1e3
        out[f"{sec}s"] = {"emissions_ms": round(ms, 2), "audio_s_per_s": round(sec / ms * 1e3, 1), "frames": int(em.shape[0]),
                          "host_trellis_walk_ms": round(host_ms, 2), "path_points": len(path)}
    a.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
