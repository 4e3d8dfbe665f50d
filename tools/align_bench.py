#!/usr/bin/env python3
"""Time the forced-alignment acoustic model (wav2vec2-large-xlsr-53 geometry, synthetic weights) on one GPU:
ms per mw_align_emissions call for segments of 5, 15 and 30 s, the host search (mw_align_path) beside it, and align()
over 48 windows of 30 s in packs of MW_ALIGN_PACK seconds (default 240: eight windows per pass; 30: one window per
pass, the earlier arrangement) with MW_ALIGN_STREAMS (default 4) passes in flight."""
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
    a = al.Aligner(cfg, device=0, max_seconds=30, n_streams=int(os.environ.get("MW_ALIGN_STREAMS", "4")),
                   pack_seconds=int(os.environ.get("MW_ALIGN_PACK", "240")))
    a.load_weights(w)
    out = {"n_streams": a.n_streams, "pack_seconds": a.desc.max_samples // 16000}
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
        path = al.ctc_path(em, tokens, 0)
        host_ms = (time.perf_counter() - t) * 1e3
        out[f"{sec}s"] = {"emissions_ms": round(ms, 2), "audio_s_per_s": round(sec / ms * 1e3, 1), "frames": int(em.shape[0]),
                          "host_path_ms": round(host_ms, 2), "path_points": len(path)}
    # a recording's worth of segments through align(): 48 windows of 30 s, 150 characters each
    chars = [c for c in localckpt.synthetic_ctc_vocab(cfg["vocab_size"]) if len(c) == 1 and ord(c) >= 0x4E00]
    meta = {"language": "zh", "dictionary": localckpt.synthetic_ctc_vocab(cfg["vocab_size"]), "type": "huggingface"}
    n_seg = 48
    audio = np.tile(synth.hash_uniform_np(30 * 16000, "wave.bench", 5) * np.float32(0.6), n_seg)
    rng = np.random.default_rng(1)
    segs = [{"text": "".join(rng.choice(chars, 150)), "start": 30.0 * i, "end": 30.0 * (i + 1)} for i in range(n_seg)]
    al.align(segs[:24], a, meta, audio, "cuda")
    t = time.perf_counter()
    res = al.align(segs, a, meta, audio, "cuda")
    dt = time.perf_counter() - t
    out["align_48x30s"] = {"seconds": round(dt, 3), "audio_s_per_s": round(n_seg * 30 / dt, 1), "words": len(res["word_segments"])}
    a.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
