#!/usr/bin/env python3
"""Golden vectors for the forced-alignment acoustic model: transformers' Wav2Vec2ForCTC (the installed library,
THIS container only) on the deterministic synthetic w2v-micro checkpoint -> tests/golden/w2v_micro.npz.
Inputs are regenerated from seeds by the tests; only the expected outputs are stored."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manual_whisper_amd  # noqa: F401,E402
from manual_whisper_amd import localckpt, synth  # noqa: E402
from transformers import Wav2Vec2Config, Wav2Vec2ForCTC  # noqa: E402

SEED, WAVE_SEED = 4242, 9
cfg = localckpt.wav2vec2_config("w2v-micro")
hf_cfg = Wav2Vec2Config(**{k: v for k, v in cfg.items() if k not in ("model_type", "architectures")},
                        hidden_dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, feat_proj_dropout=0.0,
                        final_dropout=0.0, layerdrop=0.0, mask_time_prob=0.0)
model = Wav2Vec2ForCTC(hf_cfg).eval()
w = localckpt.make_wav2vec2_weights_np(cfg, SEED)
missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
assert not unexpected and all("masked_spec_embed" in m for m in missing), (missing, unexpected)
out = {}
for tag, n in (("a", 19200), ("b", 6000), ("c", 400)):          # 1.2 s, 0.375 s, the shortest input (one frame)
    wave = synth.hash_uniform_np(n, f"wave.{tag}", WAVE_SEED) * np.float32(0.6)
    with torch.no_grad():
        logits = model(torch.from_numpy(wave)[None]).logits[0]
    out[f"logits_{tag}"] = logits.numpy().astype(np.float32)
    out[f"emissions_{tag}"] = torch.log_softmax(logits, dim=-1).numpy().astype(np.float32)
    out[f"n_{tag}"] = n
    print(tag, n, tuple(logits.shape), float(logits.std()))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "w2v_micro.npz"), seed=SEED, wave_seed=WAVE_SEED, **out)
print("wrote tests/golden/w2v_micro.npz")
