"""Developer probe: teacher-forced logits of a 64- or 100-sequence pass against the same sequences in passes of 16
(the two take different projection kernels): max |difference| per weight format."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import manual_whisper_amd as mw
from manual_whisper_amd import synth
from manual_whisper_amd.decode_options import synthetic_options
from manual_whisper_amd.engine import Engine

name = sys.argv[1] if len(sys.argv) > 1 else "micro"
for wf in ("f16", "fp8"):
    dims = mw.get_dims(name)
    w = synth.make_weights_np(dims, 4321)
    eng = Engine(dims, 0, 128, weight_format=wf)
    eng.load_weights(w)
    opts = synthetic_options(dims, max_new=6)
    eng.set_decode_options(opts)
    for n in (64, 100):
        pcm = synth.make_pcm_np(n, 23)
        rng = np.random.default_rng(5)
        toks = np.concatenate([np.tile(np.asarray(opts.prompt)[None], (n, 1)), rng.integers(0, dims.vocab, size=(n, 5))], axis=1).astype(np.int32)
        eng.logmel(pcm, want_output=False); eng.encode(n, want_output=False)
        big = eng.decode_forced(toks)
        parts = []
        for a in range(0, n, 16):
            eng.logmel(pcm[a:a + 16], want_output=False); eng.encode(len(pcm[a:a + 16]), want_output=False)
            parts.append(eng.decode_forced(toks[a:a + 16]))
        small = np.concatenate(parts)
        d = np.abs(big - small)
        print(f"{name} {wf} n={n}: max |logit difference| {d.max():.3e} (row {int(d.reshape(n, -1).max(axis=1).argmax())}), logit std {small.std():.2f}", flush=True)
    eng.close()
