"""Development aid: is the encoder's output for a recording the same bits whatever else shares the pass, and how
long does the encoder stage take?   python tools/enc_invariance.py [model]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import manual_whisper_amd as mw
from manual_whisper_amd import synth
from manual_whisper_amd.engine import Engine

name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
dims = mw.get_dims(name)
nb = 3 if name != "large-v3" else 16
eng = Engine(dims, 0, nb)
eng.load_weights(synth.make_weights_np(dims, 1234))
pcm = synth.make_pcm_np(nb, seed=1)
mel = eng.logmel(pcm)
a = eng.encode(nb, mel=mel)
for rep in range(2):
    for i in range(min(nb, 3)):
        b = eng.encode(1, mel=mel[i:i + 1])
        print(name, "rep", rep, "row", i, "identical" if np.array_equal(a[i], b[0]) else f"DIFFERS max {np.abs(a[i] - b[0]).max():.3e}", flush=True)
        if rep == 0 and not np.array_equal(a[i], b[0]):
            d = np.abs(a[i] - b[0])
            t_bad = np.nonzero(d.max(axis=1) > 0)[0]
            print("   frames that differ:", len(t_bad), "first", t_bad[:8], "last", t_bad[-8:], "columns that differ:", int((d.max(axis=0) > 0).sum()),
                  "mean abs diff", float(d.mean()), "scale", float(np.abs(b[0]).mean()))
a2 = eng.encode(nb, mel=mel)
print("same pass twice:", "identical" if np.array_equal(a, a2) else f"DIFFERS max {np.abs(a - a2).max():.3e}")
print("encoder stage ms", eng.stage_ms()["encoder"])
eng.close()
