#!/usr/bin/env python3
"""Development aid: write a 16 kHz mono 16-bit WAV of noise bursts (2-25 s) separated by pauses (0.5-3 s).
Usage: python tools/make_wav.py out.wav [minutes] [seed]"""
import sys
import wave

import numpy as np

path = sys.argv[1]
minutes = float(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
n = int(minutes * 60 * 16000)
x = np.zeros(n, np.float32)
t = 0
while t < n:
    burst = int(rng.uniform(2, 25) * 16000)
    x[t:t + burst] = 0.1 * rng.standard_normal(min(burst, n - t)).astype(np.float32)
    t += burst + int(rng.uniform(0.5, 3) * 16000)
with wave.open(path, "wb") as w:
    w.setnchannels(1)
    w.setsampwidth(2)
    w.setframerate(16000)
    w.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())
print(path, n / 16000, "s")
