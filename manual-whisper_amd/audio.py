"""Host-side audio decoding: ``load_audio(path) -> float32 mono 16 kHz`` (reference transcribe.py:117).

whisperx shells out to ffmpeg (s16le, 16 kHz, mono, /32768).  Audio decoding stays on the host
here too: WAV files are parsed directly (the container has no ffmpeg); anything else goes through
an ``ffmpeg`` binary when one is on PATH and raises otherwise.
"""
from __future__ import annotations

import shutil
import struct
import subprocess

import numpy as np

from .dims import SAMPLE_RATE


def _read_wav(path: str):
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] not in (b"RIFF", b"RF64") or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            tag, ch, rate, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and len(body) >= 26:             # WAVE_FORMAT_EXTENSIBLE: real tag in the GUID
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, ch, rate, bits)
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    tag, ch, rate, bits = fmt
    if tag == 1 and bits == 16:
        x = np.frombuffer(pcm[: len(pcm) // 2 * 2], "<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 8:
        x = (np.frombuffer(pcm, np.uint8).astype(np.float32) - 128.0) / 128.0
    elif tag == 1 and bits == 24:
        b = np.frombuffer(pcm[: len(pcm) // 3 * 3], np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
    elif tag == 1 and bits == 32:
        x = np.frombuffer(pcm[: len(pcm) // 4 * 4], "<i4").astype(np.float32) / 2147483648.0
    elif tag == 3 and bits == 32:
        x = np.frombuffer(pcm[: len(pcm) // 4 * 4], "<f4").astype(np.float32)
    elif tag == 3 and bits == 64:
        x = np.frombuffer(pcm[: len(pcm) // 8 * 8], "<f8").astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported WAV encoding (format tag {tag}, {bits} bits)")
    if ch > 1:
        x = x[: len(x) // ch * ch].reshape(-1, ch).mean(axis=1, dtype=np.float32)
    return x, rate


def _resample(x: np.ndarray, rate: int) -> np.ndarray:
    if rate == SAMPLE_RATE or x.size == 0:
        return x.astype(np.float32)
    from math import gcd
    g = gcd(rate, SAMPLE_RATE)
    try:
        from scipy.signal import resample_poly
        return resample_poly(x.astype(np.float64), SAMPLE_RATE // g, rate // g).astype(np.float32)
    except ImportError:                                         # linear interpolation as a last resort
        n_out = int(round(len(x) * SAMPLE_RATE / rate))
        t = np.arange(n_out, dtype=np.float64) * rate / SAMPLE_RATE
        return np.interp(t, np.arange(len(x)), x).astype(np.float32)


def load_audio(file: str, sr: int = SAMPLE_RATE) -> np.ndarray:
    """Decode ``file`` to float32 mono at 16 kHz in [-1, 1)."""
    if sr != SAMPLE_RATE:
        raise ValueError("the engine works at 16 kHz only")
    with open(file, "rb") as f:
        head = f.read(12)
    if head[:4] in (b"RIFF", b"RF64") and head[8:12] == b"WAVE":
        x, rate = _read_wav(file)
        return np.ascontiguousarray(_resample(x, rate))
    ffmpeg = shutil.which("ffmpeg")
    if ffmpeg is None:
        raise RuntimeError(f"{file}: only WAV can be read without ffmpeg, and no ffmpeg binary is on PATH")
    cmd = [ffmpeg, "-nostdin", "-threads", "0", "-i", file, "-f", "s16le", "-ac", "1", "-acodec", "pcm_s16le",
           "-ar", str(SAMPLE_RATE), "-"]
    out = subprocess.run(cmd, capture_output=True, check=True).stdout
    return np.frombuffer(out, np.int16).astype(np.float32) / 32768.0


def write_wav(path: str, x: np.ndarray, rate: int = SAMPLE_RATE):
    """16-bit mono WAV writer (tests and examples)."""
    pcm = np.clip(np.round(np.asarray(x, np.float64) * 32768.0), -32768, 32767).astype("<i2").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, rate, rate * 2, 2, 16))
        f.write(b"data" + struct.pack("<I", len(pcm)) + pcm)
