"""Speech-activity windows: what whisperx does between ``load_audio`` and the model.

The reference hands ``vad_options={"vad_onset": 0.5, "vad_offset": 0.363}`` to ``whisperx.load_model``
(transcribe.py:43-46,112).  whisperx (3.7.6, not in the reference tree and not installable here)
then runs a neural segmentation model over the recording, binarises its per-frame speech score
with onset/offset hysteresis, and merges the speech regions into windows of at most 30 s; each
window is one unit of the hot path and its bounds are the segment's ``start``/``end``.

Built here:
  * the per-frame score comes from an ENERGY detector (frame log-energy on the GPU,
    ``mw_frame_energy_db``), because the segmentation model and its weights do not exist offline;
  * ``binarize`` and ``merge_chunks`` restate whisperx's hysteresis and merge rules from the
    package's published behaviour (its vad.py ``Binarize.__call__`` with ``max_duration`` and
    ``merge_chunks``).  Neither the reference nor its tests hold vectors for them: parity with
    whisperx is unpinned; the tests pin hand-computed cases.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from .dims import CHUNK_SAMPLES, CHUNK_SECONDS, SAMPLE_RATE

FRAME = 400          # 25 ms, the feature frame
HOP = 160            # 10 ms: one score per log-mel frame
DEFAULT_ONSET = 0.5
DEFAULT_OFFSET = 0.363


def speech_score(energy_db: np.ndarray, smooth: int = 15, margin_db: float = 9.0, slope_db: float = 2.0,
                 silence_db: float = -55.0) -> np.ndarray:
    """Frame log-energies (dB) -> speech score in [0, 1].

    The noise floor is the 10th percentile of the recording's frame energies; a frame `margin_db`
    above it scores 0.5 and the score is a logistic in dB, then a moving average over `smooth`
    frames (150 ms) removes single-frame chatter.  A recording without dynamics (floor within 6 dB
    of the 95th percentile) is all speech if louder than `silence_db`, else all silence.
    """
    e = np.asarray(energy_db, np.float32)
    if e.size == 0:
        return e.copy()
    floor = float(np.percentile(e, 10))
    high = float(np.percentile(e, 95))
    if high - floor < 6.0:
        return np.full(e.shape, 1.0 if high > silence_db else 0.0, np.float32)
    s = 1.0 / (1.0 + np.exp(-(e - (floor + margin_db)) / slope_db))
    if smooth > 1:
        k = np.ones(smooth, np.float32) / smooth
        s = np.convolve(np.pad(s, (smooth // 2, smooth - 1 - smooth // 2), mode="edge"), k, mode="valid")
    return s.astype(np.float32)


def binarize(scores: Sequence[float], onset: float = DEFAULT_ONSET, offset: float = DEFAULT_OFFSET,
             hop_s: float = HOP / SAMPLE_RATE, max_duration: float = float(CHUNK_SECONDS)) -> List[Tuple[float, float]]:
    """Hysteresis thresholding of a score track -> speech regions [(start_s, end_s)].

    A region opens at the first frame whose score exceeds `onset` and closes at the first frame
    whose score falls below `offset`.  A region that grows past `max_duration` is cut at the
    lowest-scoring frame of its second half, and the rest carries on as a new region.  Frame i is
    stamped at its middle, (i + 0.5) * hop_s.
    """
    y = np.asarray(scores, np.float64)
    n = len(y)
    if n == 0:
        return []
    ts = (np.arange(n) + 0.5) * hop_s
    regions: List[Tuple[float, float]] = []
    start = ts[0]
    active = bool(y[0] > onset)
    cur_y: List[float] = [float(y[0])]
    cur_t: List[float] = [float(ts[0])]
    t = float(ts[0])
    for i in range(1, n):
        t, v = float(ts[i]), float(y[i])
        if active:
            if t - start > max_duration:
                after = len(cur_y) // 2
                cut = after + int(np.argmin(cur_y[after:]))
                regions.append((float(start), cur_t[cut]))
                start = cur_t[cut]
                cur_y, cur_t = cur_y[cut + 1:], cur_t[cut + 1:]
            elif v < offset:
                regions.append((float(start), t))
                start, active = t, False
                cur_y, cur_t = [], []
            cur_y.append(v)
            cur_t.append(t)
        elif v > onset:
            start, active = t, True
            cur_y, cur_t = [v], [t]
    if active:
        regions.append((float(start), t))
    return regions


def merge_chunks(regions: Sequence[Tuple[float, float]], chunk_size: float = float(CHUNK_SECONDS)) -> List[dict]:
    """Consecutive speech regions -> windows of at most `chunk_size` seconds.

    Regions are appended to the current window while the window stays within `chunk_size`; the
    window then spans from its first region's start to its last region's end (silence inside is
    kept, silence between windows is dropped).
    """
    regions = [(float(a), float(b)) for a, b in regions]
    if not regions:
        return []
    out: List[dict] = []
    cur_start, cur_end = regions[0][0], 0.0
    inside: List[Tuple[float, float]] = []
    for a, b in regions:
        if b - cur_start > chunk_size and cur_end - cur_start > 0:
            out.append({"start": cur_start, "end": cur_end, "segments": inside})
            cur_start, inside = a, []
        cur_end = b
        inside.append((a, b))
    out.append({"start": cur_start, "end": cur_end, "segments": inside})
    return out


def cut_windows(audio: np.ndarray, windows: Sequence[dict]) -> Tuple[np.ndarray, List[Tuple[float, float]]]:
    """Windows -> ([n, 480000] zero-padded chunks, [(start_s, end_s)]); a window longer than 30 s is clipped."""
    audio = np.ascontiguousarray(audio, dtype=np.float32).reshape(-1)
    out = np.zeros((len(windows), CHUNK_SAMPLES), np.float32)
    spans = []
    for i, w in enumerate(windows):
        a = max(0, int(w["start"] * SAMPLE_RATE))
        b = min(len(audio), int(w["end"] * SAMPLE_RATE), a + CHUNK_SAMPLES)
        out[i, : b - a] = audio[a:b]
        spans.append((round(float(w["start"]), 3), round(float(w["end"]), 3)))
    return out, spans


def speech_windows(engine, audio: np.ndarray, vad_options: dict) -> List[dict]:
    """The whole front end: energies on the GPU, score, hysteresis, merge."""
    energy = engine.frame_energy_db(audio, FRAME, HOP)
    score = speech_score(energy)
    regions = binarize(score, float(vad_options.get("vad_onset", DEFAULT_ONSET)),
                       float(vad_options.get("vad_offset", DEFAULT_OFFSET)))
    return merge_chunks(regions, float(vad_options.get("chunk_size", CHUNK_SECONDS)))
