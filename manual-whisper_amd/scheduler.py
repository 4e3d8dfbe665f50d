"""Chunk scheduler: fixed 30 s windows, BATCH_SIZE chunks per engine call, several engines in flight.

The reference hands the whole recording to ``model.transcribe(audio, batch_size=BATCH_SIZE)``
(transcribe.py:123) and whisperx cuts it into <= 30 s windows internally; here the windows are
fixed 30 s chunks (the last one zero-padded) and a segment's start/end are its chunk's bounds.
"""
from __future__ import annotations

import queue
from concurrent.futures import ThreadPoolExecutor
from typing import List, Sequence, Tuple

import numpy as np

from .dims import CHUNK_SAMPLES, CHUNK_SECONDS, SAMPLE_RATE


def split_chunks(audio: np.ndarray) -> Tuple[np.ndarray, List[Tuple[float, float]]]:
    """float32 mono 16 kHz [n] -> ([n_chunks, 480000], [(start_s, end_s)]); empty audio -> 0 chunks."""
    audio = np.ascontiguousarray(audio, dtype=np.float32).reshape(-1)
    n = audio.shape[0]
    n_chunks = (n + CHUNK_SAMPLES - 1) // CHUNK_SAMPLES
    out = np.zeros((n_chunks, CHUNK_SAMPLES), np.float32)
    spans = []
    for i in range(n_chunks):
        a, b = i * CHUNK_SAMPLES, min(n, (i + 1) * CHUNK_SAMPLES)
        out[i, : b - a] = audio[a:b]
        spans.append((i * float(CHUNK_SECONDS), b / float(SAMPLE_RATE)))
    return out, spans


def batches_of(n_items: int, batch_size: int) -> List[Tuple[int, int]]:
    if batch_size < 1:
        raise ValueError("batch_size must be >= 1")
    return [(a, min(n_items, a + batch_size)) for a in range(0, n_items, batch_size)]


class EnginePool:
    """Round-robins batches over an engine and its forks, one host thread per engine.

    At batch 16 a decode step is a chain of ~260 short, latency-bound launches that leave most of
    the chip idle; a second and third engine (own stream and caches, shared weights) fill those gaps.
    """

    def __init__(self, engine, n_engines: int = 1, beam_size: int = 1, patience: float = 1.0):
        if n_engines < 1:
            raise ValueError("n_engines must be >= 1")
        self.beam_size, self.patience = int(beam_size), float(patience)
        self.engines = [engine] + [engine.fork() for _ in range(n_engines - 1)]
        self._free: "queue.Queue" = queue.Queue()
        for e in self.engines:
            self._free.put(e)
        self._pool = ThreadPoolExecutor(max_workers=n_engines)

    def set_decode_options(self, opts):
        for e in self.engines:
            e.set_decode_options(opts)

    def _run(self, pcm):
        eng = self._free.get()
        try:
            if self.beam_size > 1:
                return eng.transcribe_batch_beam(pcm, self.beam_size, self.patience)
            return eng.transcribe_batch(pcm)
        finally:
            self._free.put(eng)

    def map(self, batches: Sequence) -> list:
        """Transcribe every batch (numpy on the host or a tensor on the engine's GPU); results keep
        the order of ``batches``."""
        return list(self._pool.map(self._run, batches))

    def close(self):
        self._pool.shutdown(wait=True)
        for e in reversed(self.engines):
            e.close()
