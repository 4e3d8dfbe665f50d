"""Chunk scheduler: fixed 30 s windows, BATCH_SIZE chunks per engine call, several engines in flight.

The reference hands the whole recording to ``model.transcribe(audio, batch_size=BATCH_SIZE)``
(transcribe.py:123) and whisperx cuts it into <= 30 s windows internally; here the windows are
fixed 30 s chunks (the last one zero-padded) and a segment's start/end are its chunk's bounds.
"""
from __future__ import annotations

import os
import queue
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
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

    def __init__(self, engine, n_engines: int = 1, beam_size: int = 1, patience: float = 1.0, fallback=None, detokenize=None,
                 fork_batch: Optional[int] = None, fork_cache: Optional[int] = None):
        """``fork_batch`` (with ``fork_cache``): every engine of the pool is a fork of `engine` sized for that many
        sequences with a cross-attention K/V cache for ``fork_cache`` chunks (several hypotheses per chunk share the
        chunk's cache), and `engine` itself only lends its weights."""
        if n_engines < 1:
            raise ValueError("n_engines must be >= 1")
        self.beam_size, self.patience = int(beam_size), float(patience)
        # fallback.FallbackOptions with more than one temperature: every pass walks the temperatures for the windows
        # whose text fails the tests (needs `detokenize`: ids -> text)
        self.fallback = fallback if (fallback is not None and fallback.enabled) else None
        self.detokenize = detokenize
        if self.fallback is not None and detokenize is None:
            raise ValueError("temperature fallback needs detokenize")
        _lib.check_hw_queues(n_engines)              # one stream per engine
        self.root = engine
        if fork_batch is not None:
            self.engines = [engine.fork(int(fork_batch), int(fork_cache or fork_batch)) for _ in range(n_engines)]
        else:
            self.engines = [engine] + [engine.fork() for _ in range(n_engines - 1)]
        # engines that share the GPU: see GroupedPool (set either way: a fork inherits what its parent was last told)
        shared = "1" if n_engines > 1 else "0"
        for e in self.engines:
            e.set_option("xattn_stream_per_cu", int(os.environ.get("MW_XSTREAM_PER_CU", shared)))
            e.set_option("dec_gemm_fat", int(os.environ.get("MW_DEC_GEMM_FAT", shared)))
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
            if self.fallback is not None:
                from . import fallback as fb
                return fb.transcribe_batch(eng, pcm, self.detokenize, self.fallback)[:2]
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
        self.root.close()


def group_sizes(n_batches: int, group: int) -> List[int]:
    """n_batches consecutive batches -> sizes of the decode groups: as few groups as `group` per group allows,
    evened out so that no group is more than one batch larger than another."""
    if n_batches <= 0:
        return []
    n_groups = (n_batches + group - 1) // group
    base, extra = divmod(n_batches, n_groups)
    return [base + (1 if g < extra else 0) for g in range(n_groups)]


class GroupedPool:
    """Encoder passes of BATCH_SIZE chunks, decoding in groups of several such batches.

    The reference's ``model.transcribe(audio, batch_size=BATCH_SIZE)`` (transcribe.py:123) sends BATCH_SIZE chunks
    through the model at a time.  Here the encoder half of a pass (log-mel, encoder, cross K/V) keeps that
    granularity, on `n_encoders` engines of `batch` chunks each, but writes its K/V straight into the cache of one of
    `n_decoders` decode engines of `group * batch` sequences; a decode engine runs the greedy loop over a whole
    group at once, so every decoder weight is streamed from HBM once per step for `group` batches instead of once
    per batch, and the decode launches (latency-bound at 16 sequences) carry `group` times the work.  While one
    group decodes, the encoders fill the other decode engine's cache.  Chunks are independent: the ids are those
    of batch-by-batch passes.
    """

    def __init__(self, engine, batch: int = 16, group: int = 4, n_encoders: int = 2, n_decoders: int = 2):
        if batch < 1 or group < 1 or n_encoders < 1 or n_decoders < 1:
            raise ValueError("batch, group, n_encoders and n_decoders must be >= 1")
        if batch * group > 512:
            raise ValueError("a decode group holds at most 512 sequences")
        self.batch, self.group = int(batch), int(group)
        self.root = engine
        # the engine that owns the weights serves as the first encoder when its own workspaces are large enough
        first = [engine] if engine.max_batch >= batch else []
        _lib.check_hw_queues(n_encoders + n_decoders + (0 if first else 1))       # one stream per engine, the root's included
        self.encoders = first + [engine.fork(batch) for _ in range(n_encoders - len(first))]
        self.decoders = [engine.fork(batch * group) for _ in range(n_decoders)]
        # engines that share the GPU: one resident workgroup per CU for the decode cross-attention stream of a group
        # (mw_set_option "xattn_stream_per_cu"); MW_XSTREAM_PER_CU overrides (0 = as many as the registers allow)
        # (a lone decode engine keeps the faster form: only the encoders run beside it, and not for long)
        per_cu = int(os.environ.get("MW_XSTREAM_PER_CU", "1" if n_decoders > 1 else "0"))
        # and their projections as fewer, larger workgroups ("dec_gemm_fat"; MW_DEC_GEMM_FAT overrides)
        fat = int(os.environ.get("MW_DEC_GEMM_FAT", "1" if n_decoders > 1 else "0"))
        for e in self.decoders:
            e.set_option("xattn_stream_per_cu", per_cu)
            e.set_option("dec_gemm_fat", fat)
        self._enc_pool = ThreadPoolExecutor(max_workers=n_encoders)
        self._dec_pool = ThreadPoolExecutor(max_workers=n_decoders)
        self._free_enc: "queue.Queue" = queue.Queue()
        for e in self.encoders:
            self._free_enc.put(e)
        # development aid: (kind, index, start, end) of every encoder pass and decode loop of the last map(), host clock
        self.trace: Optional[list] = [] if os.environ.get("MW_POOL_TRACE") else None

    @property
    def engines(self):
        return self.encoders + self.decoders

    def set_decode_options(self, opts):
        for e in {id(e): e for e in [self.root] + self.engines}.values():
            e.set_decode_options(opts)

    def map(self, batches: Sequence, sizes: Optional[Sequence[int]] = None) -> list:
        """Transcribe every batch (numpy on the host or a tensor on the engine's GPU, at most `batch` chunks each);
        returns [(ids, lens)] in the order of ``batches``.  ``sizes``: batches per decode group, in order (each at
        most `group`, summing to ``len(batches)``); default: `group_sizes`."""
        n = len(batches)
        if self.trace is not None:
            self.trace.clear()
        if sizes is None:
            sizes = group_sizes(n, self.group)
        sizes = [int(x) for x in sizes]
        if sum(sizes) != n or any(x < 1 or x > self.group for x in sizes):
            raise ValueError(f"group sizes {sizes} do not split {n} batches into groups of 1..{self.group}")
        n_groups, nd = len(sizes), len(self.decoders)
        group_of, slot_of, members = [], [], []
        i = 0
        for g, sz in enumerate(sizes):
            slot = 0
            members.append(list(range(i, i + sz)))
            for _ in range(sz):
                rows = int(batches[i].shape[0])
                if rows > self.batch:
                    raise ValueError(f"batch {i} has {rows} chunks; the encoders take {self.batch}")
                group_of.append(g)
                slot_of.append(slot)
                slot += rows
                i += 1
        encoded = [threading.Event() for _ in range(n)]
        cache_free = [threading.Event() for _ in range(n_groups)]      # group g's decode engine has finished group g - nd
        for g in range(min(nd, n_groups)):
            cache_free[g].set()
        results: List[Optional[tuple]] = [None] * n
        errors: list = []

        def encode(i):
            g = group_of[i]
            try:
                cache_free[g].wait()
                if errors:
                    return
                enc = self._free_enc.get()
                t_start = time.perf_counter()
                try:
                    enc.encode_into(batches[i], self.decoders[g % nd], slot_of[i])
                finally:
                    self._free_enc.put(enc)
                    if self.trace is not None:
                        self.trace.append(("encode", i, t_start, time.perf_counter()))
            except BaseException as e:         # noqa: BLE001 - handed to the caller below
                errors.append(e)
            finally:
                encoded[i].set()

        def decode(g):
            dec = self.decoders[g % nd]
            try:
                for i in members[g]:
                    encoded[i].wait()
                if not errors:
                    rows = sum(int(batches[i].shape[0]) for i in members[g])
                    dec.set_encoded(rows)
                    t_start = time.perf_counter()
                    ids, lens, _ = dec.decode_greedy(rows)
                    if self.trace is not None:
                        self.trace.append(("decode", g, t_start, time.perf_counter()))
                    for i in members[g]:
                        a, b = slot_of[i], slot_of[i] + int(batches[i].shape[0])
                        results[i] = (ids[a:b].copy(), lens[a:b].copy())
            except BaseException as e:         # noqa: BLE001
                errors.append(e)
            finally:
                if g + nd < n_groups:
                    cache_free[g + nd].set()

        enc_f = [self._enc_pool.submit(encode, i) for i in range(n)]
        dec_f = [self._dec_pool.submit(decode, g) for g in range(n_groups)]
        for f in enc_f + dec_f:
            f.result()
        if errors:
            raise errors[0]
        return results

    def close(self):
        self._enc_pool.shutdown(wait=True)
        self._dec_pool.shutdown(wait=True)
        for e in reversed(self.engines):
            if e is not self.root:
                e.close()
        self.root.close()
