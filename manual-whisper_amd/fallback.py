"""Temperature fallback: decode again, warmer, the windows whose text looks degenerate.

The reference leaves decoding to whisperx (transcribe.py:107-113 passes only ``initial_prompt`` in
``asr_options``); whisperx's option set carries ``temperatures = (0.0, 0.2, 0.4, 0.6, 0.8, 1.0)``, ``best_of = 5``,
``compression_ratio_threshold = 2.4`` and ``log_prob_threshold = -1.0``, the arguments of OpenAI Whisper's
``decode_with_fallback`` (transcribe.py of the openai-whisper package), which faster-whisper mirrors as
``generate_with_fallback``.  None of that code is in the reference tree or in this container: what follows
restates the published algorithm, and parity with the reference is unpinned.  From memory of upstream, whisperx's
*batched* pipeline decodes every window once at ``temperatures[0]`` and never walks the list, so ``load_model``
turns this loop on only when ``asr_options`` names more than one temperature.

The algorithm, per window:
  * decode at ``temperatures[0]`` (0: beam search when ``beam_size > 1``, else the arg-max);
  * the result fails when zlib's compression ratio of its UTF-8 text exceeds ``compression_ratio_threshold``
    (repetition loops compress well) or when its average log-probability, sum / (tokens + 1), is below
    ``log_prob_threshold``; a threshold of ``None`` switches its test off;
  * a failed window is decoded again at the next temperature: ``best_of`` samples, the one with the highest
    log-probability per token is kept; the first result that passes is final;
  * when every temperature fails, ``final="best"`` (faster-whisper) takes, among the results whose compression ratio
    is below the threshold (all of them if there is none), the one with the highest average log-probability;
    ``final="last"`` (OpenAI Whisper) takes the last.
``no_speech_threshold`` has no part here: whisperx cuts its windows with a voice-activity detector and does not use it.

On the GPU a pass over ``n`` windows keeps their encoder state; every further temperature is one
``Engine.decode_sample`` call in which only the failed windows are active: a window that has passed starts ended, so
its cross-attention (the K/V stream, the largest share of a step's bytes) is skipped, but its rows still ride through
the prompt prefill, the self-attention, the projections (whose weights stream once for all rows anyway) and the
logits of every step; a change of ``best_of`` between temperatures (1 at temperature 0, 5 above) re-captures the
step graphs once.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

DEFAULT_TEMPERATURES = (0.0, 0.2, 0.4, 0.6, 0.8, 1.0)


def compression_ratio(text: str) -> float:
    """len(utf-8 bytes) / len(zlib.compress(bytes)), the measure of both upstream implementations."""
    raw = text.encode("utf-8")
    return len(raw) / len(zlib.compress(raw))


@dataclass(frozen=True)
class FallbackOptions:
    temperatures: Tuple[float, ...] = DEFAULT_TEMPERATURES
    compression_ratio_threshold: Optional[float] = 2.4
    log_prob_threshold: Optional[float] = -1.0
    best_of: int = 5
    beam_size: int = 1
    patience: float = 1.0
    seed: int = 0
    final: str = "best"

    def __post_init__(self):
        t = tuple(float(x) for x in (self.temperatures if isinstance(self.temperatures, (tuple, list)) else (self.temperatures,)))
        if not t or any(x < 0 for x in t):
            raise ValueError("temperatures: a non-empty list of values >= 0")
        object.__setattr__(self, "temperatures", t)
        if self.best_of < 1 or self.beam_size < 1:
            raise ValueError("best_of and beam_size must be >= 1")
        if self.final not in ("best", "last"):
            raise ValueError("final: 'best' or 'last'")

    @property
    def enabled(self) -> bool:
        return len(self.temperatures) > 1

    def rows_per_window(self) -> int:
        """Sequences a window occupies in its widest decode."""
        return max(self.beam_size if 0.0 in self.temperatures else 1, self.best_of if any(t > 0 for t in self.temperatures) else 1)

    def fails(self, ratio: float, avg_logprob: float) -> bool:
        if self.compression_ratio_threshold is not None and ratio > self.compression_ratio_threshold:
            return True
        if self.log_prob_threshold is not None and avg_logprob < self.log_prob_threshold:
            return True
        return False

    @classmethod
    def from_asr_options(cls, asr_options: dict) -> "FallbackOptions":
        o = dict(asr_options or {})
        t = o.get("temperatures", (0.0,))
        return cls(temperatures=tuple(t) if isinstance(t, (tuple, list)) else (float(t),),
                   compression_ratio_threshold=o.get("compression_ratio_threshold", 2.4),
                   log_prob_threshold=o.get("log_prob_threshold", -1.0),
                   best_of=int(o.get("best_of", 5)), beam_size=int(o.get("beam_size", 1)), patience=float(o.get("patience", 1.0)),
                   seed=int(o.get("seed", 0)), final=o.get("fallback_final", "best"))


@dataclass
class WindowResult:
    ids: np.ndarray
    avg_logprob: float
    compression_ratio: float
    temperature: float
    tried: List[Tuple[float, float, float]] = field(default_factory=list)     # (temperature, avg_logprob, ratio) of every attempt


def walk_temperatures(decode_at: Callable[[float, np.ndarray], Tuple[Sequence[np.ndarray], Sequence[float]]], n: int,
                      detokenize: Callable[[Sequence[int]], str], o: FallbackOptions) -> List[WindowResult]:
    """The loop itself, apart from any engine: ``decode_at(temperature, active)`` decodes the windows whose ``active``
    flag is set and returns (ids per window, avg_logprob per window); entries of inactive windows are ignored."""
    attempts: List[List[WindowResult]] = [[] for _ in range(n)]
    final: List[Optional[WindowResult]] = [None] * n
    active = np.ones(n, bool)
    for t in o.temperatures:
        if not active.any():
            break
        ids, avg = decode_at(float(t), active.copy())
        for i in np.flatnonzero(active):
            toks = np.asarray(ids[i], np.int32)
            ratio = compression_ratio(detokenize(toks))
            res = WindowResult(toks, float(avg[i]), ratio, float(t))
            attempts[i].append(res)
            if not o.fails(ratio, res.avg_logprob):
                final[i] = res
                active[i] = False
    for i in range(n):
        if final[i] is None:                                    # every temperature failed
            if o.final == "last":
                final[i] = attempts[i][-1]
            else:
                thr = o.compression_ratio_threshold
                ok = [a for a in attempts[i] if thr is None or a.compression_ratio < thr]
                final[i] = max(ok or attempts[i], key=lambda a: a.avg_logprob)     # ties: the earliest
        final[i].tried = [(a.temperature, a.avg_logprob, a.compression_ratio) for a in attempts[i]]
    return final


def decode_with_fallback(engine, n_audio: int, detokenize: Callable[[Sequence[int]], str], o: FallbackOptions) -> List[WindowResult]:
    """Over the encoder state of ``engine``'s last ``encode(n_audio)``."""
    if n_audio * o.rows_per_window() > engine.max_batch:
        raise ValueError(f"{n_audio} windows x {o.rows_per_window()} sequences do not fit the engine's max_batch={engine.max_batch}")

    def decode_at(t: float, active: np.ndarray):
        if t == 0.0 and o.beam_size > 1:                        # beam search has no per-window switch: the first pass decodes all
            ids, lens, avg = engine.decode_beam(n_audio, o.beam_size, o.patience)
        else:                                                   # every temperature draws from a stream of its own
            seed = (o.seed + 0x9E3779B9 * o.temperatures.index(t)) & 0xFFFFFFFF
            ids, lens, avg = engine.decode_sample(n_audio, t, best_of=o.best_of if t > 0 else 1, seed=seed, active=active)
        return [ids[i, : lens[i]] for i in range(n_audio)], avg

    return walk_temperatures(decode_at, n_audio, detokenize, o)


def transcribe_batch(engine, pcm, detokenize, o: FallbackOptions):
    """log-mel -> encoder -> decode with fallback for one pass of chunks: -> (ids [n, max_new] padded with
    end-of-text, lens, list of WindowResult), the first two shaped as ``Engine.transcribe_batch`` returns them."""
    n = int(pcm.shape[0])
    engine.logmel(pcm, want_output=False)
    engine.encode(n, want_output=False)
    res = decode_with_fallback(engine, n, detokenize, o)
    ids = np.full((n, engine.opts.max_new), engine.opts.eot, np.int32)
    lens = np.zeros(n, np.int32)
    for i, r in enumerate(res):
        ids[i, : len(r.ids)] = r.ids
        lens[i] = len(r.ids)
    return ids, lens, res
