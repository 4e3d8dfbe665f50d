"""``load_model`` / ``Model.transcribe``: the call surface of reference transcribe.py:107-123.

    model = load_model(MODEL_SIZE, DEVICE, compute_type=..., language="zh",
                       asr_options={"initial_prompt": ...}, vad_options=...)
    result = model.transcribe(audio, batch_size=BATCH_SIZE, language="zh")
    # {"segments": [{"text", "start", "end"}, ...], "language": "zh"}

Differences that cannot be avoided offline, all stated where they bite:
  * weights come from a LOCAL directory in Hugging Face layout (config.json + model.safetensors or
    pytorch_model.bin [+ tokenizer.json]); nothing is downloaded.  ``synthetic:<size>`` builds the
    deterministic random-init checkpoint used by the tests and the benchmark.
  * windows: with ``vad_options`` (as the reference passes them) the recording is cut into speech
    windows of at most 30 s by vad.py (energy detector + whisperx's hysteresis / merge rules; the
    neural VAD model does not exist offline); ``vad_options=None`` or ``{"vad_method": "fixed"}``
    gives fixed 30 s chunks.
  * ``compute_type``: ``"float16"`` or ``"float8"`` (the reference's CPU default ``"int8"`` is a CTranslate2 format
    with no counterpart here).
  * decoding is greedy by default (what the throughput figures are quoted on); ``asr_options={"beam_size": 5}``
    switches to beam search at temperature 0 (whisperx's default beam), which runs ``max_batch // beam_size`` chunks
    per pass; ``asr_options={"temperatures": [0.0, 0.2, ...], "best_of": 5, "compression_ratio_threshold": 2.4,
    "log_prob_threshold": -1.0}`` with more than one temperature turns on the temperature fallback (fallback.py).  No timestamps; text needs tokenizer.json, otherwise a segment's text is
    its token ids written as ``<|id|>``; ``transcribe(..., return_tokens=True)`` also returns the ids under ``"tokens"``.
"""
from __future__ import annotations

import json
import os
from typing import Optional

import numpy as np

from .decode_options import DecodeOptions, synthetic_options, whisper_options
from .dims import CHUNK_SAMPLES, MODELS, SAMPLE_RATE, WhisperDims, get_dims
from .engine import Engine, EngineError
from .scheduler import EnginePool, GroupedPool, batches_of, split_chunks

_GPU_NAMES = ("cuda", "hip", "gpu", "rocm")
_FP16_NAMES = ("float16", "fp16", "half", "default")
_FP8_NAMES = ("float8", "fp8", "float8_e4m3", "float8_e4m3fn", "float8_kv8")
_KV8_NAMES = ("float8_kv8",)


def _dims_from_config(path: str) -> WhisperDims:
    with open(os.path.join(path, "config.json"), encoding="utf-8") as f:
        c = json.load(f)
    return WhisperDims(os.path.basename(os.path.normpath(path)), c["num_mel_bins"], c["d_model"],
                       c["encoder_attention_heads"], c["encoder_layers"], c["decoder_layers"],
                       c["encoder_ffn_dim"], c["vocab_size"], c.get("max_source_positions", 1500),
                       c.get("max_target_positions", 448))


def _load_state_dict(path: str) -> dict:
    """Tensors only: safetensors, or a torch checkpoint opened with weights_only=True."""
    st = os.path.join(path, "model.safetensors")
    if os.path.exists(st):
        from safetensors.numpy import load_file
        return load_file(st)
    pt = os.path.join(path, "pytorch_model.bin")
    if os.path.exists(pt):
        import torch
        return torch.load(pt, map_location="cpu", weights_only=True)
    raise FileNotFoundError(f"{path}: no model.safetensors or pytorch_model.bin")


class Model:
    def __init__(self, engine: Engine, dims: WhisperDims, language: str, tokenizer=None,
                 initial_prompt: Optional[str] = None, vad_options: Optional[dict] = None,
                 synthetic: bool = False, max_new: int = 224, n_engines: int = 1, beam_size: int = 1, patience: float = 1.0,
                 coalesce: int = 1, arrangement: str = "coalesced", group: int = 7, n_encoders: int = 2, n_decoders: int = 3,
                 fallback=None, hyp_pass_windows: Optional[int] = None):
        self.engine, self.dims, self.language = engine, dims, language
        self.tokenizer, self.initial_prompt = tokenizer, initial_prompt
        self.vad_options = dict(vad_options or {})
        self.synthetic, self.max_new = synthetic, max_new
        if beam_size < 1 or beam_size > engine.max_batch:
            raise EngineError(f"beam_size={beam_size} does not fit max_batch={engine.max_batch}")
        if arrangement not in ("grouped", "coalesced"):
            raise EngineError(f"arrangement={arrangement!r}: 'grouped' or 'coalesced'")
        self.beam_size, self.patience = int(beam_size), float(patience)
        # fallback.FallbackOptions naming more than one temperature: windows whose text fails the compression-ratio or
        # log-probability test are decoded again, warmer (fallback.py); like beam search this runs whole passes per engine
        self.fallback = fallback if (fallback is not None and fallback.enabled) else None
        if self.fallback is not None and self.fallback.rows_per_window() > engine.max_batch:
            raise EngineError(f"best_of / beam_size = {self.fallback.rows_per_window()} does not fit max_batch={engine.max_batch}")
        # "grouped" (greedy decoding only): encoder passes of `batch_size` chunks on `n_encoders` engines, the greedy
        # loop over up to `group` such batches at once on `n_decoders` engines (scheduler.GroupedPool), sized to the job
        # at the first call.  "coalesced" (and beam search): `n_engines` engines that each run whole passes of up to
        # `coalesce` batches (scheduler.EnginePool).
        self.grouped = arrangement == "grouped" and self.beam_size == 1 and self.fallback is None
        self.group, self.n_encoders, self.n_decoders = max(1, int(group)), max(1, int(n_encoders)), max(1, int(n_decoders))
        self.max_group_sequences = 128
        self.coalesce = max(1, int(coalesce))
        self.n_engines = n_engines
        # beam search / best-of-n with engines of their own: windows per pass; an engine then holds that many cached
        # windows and `rows` times as many sequences (None: the pool runs on `engine` and forks of its size)
        self.hyp_pass_windows = hyp_pass_windows
        self.pool = None                    # built at the first call: its engines are sized to what the call needs
        self._pool_key = None
        self._opts_for = None
        # measurement hook: decode rules used as they are instead of those derived from the language and the prompt
        # (bench.py: synthetic weights never emit end-of-text on cue, so its rules suppress it)
        self.fixed_options: Optional[DecodeOptions] = None

    def _pool_for(self, n_chunks: int, batch_size: int):
        """The engine pool for a job of `n_chunks` chunks in batches of `batch_size`; kept between calls and rebuilt
        only when a call needs more than the pool at hand (more chunks per batch, larger groups, more engines)."""
        if not self.grouped:
            if self.pool is None:
                rows = self.fallback.rows_per_window() if self.fallback is not None else self.beam_size
                fork_batch = self.hyp_pass_windows * rows if self.hyp_pass_windows else None
                self.pool = EnginePool(self.engine, self.n_engines, self.beam_size, self.patience, self.fallback, self.detokenize,
                                       fork_batch=fork_batch, fork_cache=self.hyp_pass_windows)
                self._pool_key = ("coalesced",)
            return self.pool
        n_batches = max(1, -(-n_chunks // batch_size))
        have = self._pool_key
        kept = self.pool is not None and have[0] == "grouped"
        # a pool whose encoder engines take larger batches serves smaller ones too
        batch = max(batch_size, have[1]) if kept else batch_size
        # a decode engine holds its group's cross-attention K/V (0.25 GB per sequence at large-v3) and, like every
        # engine, encoder-sized workspaces: groups are kept to `max_group_sequences` so that three of them fit beside
        # the encoders whatever batch_size the caller picks
        cap = max(1, self.max_group_sequences // batch)
        group = max(1, min(self.group, n_batches, cap))
        # Short jobs (a rank's share of a recording split over many GPUs): with two or three batches every batch is
        # decoded on an engine of its own as soon as it is encoded, so that decoding overlaps the encoder passes that
        # remain (large-v3, batches of 16, whole job: [1, 1] 456 ms against [2] 498; [1, 1, 1] 648 against [3] 748);
        # from four batches on the fewest, largest groups win ([4] = [2, 2] 834; [4, 4] 1404 against [3, 3, 2] 1438;
        # [7, 7] 2144 against [5, 5, 4] 2248).  One batch cannot be split to advantage: a greedy step over 16 sequences is
        # a chain of ~260 launches whose length does not shrink with fewer sequences ([8 + 7] in one group 328 ms, as two
        # groups 344, against 325 for the pass of 15).
        if n_batches <= min(3, self.n_decoders):
            group = 1
        n_groups = -(-n_batches // group)
        need = (batch, group, min(self.n_encoders, n_batches), min(self.n_decoders, n_groups))
        if kept and all(h >= n for h, n in zip(have[1:], need)):
            return self.pool
        if self.pool is not None:
            self._close_pool()
            if kept:        # keep what the old pool could do where the memory rule allows it (batch * group <= max_group_sequences)
                need = (batch, max(1, min(max(have[2], group), cap)), max(have[3], need[2]), max(have[4], need[3]))
        self.pool = self._make_grouped_pool(*need)
        self._pool_key = ("grouped",) + need
        self._opts_for = None
        return self.pool

    def _make_grouped_pool(self, batch, group, n_encoders, n_decoders):
        return GroupedPool(self.engine, batch, group, n_encoders, n_decoders)

    def _close_pool(self):
        """Close the pool's engines but keep the root engine (it owns the weights)."""
        pool, self.pool = self.pool, None
        if pool is None:
            return
        for ex in (getattr(pool, "_pool", None), getattr(pool, "_enc_pool", None), getattr(pool, "_dec_pool", None)):
            if ex is not None:
                ex.shutdown(wait=True)
        for e in reversed(pool.engines):
            if e is not self.engine:
                e.close()

    # -- decoding rules ---------------------------------------------------------------
    def decode_options(self, language: str) -> DecodeOptions:
        if self.fixed_options is not None:
            return self.fixed_options
        if self.synthetic and self.dims.vocab < 51865:
            return synthetic_options(self.dims, max_new=min(self.max_new, 64))
        prompt_ids = ()
        if self.initial_prompt and self.tokenizer is not None:
            prompt_ids = tuple(self.tokenizer.encode(" " + self.initial_prompt.strip(), add_special_tokens=False).ids)
        elif self.initial_prompt:
            print("⚠️ initial_prompt ignored: no tokenizer.json next to the weights")
        return whisper_options(self.dims, language, prompt_ids, max_new=self.max_new)

    def detokenize(self, ids) -> str:
        ids = [int(i) for i in ids]
        if self.tokenizer is not None:
            return self.tokenizer.decode(ids, skip_special_tokens=True)
        return "".join(f"<|{i}|>" for i in ids)

    # -- windows ------------------------------------------------------------------------
    def uses_vad(self) -> bool:
        return bool(self.vad_options) and self.vad_options.get("vad_method", "energy") != "fixed"

    def window_list(self, audio: np.ndarray):
        """Recording -> [{"start": s, "end": s}, ...]: speech windows of at most 30 s, or fixed 30 s chunks.
        Small enough to send to other ranks, which cut their own share with ``cut``."""
        if not self.uses_vad():
            n = int(np.asarray(audio).reshape(-1).shape[0])
            return [{"start": a / float(SAMPLE_RATE), "end": min(n, a + CHUNK_SAMPLES) / float(SAMPLE_RATE)}
                    for a in range(0, n, CHUNK_SAMPLES)]
        from . import vad
        return [{"start": w["start"], "end": w["end"]} for w in vad.speech_windows(self.engine, audio, self.vad_options)]

    @staticmethod
    def cut(audio: np.ndarray, windows):
        """Windows -> ([n, 480000] zero-padded chunks, [(start_s, end_s)])."""
        from . import vad
        return vad.cut_windows(audio, windows)

    def windows(self, audio: np.ndarray):
        """Recording -> ([n, 480000] chunks, [(start_s, end_s)]): speech windows, or fixed 30 s chunks."""
        return self.cut(audio, self.window_list(audio))

    # -- the hot path -------------------------------------------------------------------
    def transcribe_ids(self, audio: np.ndarray, batch_size: int = 16, language: Optional[str] = None):
        """-> (chunk spans [(start, end)], list of id arrays), no text handling."""
        chunks, spans = self.windows(audio)
        return spans, self.transcribe_ids_chunks(chunks, batch_size, language, durations=[b - a for a, b in spans])

    def transcribe_ids_chunks(self, chunks: np.ndarray, batch_size: int = 16, language: Optional[str] = None, durations=None):
        """Already-cut chunks [n, 480000] -> list of id arrays, in the order of ``chunks`` (also the multi-rank path).

        ``durations`` (seconds of audio per chunk): a pass lasts as long as its longest sequence, and a window's token
        count follows its duration, so passes are filled with windows of like duration, longest first (chunks are
        independent: the ids do not change)."""
        language = language or self.language
        n = len(chunks)
        if n == 0:
            return []
        if self.grouped:
            bs = max(1, min(int(batch_size), self.max_group_sequences))
        else:
            rows = self.fallback.rows_per_window() if self.fallback is not None else self.beam_size
            per_pass = self.hyp_pass_windows if self.hyp_pass_windows else self.engine.max_batch // rows
            bs = max(1, min(int(batch_size) * self.coalesce, per_pass))
        pool = self._pool_for(n, bs)
        if self._opts_for != language:
            pool.set_decode_options(self.decode_options(language))
            self._opts_for = language
        order = list(range(n))
        if durations is not None:
            order = sorted(order, key=lambda i: -durations[i])        # stable: equal durations keep their order
        if order != list(range(n)):
            chunks = chunks[order]
        got = []
        for ids, lens in pool.map([chunks[a:b] for a, b in batches_of(n, bs)]):
            got.extend(ids[i, : lens[i]].copy() for i in range(ids.shape[0]))
        out = [None] * n
        for rank, i in enumerate(order):
            out[i] = got[rank]
        return out

    def transcribe(self, audio: np.ndarray, batch_size: int = 16, language: Optional[str] = None, return_tokens: bool = False,
                   **_ignored) -> dict:
        """The reference's result dict (transcribe.py:162-176,221-223): segments of {"text", "start", "end"}.
        ``return_tokens=True`` adds each segment's ids under "tokens" (no counterpart in the reference)."""
        language = language or self.language
        spans, ids = self.transcribe_ids(audio, batch_size, language)
        segments = [{"text": self.detokenize(t), "start": round(a, 3), "end": round(b, 3)} for (a, b), t in zip(spans, ids)]
        if return_tokens:
            for seg, t in zip(segments, ids):
                seg["tokens"] = [int(i) for i in t]
        return {"segments": segments, "language": language}

    def close(self):
        self._close_pool()
        self.engine.close()


def load_model(whisper_arch: str, device: str = "cuda", device_index: int = 0, compute_type: str = "float16",
               asr_options: Optional[dict] = None, language: Optional[str] = None, vad_options: Optional[dict] = None,
               task: str = "transcribe", max_batch: Optional[int] = None, n_engines: Optional[int] = None, coalesce: int = 4,
               arrangement: str = "grouped", group: int = 7, n_encoders: int = 2, n_decoders: int = 3,
               **_ignored) -> Model:
    """Build an engine on GPU ``device_index``.

    whisper_arch: a local directory in Hugging Face layout, a model size whose directory is
    ``$MW_MODEL_DIR/<size>``, or ``synthetic:<size>``.
    The engine's own knobs have defaults a caller of the reference's signature never needs to touch.
    ``arrangement="grouped"`` (greedy decoding, the default): the encoder half of every batch of
    ``transcribe(batch_size=...)`` chunks runs on one of ``n_encoders`` engines, and ``n_decoders`` engines each run
    the greedy loop over up to ``group`` such batches at once, so the decoder weights stream once per token per
    group; the engines are created at the first call, sized to the recording (a decode engine takes about 0.32 GB
    per chunk at large-v3).  ``arrangement="coalesced"`` (always used for beam search): ``n_engines`` engines in
    flight (default 2; 3 for beam search) that each run whole passes of ``coalesce`` batches, ``max_batch`` chunks at
    most (default 16 * coalesce).
    """
    coalesce = max(1, int(coalesce))
    beam = int((asr_options or {}).get("beam_size", 1))
    if n_engines is None:       # measured (tools/beam_probe.py): beam 5 on passes of 25 recordings, 2 engines 1577 audio-s/s, 3 engines 1752
        n_engines = 3 if beam > 1 else 2
    from .fallback import FallbackOptions
    fallback = FallbackOptions.from_asr_options(asr_options or {})
    hyp_pass_windows = None
    if max_batch is None:
        rows = fallback.rows_per_window() if fallback.enabled else beam
        if rows > 1:
            # several hypotheses per window: the pool's engines hold up to 48 windows per pass (their K/V cached once
            # each) and `rows` times as many sequences; this engine only lends its weights.  Measured, beam 5, three
            # engines: passes of 25 windows 1752 audio-s/s, of 50 windows 1889.
            hyp_pass_windows = max(1, min(48, 512 // rows, 16 * coalesce))
            max_batch = max(16, rows)
        else:
            max_batch = 16 if arrangement == "grouped" else min(128, 16 * coalesce)
    if device not in _GPU_NAMES:
        raise EngineError(f"device={device!r}: this engine runs on an MI355X GPU only (use 'cuda'); there is no CPU path")
    if compute_type not in _FP16_NAMES + _FP8_NAMES:
        raise EngineError(f"compute_type={compute_type!r} is not available: 'float16' (fp16 weights, fp32 accumulation) or "
                          "'float8' (e4m3 Linear weights with a scale per output row, quantised at load) or 'float8_kv8' "
                          "(the same plus an e4m3 cross-attention K/V cache)")
    if task != "transcribe":
        raise EngineError("only task='transcribe' is implemented")
    asr_options = dict(asr_options or {})
    synthetic = whisper_arch.startswith("synthetic:")
    tokenizer = None
    if synthetic:
        from . import synth
        dims = get_dims(whisper_arch.split(":", 1)[1])
        try:
            import torch
            weights = synth.make_weights_torch(dims, 1234, torch.device("cuda", device_index))
        except Exception:
            weights = synth.make_weights_np(dims, 1234)
    else:
        path = whisper_arch
        if not os.path.isdir(path):
            root = os.environ.get("MW_MODEL_DIR")
            cand = os.path.join(root, whisper_arch) if root else None
            if cand and os.path.isdir(cand):
                path = cand
            else:
                raise FileNotFoundError(
                    f"model {whisper_arch!r}: not a directory, and no $MW_MODEL_DIR/{whisper_arch}. Nothing is downloaded: "
                    "point to a local Hugging Face-layout checkpoint (known sizes: " + ", ".join(sorted(MODELS)) + ")")
        dims = _dims_from_config(path)
        weights = _load_state_dict(path)
        tok = os.path.join(path, "tokenizer.json")
        if os.path.exists(tok):
            from tokenizers import Tokenizer
            tokenizer = Tokenizer.from_file(tok)
    engine = Engine(dims, device=device_index, max_batch=max_batch, weight_format="fp8" if compute_type in _FP8_NAMES else "f16",
                    kv_format="fp8" if compute_type in _KV8_NAMES else "f16")
    engine.load_weights(weights)
    del weights
    return Model(engine, dims, language or "zh", tokenizer, asr_options.get("initial_prompt"), vad_options,
                 synthetic=synthetic, max_new=int(asr_options.get("max_new_tokens", 224)), n_engines=n_engines,
                 beam_size=int(asr_options.get("beam_size", 1)), patience=float(asr_options.get("patience", 1.0)), coalesce=coalesce,
                 arrangement=arrangement, group=group, n_encoders=n_encoders, n_decoders=n_decoders, fallback=fallback,
                 hyp_pass_windows=hyp_pass_windows)
