"""Python face of the HIP engine: thin, typed wrappers over the C ABI.

PyTorch appears only as a container for weights (``tensor.data_ptr()``); all
compute happens inside libmwhip.so.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Mapping, Optional, Sequence

import numpy as np

from . import _lib
from .decode_options import DecodeOptions
from .dims import CHUNK_SAMPLES, N_FRAMES, WhisperDims

_DTYPES = {"float32": 0, "float16": 1, "bfloat16": 2}


class EngineError(RuntimeError):
    pass


def _ptr(a) -> C.c_void_p:
    """Address of a numpy array or a torch tensor (host or device).

    The engine reads device buffers on its own stream, which is not ordered against the stream that filled
    them: a tensor on a GPU is handed over only after the work queued on torch's current stream for that
    device has finished."""
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)
    if getattr(a, "is_cuda", False):
        import torch
        torch.cuda.current_stream(a.device).synchronize()
    return C.c_void_p(a.data_ptr())


class Engine:
    """One GPU's engine.  ``max_batch`` chunks of 30 s are processed per call.

    ``weight_format``: ``"f16"``, or ``"fp8"`` (Linear weights of the layers kept as OCP e4m3 codes with a
    scale per output row, quantised as they are loaded; see ``mw_weight_format`` in include/mwhip.h).
    ``kv_format``: ``"f16"``, or ``"fp8"`` (the cached cross-attention K/V as e4m3 codes with a scale per row
    of 64; fp8 weights only; ``mw_kv_format``).
    """

    def __init__(self, dims: WhisperDims, device: int = 0, max_batch: int = 16, weight_format: str = "f16",
                 kv_format: str = "f16"):
        self._lib = _lib.load()
        self.dims = dims
        self.device = device
        self.max_batch = max_batch
        self.cache_sequences = max_batch
        self.opts: Optional[DecodeOptions] = None
        if weight_format not in _lib.WEIGHT_FORMATS:
            raise EngineError(f"weight_format={weight_format!r}: expected one of {sorted(_lib.WEIGHT_FORMATS)}")
        if kv_format not in _lib.KV_FORMATS:
            raise EngineError(f"kv_format={kv_format!r}: expected one of {sorted(_lib.KV_FORMATS)}")
        self.weight_format, self.kv_format = weight_format, kv_format
        desc = _lib.ModelDesc(dims.n_mels, dims.d_model, dims.n_heads, dims.enc_layers, dims.dec_layers,
                              dims.ffn_dim, dims.vocab, dims.n_audio_ctx, dims.n_text_ctx, _lib.WEIGHT_FORMATS[weight_format],
                              _lib.KV_FORMATS[kv_format])
        h = C.c_void_p()
        rc = self._lib.mw_create(C.byref(desc), device, max_batch, C.byref(h))
        if rc != 0:
            raise EngineError(f"mw_create failed ({rc}): {self._lib.mw_last_error(None).decode()}")
        self._h = h
        self._forks = []

    def fork(self, max_batch: Optional[int] = None, cache_sequences: Optional[int] = None) -> "Engine":
        """A second engine on the same GPU sharing these weights (own stream and caches, sized for
        ``max_batch`` chunks: this engine's by default); safe to drive from another thread while this one is busy.
        ``cache_sequences``: chunks its cross-attention K/V cache holds (default ``max_batch``); an engine that decodes
        several hypotheses per chunk (beam search, best-of-n) needs ``max_batch`` = chunks x hypotheses sequences but
        a cache for the chunks only (``mw_fork_cached``)."""
        h = C.c_void_p()
        max_batch = self.max_batch if max_batch is None else int(max_batch)
        cache = max_batch if cache_sequences is None else int(cache_sequences)
        self._check(self._lib.mw_fork_cached(self._h, max_batch, cache, C.byref(h)), "mw_fork_cached")
        child = object.__new__(Engine)
        child._lib, child.dims, child.device, child.max_batch = self._lib, self.dims, self.device, max_batch
        child.cache_sequences = cache
        child.weight_format, child.kv_format = self.weight_format, self.kv_format
        child.opts = self.opts
        child._h = h
        child._parent = self          # keeps the weights alive
        child._forks = []
        self._forks.append(weakref.ref(child))
        return child

    # -- plumbing -------------------------------------------------------------
    def _check(self, rc: int, what: str):
        if rc != 0:
            raise EngineError(f"{what} failed ({rc}): {self._lib.mw_last_error(self._h).decode()}")

    def close(self):
        for ref in getattr(self, "_forks", []):      # forks borrow this engine's weights: they go first
            child = ref()
            if child is not None:
                child.close()
        if getattr(self, "_h", None):
            self._lib.mw_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key: str, value: int):
        self._check(self._lib.mw_set_option(self._h, key.encode(), int(value)), "mw_set_option")

    # -- weights ----------------------------------------------------------------
    def load_weights(self, weights: Mapping[str, object]):
        """``weights``: HF-layout state dict of numpy arrays or torch tensors (CPU or this GPU)."""
        for name, w in weights.items():
            if isinstance(w, np.ndarray):
                w = np.ascontiguousarray(w)
                dt = _DTYPES.get(str(w.dtype))
                shape = w.shape
            else:
                w = w.contiguous()
                dt = _DTYPES.get(str(w.dtype).replace("torch.", ""))
                shape = tuple(w.shape)
            if dt is None:
                raise EngineError(f"weight {name}: unsupported dtype {w.dtype}")
            shp = (C.c_int64 * len(shape))(*shape)
            self._check(self._lib.mw_set_weight(self._h, name.encode(), _ptr(w), shp, len(shape), dt),
                        f"mw_set_weight({name})")
        self._check(self._lib.mw_finalize(self._h), "mw_finalize")

    def set_decode_options(self, o: DecodeOptions):
        prompt = (C.c_int32 * len(o.prompt))(*o.prompt)
        sup_l = o.effective_suppress()
        sup = (C.c_int32 * max(1, len(sup_l)))(*sup_l)
        # a suppressed EOT must not be re-enabled by the begin list; it is a union anyway
        bsup = (C.c_int32 * max(1, len(o.begin_suppress)))(*o.begin_suppress)
        s = _lib.DecodeOpts(prompt, len(o.prompt), sup, len(sup_l), bsup, len(o.begin_suppress), o.eot, o.max_new)
        self._check(self._lib.mw_set_decode_opts(self._h, C.byref(s)), "mw_set_decode_opts")
        self.opts = o

    # -- stages -------------------------------------------------------------------
    @staticmethod
    def _pcm_batch(pcm) -> int:
        shape = tuple(pcm.shape)
        if len(shape) != 2 or shape[1] != CHUNK_SAMPLES:
            raise EngineError(f"pcm must be [batch, {CHUNK_SAMPLES}] float32, got {shape}")
        if "float32" not in str(pcm.dtype):
            raise EngineError(f"pcm must be float32, got {pcm.dtype}")
        return shape[0]

    def logmel(self, pcm, want_output: bool = True) -> Optional[np.ndarray]:
        B = self._pcm_batch(pcm)
        out = np.empty((B, self.dims.n_mels, N_FRAMES), np.float32) if want_output else None
        self._check(self._lib.mw_logmel(self._h, _ptr(pcm), B, _ptr(out) if want_output else None), "mw_logmel")
        return out

    def logmel_noalloc(self, pcm, mel_out) -> None:
        """``logmel`` through the entry point that allocates nothing: ``pcm`` [B, 480000] and ``mel_out``
        [B, n_mels, 3000] are float32 tensors on this GPU (contiguous); at most 128 chunks."""
        B = self._pcm_batch(pcm)
        if tuple(mel_out.shape) != (B, self.dims.n_mels, N_FRAMES) or "float32" not in str(mel_out.dtype):
            raise EngineError(f"mel_out must be float32 {(B, self.dims.n_mels, N_FRAMES)}")
        self._check(self._lib.mw_logmel_noalloc(self._h, _ptr(pcm), B, _ptr(mel_out)), "mw_logmel_noalloc")

    def frame_energy_db(self, audio, frame: int = 400, hop: int = 160) -> np.ndarray:
        """Frame log-energies (dB) of a whole recording: float32 [n] (numpy, or a tensor on this GPU)
        -> float32 [ceil(n / hop)].  Input of the speech-activity window search (vad.py)."""
        if isinstance(audio, np.ndarray):
            audio = np.ascontiguousarray(audio, dtype=np.float32).reshape(-1)
            n = int(audio.shape[0])
        else:
            if str(audio.dtype) != "torch.float32" or audio.dim() != 1 or not audio.is_contiguous():
                raise EngineError("audio tensor must be contiguous float32 [n]")
            n = int(audio.shape[0])
        if n == 0:
            return np.zeros(0, np.float32)
        n_frames = (n + hop - 1) // hop
        out = np.empty(n_frames, np.float32)
        self._check(self._lib.mw_frame_energy_db(self._h, _ptr(audio), n, frame, hop, _ptr(out), n_frames), "mw_frame_energy_db")
        return out

    def encode(self, batch: int, mel: Optional[np.ndarray] = None, want_output: bool = True) -> Optional[np.ndarray]:
        if mel is not None:
            mel = np.ascontiguousarray(mel, dtype=np.float32)
            if mel.shape != (batch, self.dims.n_mels, N_FRAMES):
                raise EngineError(f"mel must be {(batch, self.dims.n_mels, N_FRAMES)}, got {mel.shape}")
        out = np.empty((batch, self.dims.n_audio_ctx, self.dims.d_model), np.float32) if want_output else None
        self._check(self._lib.mw_encode(self._h, _ptr(mel) if mel is not None else None, batch,
                                        _ptr(out) if want_output else None), "mw_encode")
        return out

    def encode_into(self, pcm, dst: "Engine", slot0: int) -> None:
        """log-mel + encoder + cross K/V of ``pcm`` [B, 480000] on this engine, the K/V into sequences
        ``slot0 .. slot0 + B - 1`` of ``dst``'s cache (this engine, its parent or another fork of it)."""
        B = self._pcm_batch(pcm)
        self._check(self._lib.mw_encode_into(self._h, _ptr(pcm), B, dst._h, int(slot0)), "mw_encode_into")

    def set_encoded(self, batch: int) -> None:
        """Sequences 0 .. batch - 1 of this engine's cache have been filled by ``encode_into`` calls."""
        self._check(self._lib.mw_set_encoded(self._h, int(batch)), "mw_set_encoded")

    def read_cross_kv(self, layer: int, sequence: int):
        """(K, V) float32 [n_heads, n_audio_ctx, 64] that decoder layer ``layer`` holds for ``sequence`` (test hook)."""
        shape = (self.dims.n_heads, self.dims.n_audio_ctx, 64)
        k, v = np.empty(shape, np.float32), np.empty(shape, np.float32)
        self._check(self._lib.mw_read_cross_kv(self._h, int(layer), int(sequence), _ptr(k), _ptr(v)), "mw_read_cross_kv")
        return k, v

    def decode_greedy(self, batch: int, want_logits: bool = False):
        if self.opts is None:
            raise EngineError("set_decode_options first")
        ids = np.empty((batch, self.opts.max_new), np.int32)
        lens = np.empty(batch, np.int32)
        logits = np.zeros((self.opts.max_new, batch, self.dims.vocab), np.float32) if want_logits else None
        self._check(self._lib.mw_decode_greedy(self._h, batch, _ptr(ids), _ptr(lens),
                                               _ptr(logits) if want_logits else None), "mw_decode_greedy")
        return ids, lens, logits

    def decode_beam(self, n_audio: int, beam: int = 5, patience: float = 1.0, want_hypotheses: bool = False):
        """Beam search over the last ``encode(n_audio)``: -> (ids [n_audio, max_new], lens, avg_logprob)
        and, with ``want_hypotheses``, (hyp_ids [n_audio, beam, max_new], hyp_len, hyp_sum_logprob)."""
        if self.opts is None:
            raise EngineError("set_decode_options first")
        ids = np.empty((n_audio, self.opts.max_new), np.int32)
        lens = np.empty(n_audio, np.int32)
        avg = np.empty(n_audio, np.float32)
        hi = np.empty((n_audio, beam, self.opts.max_new), np.int32) if want_hypotheses else None
        hl = np.empty((n_audio, beam), np.int32) if want_hypotheses else None
        hs = np.empty((n_audio, beam), np.float32) if want_hypotheses else None
        self._check(self._lib.mw_decode_beam(self._h, n_audio, beam, C.c_float(patience), _ptr(ids), _ptr(lens), _ptr(avg),
                                             _ptr(hi) if want_hypotheses else None, _ptr(hl) if want_hypotheses else None,
                                             _ptr(hs) if want_hypotheses else None), "mw_decode_beam")
        return (ids, lens, avg, hi, hl, hs) if want_hypotheses else (ids, lens, avg)

    def decode_sample(self, n_audio: int, temperature: float, best_of: int = 1, seed: int = 0, active=None,
                      want_samples: bool = False, want_logits: bool = False):
        """``best_of`` samples per recording at ``temperature`` over the last ``encode(n_audio)`` (0 = the arg-max with its
        log-probabilities): -> (ids [n_audio, max_new], lens, avg_logprob) of each recording's best sample; with
        ``want_samples`` also (smp_ids [n_audio, best_of, max_new], smp_len, smp_sum_logprob); with ``want_logits`` also
        the logits [max_new, n_audio * best_of, vocab].  ``active`` (bool [n_audio]): recordings to decode, the others
        come back empty."""
        if self.opts is None:
            raise EngineError("set_decode_options first")
        ids = np.empty((n_audio, self.opts.max_new), np.int32)
        lens = np.empty(n_audio, np.int32)
        avg = np.empty(n_audio, np.float32)
        act = None if active is None else np.ascontiguousarray(np.asarray(active).astype(bool), dtype=np.uint8)
        if act is not None and act.shape != (n_audio,):
            raise EngineError(f"active must have shape ({n_audio},)")
        si = np.empty((n_audio, best_of, self.opts.max_new), np.int32) if want_samples else None
        sl = np.empty((n_audio, best_of), np.int32) if want_samples else None
        ss = np.empty((n_audio, best_of), np.float32) if want_samples else None
        lg = np.zeros((self.opts.max_new, n_audio * best_of, self.dims.vocab), np.float32) if want_logits else None
        opt = lambda a: _ptr(a) if a is not None else None     # noqa: E731
        self._check(self._lib.mw_decode_sample(self._h, n_audio, best_of, C.c_float(temperature), C.c_uint32(seed & 0xFFFFFFFF), opt(act),
                                               _ptr(ids), _ptr(lens), _ptr(avg), opt(si), opt(sl), opt(ss), opt(lg)), "mw_decode_sample")
        out = (ids, lens, avg)
        if want_samples:
            out += (si, sl, ss)
        if want_logits:
            out += (lg,)
        return out

    def transcribe_batch_beam(self, pcm, beam: int, patience: float = 1.0):
        """log-mel -> encoder -> beam search for up to max_batch // beam chunks: -> (ids, lens), as transcribe_batch."""
        n = self._pcm_batch(pcm)
        self.logmel(pcm, want_output=False)
        self.encode(n, want_output=False)
        ids, lens, _ = self.decode_beam(n, beam, patience)
        return ids, lens

    def decode_forced(self, tokens: np.ndarray) -> np.ndarray:
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        B, n = tokens.shape
        out = np.empty((B, n, self.dims.vocab), np.float32)
        self._check(self._lib.mw_decode_forced(self._h, B, _ptr(tokens), n, _ptr(out)), "mw_decode_forced")
        return out

    def transcribe_batch(self, pcm):
        """pcm: numpy [B, 480000] float32 on the host, or a torch tensor already on this GPU."""
        if self.opts is None:
            raise EngineError("set_decode_options first")
        B = self._pcm_batch(pcm)
        ids = np.empty((B, self.opts.max_new), np.int32)
        lens = np.empty(B, np.int32)
        self._check(self._lib.mw_transcribe_batch(self._h, _ptr(pcm), B, _ptr(ids), _ptr(lens)), "mw_transcribe_batch")
        return ids, lens

    # -- measurement --------------------------------------------------------------------
    def stage_ms(self) -> dict:
        a = (C.c_float * 5)()
        self._check(self._lib.mw_stage_ms(self._h, a), "mw_stage_ms")
        return {"logmel": a[0], "encoder": a[1], "cross_kv": a[2], "decode": a[3], "decode_steps": int(a[4])}

    def profile_cross_attention(self, batch: int, iters: int = 64):
        ms, nbytes = C.c_float(), C.c_double()
        self._check(self._lib.mw_profile_cross_attention(self._h, batch, iters, C.byref(ms), C.byref(nbytes)),
                    "mw_profile_cross_attention")
        return ms.value, nbytes.value

    def profile_decode_gemvs(self, batch: int, iters: int = 256):
        """(mean ms per decoder layer of its six projection launches, weight bytes per layer)."""
        ms, nbytes = C.c_float(), C.c_double()
        self._check(self._lib.mw_profile_decode_gemvs(self._h, batch, iters, C.byref(ms), C.byref(nbytes)),
                    "mw_profile_decode_gemvs")
        return ms.value, nbytes.value

    def synchronize(self):
        self._check(self._lib.mw_synchronize(self._h), "mw_synchronize")
