"""Forced alignment: ``load_align_model`` / ``align``, the two whisperx calls of reference transcribe.py:127-135.

    model_a, metadata = whisperx.load_align_model(language_code="zh", device=device, model_name=None)
    result_aligned = whisperx.align(result["segments"], model_a, metadata, audio, device)

For "zh" whisperx resolves ``model_name=None`` to a Hugging Face ``Wav2Vec2ForCTC`` checkpoint of the
wav2vec2-large-xlsr-53 family.  Here the acoustic model runs in libmwhip.so (``mw_align_emissions``: raw
waveform -> frame-wise log-probabilities over the character vocabulary); this module is the host logic around it,
restated from whisperx's alignment.py as published (3.x: characters outside the vocabulary align as a wildcard,
trellis over (frames, tokens), backtracking, repeats merged into character segments, characters gathered into
words and sentences, missing times filled from the nearest neighbour).  Recent whisperx releases walk the trellis
back with a beam of width 2 instead of the plain walk; how that beam ranks its partial paths could not be checked
offline, so the plain walk, which returns the trellis optimum, is what runs here.

whisperx is not in the reference tree and cannot be installed here, the reference holds no alignment vectors, and
nltk's Punkt sentence tokenizer (which whisperx uses to cut a segment into sentences) is absent too:
PARITY WITH WHISPERX IS UNPINNED.  Sentences are cut by the plain rule in ``sentence_spans``.

There is no CPU fallback: without libmwhip.so and a GPU, ``load_align_model`` raises.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
import queue
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .dims import SAMPLE_RATE
from .engine import EngineError, _DTYPES, _ptr

LANGUAGES_WITHOUT_SPACES = ("ja", "zh")
MAX_ALIGN_SECONDS = 60                    # longest segment one call aligns (mw_align_desc.max_samples)
MIN_SAMPLES = 400                         # shorter segments are zero-padded to this (one receptive field)


# ------------------------------------------------------------------------------------------------
# the acoustic model
# ------------------------------------------------------------------------------------------------
class Aligner:
    """A wav2vec2 CTC model on one GPU: ``emissions(wave) -> float32 [frames][vocab]`` log-probabilities.

    ``emissions`` may be called from several threads at once: up to ``n_streams`` calls run concurrently, each on a
    copy of the workspaces with its own stream (``mw_align_fork``; the weights are shared).  One segment is about
    1500 frames, a fraction of what fills the GPU."""

    def __init__(self, cfg: dict, device: int = 0, max_seconds: int = MAX_ALIGN_SECONDS, n_streams: int = 4, pack_seconds: int = 240):
        if cfg.get("feat_extract_norm", "group") != "layer" or not cfg.get("do_stable_layer_norm", False) or not cfg.get("conv_bias", False):
            raise EngineError("only the wav2vec2 variant with a layer-norm feature extractor, convolution biases and stable "
                              "layer norm (wav2vec2-large-xlsr-53 and its fine-tunes) is built")
        if len(set(cfg["conv_dim"])) != 1:
            raise EngineError("the feature extractor's layers must share one width")
        self._lib = _lib.load()
        self.cfg = cfg
        self.device = device
        d = _lib.AlignDesc()
        d.n_conv, d.conv_dim = len(cfg["conv_dim"]), cfg["conv_dim"][0]
        for i, (k, s) in enumerate(zip(cfg["conv_kernel"], cfg["conv_stride"])):
            d.conv_kernel[i], d.conv_stride[i] = k, s
        d.hidden, d.n_heads, d.n_layers = cfg["hidden_size"], cfg["num_attention_heads"], cfg["num_hidden_layers"]
        d.ffn_dim = cfg["intermediate_size"]
        d.pos_kernel, d.pos_groups = cfg["num_conv_pos_embeddings"], cfg["num_conv_pos_embedding_groups"]
        d.vocab = cfg["vocab_size"]
        # One pass of the model takes a PACK of segments (mw_align_segments): `pack_seconds` of audio as slices of the
        # pack's longest segment; a single segment may be up to `max_seconds` long.
        self.max_segment_samples = int(max_seconds) * SAMPLE_RATE
        d.max_samples = max(int(max_seconds), int(pack_seconds)) * SAMPLE_RATE
        self.desc = d
        h = C.c_void_p()
        rc = self._lib.mw_align_create(C.byref(d), device, C.byref(h))
        if rc != 0:
            raise EngineError(f"mw_align_create failed ({rc}): {self._lib.mw_align_last_error(None).decode()}")
        self._h = h
        self.n_streams = max(1, int(n_streams))
        self._forks: List[C.c_void_p] = []
        self._free: "queue.Queue" = queue.Queue()
        self._ready = False

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise EngineError(f"{what} failed ({rc}): {self._lib.mw_align_last_error(self._h).decode()}")

    def load_weights(self, weights) -> None:
        """``weights``: the checkpoint's state dict (Hugging Face names), numpy arrays or torch tensors."""
        for name, w in weights.items():
            if "masked_spec_embed" in name or name.startswith(("quantizer.", "project_q.", "project_hid.")):
                continue                                           # pre-training heads: not part of the CTC forward pass
            if isinstance(w, np.ndarray):
                w = np.ascontiguousarray(w)
                dt, shape = _DTYPES.get(str(w.dtype)), w.shape
            else:
                w = w.contiguous()
                dt, shape = _DTYPES.get(str(w.dtype).replace("torch.", "")), tuple(w.shape)
            if dt is None:
                raise EngineError(f"weight {name}: unsupported dtype {w.dtype}")
            shp = (C.c_int64 * len(shape))(*shape)
            self._check(self._lib.mw_align_set_weight(self._h, name.encode(), _ptr(w), shp, len(shape), dt), f"mw_align_set_weight({name})")
        self._check(self._lib.mw_align_finalize(self._h), "mw_align_finalize")
        self._free.put(self._h)
        for _ in range(self.n_streams - 1):
            h = C.c_void_p()
            self._check(self._lib.mw_align_fork(self._h, C.byref(h)), "mw_align_fork")
            self._forks.append(h)
            self._free.put(h)
        self._ready = True

    def n_frames(self, n_samples: int) -> int:
        return int(self._lib.mw_align_frames(C.byref(self.desc), int(n_samples)))

    def _check_length(self, n: int, what: str):
        if n > self.max_segment_samples:
            raise EngineError(f"{what}: more than max_samples samples ({self.max_segment_samples}: max_seconds of this aligner)")

    def emissions(self, wave) -> np.ndarray:
        wave = np.ascontiguousarray(wave, dtype=np.float32).reshape(-1)
        self._check_length(len(wave), "mw_align_emissions")
        out = np.empty((max(self.n_frames(len(wave)), 1), self.cfg["vocab_size"]), np.float32)
        n = C.c_int32(0)
        h = self._free.get() if self._ready else self._h        # before load_weights the call itself reports the state
        try:
            rc = self._lib.mw_align_emissions(h, _ptr(wave), len(wave), _ptr(out), C.byref(n))
            if rc != 0:
                raise EngineError(f"mw_align_emissions failed ({rc}): {self._lib.mw_align_last_error(h).decode()}")
        finally:
            if self._ready:
                self._free.put(h)
        return out[: n.value]

    MAX_SEGMENT_TOKENS = 1024

    def segment_path(self, wave, tokens: Sequence[int], blank_id: int = 0):
        """``ctc_path(self.emissions(wave), tokens, blank_id)`` in one call (``mw_align_segment``): the emissions stay on
        the GPU, only the columns of the blank and of the segment's tokens come to the host.
        -> (path or None, number of frames)."""
        if len(tokens) > self.MAX_SEGMENT_TOKENS:
            em = self.emissions(wave)
            return ctc_path(em, tokens, blank_id), em.shape[0]
        wave = np.ascontiguousarray(wave, dtype=np.float32).reshape(-1)
        self._check_length(len(wave), "mw_align_segment")
        tok = np.ascontiguousarray(tokens, dtype=np.int32)
        T = max(self.n_frames(len(wave)), 1)
        idx, sc = np.empty(T, np.int32), np.empty(T, np.float32)
        n = C.c_int32(0)
        h = self._free.get() if self._ready else self._h
        try:
            rc = self._lib.mw_align_segment(h, _ptr(wave), len(wave), tok.ctypes.data, len(tok), int(blank_id), idx.ctypes.data,
                                            sc.ctypes.data, C.byref(n))
            if rc not in (0, -2):
                raise EngineError(f"mw_align_segment failed ({rc}): {self._lib.mw_align_last_error(h).decode()}")
        finally:
            if self._ready:
                self._free.put(h)
        if rc == -2 and n.value == 0:                           # MW_ERR_STATE before the model ran: not finalized
            raise EngineError(f"mw_align_segment failed ({rc}): {self._lib.mw_align_last_error(h).decode()}")
        if rc == -2:
            return None, n.value
        return PathArrays(idx[: n.value].copy(), sc[: n.value].copy()), n.value

    MAX_PACK_SEGMENTS = 32

    def pack_fits(self, sample_counts: Sequence[int], token_counts: Sequence[int]) -> bool:
        """Whether segments of these lengths (samples) and token counts go through ``segments_path`` as one pack."""
        n = len(sample_counts)
        if n < 1 or n > self.MAX_PACK_SEGMENTS or max(token_counts) > self.MAX_SEGMENT_TOKENS:
            return False
        if n * max(sample_counts) > self.desc.max_samples or (n == 1 and sample_counts[0] > self.max_segment_samples):
            return False
        gather = sum(max(self.n_frames(c), 1) * (t + 1) for c, t in zip(sample_counts, token_counts))
        return gather <= (self.n_frames(self.desc.max_samples) + 64) * (self.MAX_SEGMENT_TOKENS + 1)

    def segments_path(self, waves: Sequence[np.ndarray], tokens: Sequence[Sequence[int]], blank_id: int = 0):
        """``segment_path`` for a pack of segments in one pass of the model (``mw_align_segments``): -> [(path or None,
        frames)] in the order given.  The caller checks ``pack_fits`` first."""
        n = len(waves)
        waves = [np.ascontiguousarray(w, dtype=np.float32).reshape(-1) for w in waves]
        for w in waves:
            self._check_length(len(w), "mw_align_segments")
        toks = [np.ascontiguousarray(t, dtype=np.int32) for t in tokens]
        frames = [max(self.n_frames(len(w)), 1) for w in waves]
        ptrs = (C.c_void_p * n)(*[w.ctypes.data for w in waves])
        counts = (C.c_int64 * n)(*[len(w) for w in waves])
        tok_all = np.ascontiguousarray(np.concatenate(toks), dtype=np.int32)
        n_tok = (C.c_int32 * n)(*[len(t) for t in toks])
        idx, sc = np.empty(sum(frames), np.int32), np.empty(sum(frames), np.float32)
        got, status = (C.c_int32 * n)(), (C.c_int32 * n)()
        h = self._free.get() if self._ready else self._h
        try:
            rc = self._lib.mw_align_segments(h, n, ptrs, counts, tok_all.ctypes.data, n_tok, int(blank_id), idx.ctypes.data, sc.ctypes.data,
                                             got, status)
            if rc != 0:
                raise EngineError(f"mw_align_segments failed ({rc}): {self._lib.mw_align_last_error(h).decode()}")
        finally:
            if self._ready:
                self._free.put(h)
        out, at = [], 0
        for b in range(n):
            T = got[b]
            if status[b] != 0:
                out.append((None, T))
            else:
                out.append((PathArrays(idx[at: at + T].copy(), sc[at: at + T].copy()), T))
            at += T
        return out

    def close(self):
        for h in getattr(self, "_forks", []):                  # forks borrow the weights: they go first
            self._lib.mw_align_destroy(h)
        self._forks = []
        if getattr(self, "_h", None):
            self._lib.mw_align_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _load_state_dict(path: str) -> dict:
    st = os.path.join(path, "model.safetensors")
    if os.path.exists(st):
        from safetensors.numpy import load_file
        return load_file(st)
    pt = os.path.join(path, "pytorch_model.bin")
    if os.path.exists(pt):
        import torch
        return torch.load(pt, map_location="cpu", weights_only=True)
    raise FileNotFoundError(f"{path}: no model.safetensors or pytorch_model.bin")


def load_align_model(language_code: str, device="cuda", model_name: Optional[str] = None, model_dir: Optional[str] = None):
    """-> (model, metadata), like whisperx.load_align_model (reference transcribe.py:130).

    There is no network: ``model_name`` (or, when it is None, the directory ``MW_ALIGN_MODEL`` names) must be a LOCAL
    directory in Hugging Face layout (config.json, model.safetensors or pytorch_model.bin, vocab.json); ``model_dir``
    is joined in front of a relative ``model_name``.  metadata = {"language", "dictionary", "type"} as whisperx returns
    it, the dictionary being the tokenizer's vocabulary with lower-cased keys."""
    name = model_name or os.environ.get("MW_ALIGN_MODEL")
    if not name:
        raise FileNotFoundError(f"no alignment model for language {language_code!r}: pass model_name=<local directory> or set MW_ALIGN_MODEL "
                                "(whisperx would download a wav2vec2 checkpoint here; this build has no network)")
    path = name if os.path.isabs(name) or not model_dir else os.path.join(model_dir, name)
    with open(os.path.join(path, "config.json"), encoding="utf-8") as f:
        cfg = json.load(f)
    with open(os.path.join(path, "vocab.json"), encoding="utf-8") as f:
        vocab = json.load(f)
    index = 0
    if isinstance(device, str) and ":" in device:
        index = int(device.split(":")[1])
    elif isinstance(device, int):
        index = device
    model = Aligner(cfg, device=index)
    model.load_weights(_load_state_dict(path))
    metadata = {"language": language_code, "dictionary": {ch.lower(): int(code) for ch, code in vocab.items()}, "type": "huggingface"}
    return model, metadata


# ------------------------------------------------------------------------------------------------
# alignment of a token sequence to the emissions
# ------------------------------------------------------------------------------------------------
@dataclass
class Point:
    token_index: int
    time_index: int
    score: float


class PathArrays:
    """A path as the library returns it: token index and score per frame, as arrays.  Reads like the list of
    ``Point`` it stands for (iteration, indexing, length), and ``merge_repeats`` takes it without building one object
    per frame (a 30 s segment has 1499)."""
    __slots__ = ("token_index", "score")

    def __init__(self, token_index: np.ndarray, score: np.ndarray):
        self.token_index, self.score = token_index, score

    def __len__(self) -> int:
        return int(self.token_index.shape[0])

    def __getitem__(self, t):
        if isinstance(t, slice):
            return [self[i] for i in range(*t.indices(len(self)))]
        if t < 0:
            t += len(self)
        return Point(int(self.token_index[t]), int(t), float(self.score[t]))

    def __iter__(self):
        return (Point(int(j), t, float(p)) for t, (j, p) in enumerate(zip(self.token_index, self.score)))


@dataclass
class Segment:
    label: str
    start: int
    end: int
    score: float

    @property
    def length(self) -> int:
        return self.end - self.start


def ctc_path(emission: np.ndarray, tokens: Sequence[int], blank_id: int = 0) -> Optional["PathArrays"]:
    """The most likely monotone path of `tokens` (-1: wildcard, a character outside the vocabulary, which scores the
    best non-blank entry of a frame) through the emissions: one Point per frame, or None when there are fewer frames
    than tokens.  Trellis and walk back run in libmwhip.so (``mw_align_path``), in float32 like the torch code of
    whisperx's alignment.py: trellis[t + 1][j] = max(trellis[t][j] + blank(t), trellis[t][j - 1] + token_j(t)), token 0
    held from frame 0, the walk started on the last token at the last frame."""
    em = np.ascontiguousarray(emission, dtype=np.float32)
    tok = np.ascontiguousarray(tokens, dtype=np.int32)
    T, V = em.shape
    idx = np.empty(T, np.int32)
    sc = np.empty(T, np.float32)
    rc = _lib.load().mw_align_path(em.ctypes.data, T, V, tok.ctypes.data, len(tok), int(blank_id), idx.ctypes.data, sc.ctypes.data)
    if rc == -2:                                  # MW_ERR_STATE: no path
        return None
    if rc != 0:
        raise EngineError(f"mw_align_path failed ({rc}): bad argument")
    return PathArrays(idx, sc)


def merge_repeats(path: Sequence[Point], transcript: str) -> List[Segment]:
    """Runs of one token index -> one segment [first frame, last frame + 1) scored by the mean point score."""
    if isinstance(path, PathArrays):
        idx = path.token_index
        if idx.shape[0] == 0:
            return []
        starts = np.concatenate(([0], np.nonzero(idx[1:] != idx[:-1])[0] + 1))
        ends = np.concatenate((starts[1:], [idx.shape[0]]))
        sums = np.add.reduceat(path.score.astype(np.float64), starts)       # in frame order, in double, as the loop below adds them
        return [Segment(transcript[int(idx[a])], int(a), int(b), float(sm / (b - a))) for a, b, sm in zip(starts, ends, sums)]
    i1, i2, segments = 0, 0, []
    while i1 < len(path):
        while i2 < len(path) and path[i1].token_index == path[i2].token_index:
            i2 += 1
        score = sum(path[k].score for k in range(i1, i2)) / (i2 - i1)
        segments.append(Segment(transcript[path[i1].token_index], path[i1].time_index, path[i2 - 1].time_index + 1, score))
        i1 = i2
    return segments


# ------------------------------------------------------------------------------------------------
# segments -> aligned segments
# ------------------------------------------------------------------------------------------------
def sentence_spans(text: str) -> List[Tuple[int, int]]:
    """(start, end) of the sentences of `text`: a sentence ends after a run of . ? ! (or their full-width forms)
    that is followed by white space; the next one starts at the next character that is not white space.
    (whisperx uses nltk's Punkt tokenizer with a list of abbreviations; this is its basic rule without the
    learned exceptions.  Text without white space, the usual Chinese output, is one sentence in both.)"""
    spans, start, i, n = [], None, 0, len(text)
    while i < n:
        if start is None:
            if not text[i].isspace():
                start = i
            i += 1
            continue
        if text[i] in ".?!。？！" and i + 1 < n and text[i + 1].isspace():
            spans.append((start, i + 1))
            start = None
        i += 1
    if start is not None:
        end = len(text.rstrip())
        spans.append((start, max(end, start + 1)))
    return spans


def _nanmin(vals):
    vals = [v for v in vals if v is not None and not (isinstance(v, float) and math.isnan(v))]
    return min(vals) if vals else math.nan


def _nanmax(vals):
    vals = [v for v in vals if v is not None and not (isinstance(v, float) and math.isnan(v))]
    return max(vals) if vals else math.nan


def interpolate_nans(values: Sequence[float], method: str = "nearest") -> List[float]:
    """Missing entries (NaN) take the value of the nearest known entry (the earlier one on a tie) when at least two
    entries are known, else the known one everywhere; `method` "linear" interpolates on the index instead.
    Entries before the first / after the last known one copy it."""
    v = [math.nan if x is None else float(x) for x in values]
    known = [i for i, x in enumerate(v) if not math.isnan(x)]
    if not known:
        return v
    out = list(v)
    for i, x in enumerate(v):
        if not math.isnan(x):
            continue
        left = max((k for k in known if k < i), default=None)
        right = min((k for k in known if k > i), default=None)
        if left is None:
            out[i] = v[right]
        elif right is None:
            out[i] = v[left]
        elif method == "linear":
            out[i] = v[left] + (v[right] - v[left]) * (i - left) / (right - left)
        else:
            out[i] = v[left] if i - left <= right - i else v[right]
    return out


def _prepare(segment: dict, dictionary: Dict[str, int], lang: str) -> dict:
    text = segment["text"]
    lead = len(text) - len(text.lstrip())
    trail = len(text) - len(text.rstrip())
    clean_char, clean_cdx = [], []
    for cdx, ch in enumerate(text):
        c = ch.lower()
        if lang not in LANGUAGES_WITHOUT_SPACES:
            c = c.replace(" ", "|")
        if cdx < lead or cdx > len(text) - trail - 1:
            continue
        clean_char.append(c if c in dictionary else "*")      # "*": aligned as a wildcard
        clean_cdx.append(cdx)
    return {"clean_char": clean_char, "clean_cdx": clean_cdx, "sentence_spans": sentence_spans(text)}


def align(transcript: Sequence[dict], model: Aligner, align_model_metadata: dict, audio, device=None,
          interpolate_method: str = "nearest", return_char_alignments: bool = False, print_progress: bool = False,
          combined_progress: bool = False) -> dict:
    """whisperx.align (reference transcribe.py:131): segments {"text", "start", "end"} ->
    {"segments": [{"start", "end", "text", "words": [{"word", "start", "end", "score"}], ("chars")}], "word_segments": [...]}.

    Per segment: the characters of its text (outside the vocabulary: wildcard) are aligned to the emissions of
    audio[start:end]; frame indices become seconds by (end - start) / frames; characters are gathered into words
    (every character is a word for Chinese and Japanese, else split at spaces) and sentences, each sentence one
    output segment; sentences with equal times are joined."""
    if isinstance(audio, str):
        from .audio import load_audio
        audio = load_audio(audio)
    audio = np.asarray(audio, np.float32).reshape(-1)
    max_duration = len(audio) / SAMPLE_RATE
    dictionary: Dict[str, int] = align_model_metadata["dictionary"]
    lang = align_model_metadata["language"]
    blank_id = 0
    for ch, code in dictionary.items():
        if ch in ("[pad]", "<pad>"):
            blank_id = code

    def prepare(sdx, segment):
        """-> ("done", [segment as it is]) for a segment nothing can be aligned in, else ("todo", what `finish` needs)."""
        prep = _prepare(segment, dictionary, lang)
        t1, t2, text = segment["start"], segment["end"], segment["text"]
        fallback = {"start": t1, "end": t2, "text": text, "words": [], "chars": [] if return_char_alignments else None}
        if print_progress:
            print(f"Progress: {(sdx + 1) / max(len(transcript), 1) * (50 if combined_progress else 100) + (50 if combined_progress else 0):.2f}%...")
        if not prep["clean_char"]:
            print(f'Failed to align segment ("{text}"): no characters in this segment found in model dictionary, resorting to original...')
            return "done", [fallback]
        if t1 >= max_duration:
            print(f'Failed to align segment ("{text}"): original start time longer than audio duration, skipping...')
            return "done", [fallback]
        text_clean = "".join(prep["clean_char"])
        tokens = [dictionary.get(c, -1) for c in text_clean]
        wave = audio[int(t1 * SAMPLE_RATE): int(t2 * SAMPLE_RATE)]
        if len(wave) < MIN_SAMPLES:
            wave = np.pad(wave, (0, MIN_SAMPLES - len(wave)))
        return "todo", {"prep": prep, "fallback": fallback, "text_clean": text_clean, "tokens": tokens, "wave": wave, "segment": segment}

    def run_one(todo):
        """The acoustic model and the search for one segment -> (path or None, frames); an engine failure -> None."""
        wave, tokens, text = todo["wave"], todo["tokens"], todo["segment"]["text"]
        try:
            if hasattr(model, "segment_path"):
                return model.segment_path(wave, tokens, blank_id)
            emission = model.emissions(wave)                    # any object with emissions(wave) -> [frames][vocab]
            return ctc_path(emission, tokens, blank_id), emission.shape[0]
        except EngineError as e:
            # one segment the model cannot take (longer than the aligner's max_seconds, for one) must not lose the
            # others: like every other per-segment failure it keeps its original times
            print(f'Failed to align segment ("{text}"): {e}, resorting to original...')
            return None

    def finish(todo, result):
        prep, fallback, text_clean = todo["prep"], todo["fallback"], todo["text_clean"]
        segment = todo["segment"]
        t1, t2, text = segment["start"], segment["end"], segment["text"]
        if result is None:
            return [fallback]
        path, n_frames = result
        if path is None:
            print(f'Failed to align segment ("{text}"): backtrack failed, resorting to original...')
            return [fallback]
        char_segments = merge_repeats(path, text_clean)
        ratio = (t2 - t1) / (n_frames - 1) if n_frames > 1 else 0.0      # whisperx: duration / (trellis rows - 1)

        chars = []
        word_idx = 0
        pos_of = {cdx: i for i, cdx in enumerate(prep["clean_cdx"])}
        for cdx, ch in enumerate(text):
            start = end = score = None
            i = pos_of.get(cdx)
            if i is not None and i < len(char_segments):
                seg = char_segments[i]
                start, end, score = round(seg.start * ratio + t1, 3), round(seg.end * ratio + t1, 3), round(seg.score, 3)
            chars.append({"char": ch, "start": start, "end": end, "score": score, "word-idx": word_idx})
            if lang in LANGUAGES_WITHOUT_SPACES or cdx == len(text) - 1 or text[cdx + 1] == " ":
                word_idx += 1

        subsegments = []
        for sstart, send in prep["sentence_spans"]:
            cur = chars[sstart: send + 1]                       # inclusive of `send`, as whisperx selects them
            words = []
            by_word: Dict[int, list] = {}                       # word index -> its characters, words in order of appearance
            for c in cur:
                by_word.setdefault(c["word-idx"], []).append(c)
            for wchars in by_word.values():
                wtext = "".join(c["char"] for c in wchars).strip()
                if not wtext:
                    continue
                wchars = [c for c in wchars if c["char"] != " "]
                ws, we = _nanmin(c["start"] for c in wchars), _nanmax(c["end"] for c in wchars)
                scores = [c["score"] for c in wchars if c["score"] is not None]
                word = {"word": wtext}
                if not math.isnan(ws):
                    word["start"] = ws
                if not math.isnan(we):
                    word["end"] = we
                if scores:
                    word["score"] = round(sum(scores) / len(scores), 3)
                words.append(word)
            sub = {"text": text[sstart:send], "start": _nanmin(c["start"] for c in cur),
                   "end": _nanmax(c["end"] for c in cur if c["char"] != " "), "words": words}
            if return_char_alignments:
                sub["chars"] = [{k: v for k, v in c.items() if k != "word-idx" and v is not None} for c in cur]
            subsegments.append(sub)
        starts = interpolate_nans([s["start"] for s in subsegments], interpolate_method)
        ends = interpolate_nans([s["end"] for s in subsegments], interpolate_method)
        merged: Dict[Tuple[float, float], dict] = {}
        joiner = "" if lang in LANGUAGES_WITHOUT_SPACES else " "
        for s, a, b in zip(subsegments, starts, ends):
            if math.isnan(a) or math.isnan(b):
                continue                                        # nothing in the segment got a time
            key = (a, b)
            if key in merged:
                merged[key]["text"] = merged[key]["text"] + joiner + s["text"]
                merged[key]["words"] = merged[key]["words"] + s["words"]
                if return_char_alignments:
                    merged[key]["chars"] = merged[key]["chars"] + s["chars"]
            else:
                merged[key] = dict(s, start=a, end=b)
        return [merged[k] for k in sorted(merged)]

    # Segments are independent.  With the GPU model they go through it in PACKS (one pass over up to 32 segments of
    # like length, Aligner.segments_path), several packs in flight (the model runs n_streams at once, and one pack's
    # search on the host overlaps another's acoustic model); results keep the segments' order.
    prepared = [prepare(sdx, seg) for sdx, seg in enumerate(transcript)]
    parts: List[Optional[list]] = [p[1] if p[0] == "done" else None for p in prepared]
    todo = [(i, p[1]) for i, p in enumerate(prepared) if p[0] == "todo"]
    workers = max(1, int(getattr(model, "n_streams", 1)))
    if hasattr(model, "segments_path") and hasattr(model, "pack_fits"):
        packs: List[List[Tuple[int, dict]]] = []
        for i, t in sorted(todo, key=lambda it: -len(it[1]["wave"])):           # longest first: a pack's slices are its first segment's length
            cur = packs[-1] if packs else None
            if cur is not None and model.pack_fits([len(x[1]["wave"]) for x in cur] + [len(t["wave"])],
                                                   [len(x[1]["tokens"]) for x in cur] + [len(t["tokens"])]):
                cur.append((i, t))
            else:
                packs.append([(i, t)])

        def run_pack(pack):
            """A pack through the model and the search, then its segments' words and sentences (in the worker: one pack's
            host work runs beside another pack's acoustic model)."""
            if len(pack) == 1 and not model.pack_fits([len(pack[0][1]["wave"])], [len(pack[0][1]["tokens"])]):
                res = [run_one(pack[0][1])]                     # too long or too many characters for a pack: the single-segment call decides
            else:
                try:
                    res = model.segments_path([t["wave"] for _, t in pack], [t["tokens"] for _, t in pack], blank_id)
                except EngineError as e:
                    print(f"Failed to align a pack of {len(pack)} segments ({e}): aligning them one at a time...")
                    res = [run_one(t) for _, t in pack]
            return [finish(t, r) for (_, t), r in zip(pack, res)]

        if workers > 1 and len(packs) > 1:
            with ThreadPoolExecutor(max_workers=workers) as pool:
                results = list(pool.map(run_pack, packs))
        else:
            results = [run_pack(p) for p in packs]
        for pack, res in zip(packs, results):
            for (i, _), part in zip(pack, res):
                parts[i] = part
    else:
        def run_and_finish(it):
            return finish(it[1], run_one(it[1]))

        if workers > 1 and len(todo) > 1:
            with ThreadPoolExecutor(max_workers=workers) as pool:
                results = list(pool.map(run_and_finish, todo))
        else:
            results = [run_and_finish(it) for it in todo]
        for (i, _), part in zip(todo, results):
            parts[i] = part
    aligned_segments: List[dict] = [seg for part in parts for seg in part]

    word_segments: List[dict] = []
    for seg in aligned_segments:
        word_segments += seg["words"]
    return {"segments": aligned_segments, "word_segments": word_segments}
