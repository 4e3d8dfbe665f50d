"""CPU oracle of the forced-alignment step (reference transcribe.py:127-135): TEST INFRASTRUCTURE ONLY.

The reference calls ``whisperx.load_align_model(language_code="zh", ...)`` and ``whisperx.align(...)``
(transcribe.py:130-131).  Neither whisperx nor its wav2vec2 checkpoint exists offline, and the reference holds no
vectors for this step: PARITY WITH THE REFERENCE IS UNPINNED.  What is restated here:

  * the acoustic model: Hugging Face ``Wav2Vec2ForCTC`` in the variant whisperx loads for Chinese
    (jonatasgrosman/wav2vec2-large-xlsr-53-chinese-zh-cn: layer-norm feature extractor with conv bias,
    ``do_stable_layer_norm``), following transformers 5.15 models/wav2vec2/modeling_wav2vec2.py:
    conv layers :275-299, feature projection :422-434, positional conv embedding :326-379 (weight norm over
    dim 2), attention :466-548 with eager_attention_forward :438-463, encoder layer :611-654, encoder :729-802,
    CTC head :1597-1742.  Pinned against transformers itself by tests/golden/w2v_micro.npz
    (tools/make_align_golden.py).
  * the alignment: trellis, backtracking and merging of repeats as published in torchaudio's forced-alignment
    tutorial, which whisperx's alignment.py carries (restated from memory of whisperx 3.x; the pinned 3.7.6 is
    not available to check against).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
from scipy.special import erf


def _gelu(x):
    return 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))


def _layer_norm(x, g, b, eps):
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def pos_conv_weight(weights: dict) -> np.ndarray:
    """The positional convolution's weight [D][D/groups][k] with its weight norm (dim 2) folded in:
    w = g * v / ||v||, the norm taken over dims (0, 1) for every tap (modeling_wav2vec2.py:337-355)."""
    pre = "wav2vec2.encoder.pos_conv_embed.conv."
    if pre + "weight" in weights:
        return np.asarray(weights[pre + "weight"], np.float32)
    for gk, vk in ((pre + "parametrizations.weight.original0", pre + "parametrizations.weight.original1"),
                   (pre + "weight_g", pre + "weight_v")):
        if gk in weights:
            g = np.asarray(weights[gk], np.float64)
            v = np.asarray(weights[vk], np.float64)
            norm = np.sqrt((v ** 2).sum(axis=(0, 1), keepdims=True))
            return (g * v / norm).astype(np.float32)
    raise KeyError("no positional convolution weight in the state dict")


def n_frames(cfg: dict, n_samples: int) -> int:
    """Frames the feature extractor makes of n_samples (valid convolutions)."""
    n = int(n_samples)
    for k, s in zip(cfg["conv_kernel"], cfg["conv_stride"]):
        n = (n - k) // s + 1
    return max(n, 0)


class Wav2Vec2Oracle:
    """fp32 numpy forward pass of Wav2Vec2ForCTC (layer-norm feature extractor, stable layer norm)."""

    def __init__(self, cfg: dict, weights: dict):
        if cfg.get("feat_extract_norm", "layer") != "layer" or not cfg.get("do_stable_layer_norm", True):
            raise ValueError("only the layer-norm / stable-layer-norm variant (wav2vec2-large-xlsr) is restated")
        self.cfg = cfg
        self.w = {k: np.asarray(v, np.float32) for k, v in weights.items() if "parametrizations" not in k and not k.endswith(("weight_g", "weight_v"))}
        self.w["pos_conv"] = pos_conv_weight(weights)
        self.eps = float(cfg.get("layer_norm_eps", 1e-5))

    # -- feature extractor: conv -> LayerNorm over channels -> GELU (modeling_wav2vec2.py:291-299)
    def features(self, wave: np.ndarray) -> np.ndarray:
        h = np.asarray(wave, np.float32).reshape(1, -1)                    # [C_in][n]
        for i, (k, s) in enumerate(zip(self.cfg["conv_kernel"], self.cfg["conv_stride"])):
            p = f"wav2vec2.feature_extractor.conv_layers.{i}."
            w, b = self.w[p + "conv.weight"], self.w[p + "conv.bias"]      # [C_out][C_in][k]
            n_out = (h.shape[1] - k) // s + 1
            idx = np.arange(n_out)[:, None] * s + np.arange(k)[None, :]    # [n_out][k]
            patches = h[:, idx]                                            # [C_in][n_out][k]
            y = np.einsum("ctk,ock->to", patches, w, optimize=True) + b    # [n_out][C_out]
            y = _layer_norm(y, self.w[p + "layer_norm.weight"], self.w[p + "layer_norm.bias"], 1e-5)   # nn.LayerNorm default eps (:288)
            h = _gelu(y).T.astype(np.float32)
        return h.T                                                         # [T][C]

    def hidden(self, wave: np.ndarray) -> np.ndarray:
        c = self.cfg
        D, H = c["hidden_size"], c["num_attention_heads"]
        x = self.features(wave)
        x = _layer_norm(x, self.w["wav2vec2.feature_projection.layer_norm.weight"], self.w["wav2vec2.feature_projection.layer_norm.bias"], self.eps)
        x = x @ self.w["wav2vec2.feature_projection.projection.weight"].T + self.w["wav2vec2.feature_projection.projection.bias"]
        T = x.shape[0]
        # positional convolution: grouped, padding k//2 on both sides, the last output dropped when k is even, GELU
        k, G = c["num_conv_pos_embeddings"], c["num_conv_pos_embedding_groups"]
        wpos, bpos = self.w["pos_conv"], self.w["wav2vec2.encoder.pos_conv_embed.conv.bias"]
        xp = np.pad(x, ((k // 2, k // 2), (0, 0)))
        n_out = T + 2 * (k // 2) - k + 1
        idx = np.arange(n_out)[:, None] + np.arange(k)[None, :]
        pos = np.empty((n_out, D), np.float32)
        gw = D // G
        for g in range(G):
            patches = xp[:, g * gw:(g + 1) * gw][idx]                      # [n_out][k][gw]
            pos[:, g * gw:(g + 1) * gw] = np.einsum("tkc,ock->to", patches, wpos[g * gw:(g + 1) * gw], optimize=True)
        pos = pos + bpos
        if k % 2 == 0:
            pos = pos[:-1]
        x = x + _gelu(pos)
        hd = D // H
        for i in range(c["num_hidden_layers"]):
            p = f"wav2vec2.encoder.layers.{i}."
            y = _layer_norm(x, self.w[p + "layer_norm.weight"], self.w[p + "layer_norm.bias"], self.eps)
            q = (y @ self.w[p + "attention.q_proj.weight"].T + self.w[p + "attention.q_proj.bias"]).reshape(T, H, hd).transpose(1, 0, 2)
            kk = (y @ self.w[p + "attention.k_proj.weight"].T + self.w[p + "attention.k_proj.bias"]).reshape(T, H, hd).transpose(1, 0, 2)
            v = (y @ self.w[p + "attention.v_proj.weight"].T + self.w[p + "attention.v_proj.bias"]).reshape(T, H, hd).transpose(1, 0, 2)
            s = (q @ kk.transpose(0, 2, 1)) * (hd ** -0.5)
            s = s - s.max(axis=-1, keepdims=True)
            e = np.exp(s)
            a = (e / e.sum(axis=-1, keepdims=True)) @ v                    # [H][T][hd]
            a = a.transpose(1, 0, 2).reshape(T, D)
            x = x + a @ self.w[p + "attention.out_proj.weight"].T + self.w[p + "attention.out_proj.bias"]
            y = _layer_norm(x, self.w[p + "final_layer_norm.weight"], self.w[p + "final_layer_norm.bias"], self.eps)
            y = _gelu(y @ self.w[p + "feed_forward.intermediate_dense.weight"].T + self.w[p + "feed_forward.intermediate_dense.bias"])
            x = x + y @ self.w[p + "feed_forward.output_dense.weight"].T + self.w[p + "feed_forward.output_dense.bias"]
        return _layer_norm(x, self.w["wav2vec2.encoder.layer_norm.weight"], self.w["wav2vec2.encoder.layer_norm.bias"], self.eps)

    def logits(self, wave: np.ndarray) -> np.ndarray:
        return (self.hidden(wave) @ self.w["lm_head.weight"].T + self.w["lm_head.bias"]).astype(np.float32)

    def emissions(self, wave: np.ndarray) -> np.ndarray:
        """log-softmax of the CTC logits, float32 [T][vocab]."""
        z = self.logits(wave).astype(np.float64)
        z = z - z.max(axis=-1, keepdims=True)
        return (z - np.log(np.exp(z).sum(axis=-1, keepdims=True))).astype(np.float32)


# ------------------------------------------------------------------------------------------------
# alignment of a token sequence to the emissions (torchaudio forced-alignment tutorial in the form whisperx 3.x
# carries: trellis over (frames, tokens), characters outside the vocabulary as a wildcard).  Written with plain
# loops, independently of the vectorised host code it checks.
# ------------------------------------------------------------------------------------------------
@dataclass
class Point:
    token_index: int
    time_index: int
    score: float


@dataclass
class Segment:
    label: str
    start: int
    end: int
    score: float


def token_logp(frame: np.ndarray, token: int, blank_id: int) -> float:
    """A wildcard token (-1) scores the best non-blank entry of the frame."""
    if token != -1:
        return float(frame[token])
    return float(max(frame[v] for v in range(len(frame)) if v != blank_id))


def get_trellis(emission: np.ndarray, tokens: Sequence[int], blank_id: int = 0) -> np.ndarray:
    """trellis[t][j]: best log-probability of frames [0, t) with the path on token j at frame t."""
    T, J = emission.shape[0], len(tokens)
    tr = np.zeros((T, J), np.float32)
    for t in range(1, T):
        tr[t, 0] = tr[t - 1, 0] + emission[t, blank_id]
    for j in range(1, J):
        tr[0, j] = -np.inf
    for t in range(T - J + 1, T):
        if J > 1 and t >= 0:
            tr[t, 0] = np.inf
    for t in range(T - 1):
        for j in range(1, J):
            stay = np.float32(tr[t, j] + emission[t, blank_id])
            move = np.float32(tr[t, j - 1] + np.float32(token_logp(emission[t], tokens[j], blank_id)))
            tr[t + 1, j] = max(stay, move)
    return tr


def backtrack(trellis: np.ndarray, emission: np.ndarray, tokens: Sequence[int], blank_id: int = 0) -> Optional[List[Point]]:
    t, j = trellis.shape[0] - 1, trellis.shape[1] - 1
    path = [Point(j, t, float(np.exp(emission[t, blank_id])))]
    while j > 0:
        if t <= 0:
            return None
        p_stay = float(emission[t - 1, blank_id])
        p_change = token_logp(emission[t - 1], tokens[j], blank_id)
        changed = np.float32(trellis[t - 1, j - 1] + np.float32(p_change)) > np.float32(trellis[t - 1, j] + np.float32(p_stay))
        t -= 1
        if changed:
            j -= 1
        path.append(Point(j, t, float(np.exp(p_change if changed else p_stay))))
    while t > 0:
        path.append(Point(j, t - 1, float(np.exp(emission[t - 1, blank_id]))))
        t -= 1
    return path[::-1]


def merge_repeats(path: Sequence[Point], transcript: str) -> List[Segment]:
    segments, i = [], 0
    while i < len(path):
        k = i
        while k < len(path) and path[k].token_index == path[i].token_index:
            k += 1
        segments.append(Segment(transcript[path[i].token_index], path[i].time_index, path[k - 1].time_index + 1,
                                sum(p.score for p in path[i:k]) / (k - i)))
        i = k
    return segments


def best_alignment_bruteforce(emission: np.ndarray, tokens: Sequence[int], blank_id: int = 0):
    """Exhaustive search (small cases) over the change frames c_1 < ... < c_{J-1} in [0, T-2]: token j is entered
    at frame c_j and pays its emission there, the frames after c_1 that are no change pay the blank, and the stay on
    token 0 pays the blanks of frames 1..c_1 (the first column of the trellis is a cumulative sum that starts at
    frame 1).  -> (best score = trellis[T-1][J-1], its change frames)."""
    import itertools
    T, J = emission.shape[0], len(tokens)
    if J == 1:
        return float(sum(emission[u, blank_id] for u in range(1, T))), ()
    best, arg = -np.inf, None
    for changes in itertools.combinations(range(T - 1), J - 1):
        order = {c: j + 1 for j, c in enumerate(changes)}
        sc = sum(float(emission[u, blank_id]) for u in range(1, changes[0] + 1))
        for t in range(changes[0], T - 1):
            sc += token_logp(emission[t], tokens[order[t]], blank_id) if t in order else float(emission[t, blank_id])
        if sc > best:
            best, arg = sc, changes
    return best, arg
