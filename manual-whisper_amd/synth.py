"""Deterministic synthetic Whisper checkpoints and audio.

No pretrained weights, vocabulary or audio exist offline, so parity and
throughput runs use tensors produced by a 32-bit integer hash of
(tensor name, element index).  The same hash is written twice, for numpy
and for torch (any device), and both produce bit-identical float32 values:
every step is integer arithmetic or a single IEEE float32 operation.

Matrices are rounded through float16, as a released fp16 checkpoint would be,
so the fp32 CPU oracle and the fp16 HIP engine start from identical values.
Tensor names and shapes follow the Hugging Face Whisper state-dict layout
(``model.encoder.conv1.weight`` ...), which is what a real local checkpoint
handed to ``load_model`` contains.
"""
from __future__ import annotations

import math
import zlib

import numpy as np

from .dims import WhisperDims, CHUNK_SAMPLES

_M1 = 0x7FEB352D
_M2 = 0x846CA68B
_MASK = 0xFFFFFFFF


def _name_seed(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & _MASK


def hash_uniform_np(n: int, name: str, seed: int) -> np.ndarray:
    """float32 uniform in [-0.5, 0.5), element i = hash(i ^ name_seed)."""
    x = np.arange(n, dtype=np.uint64)
    x ^= np.uint64(_name_seed(name, seed))
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(_M1)) & np.uint64(_MASK)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(_M2)) & np.uint64(_MASK)
    x ^= x >> np.uint64(16)
    u = (x >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -24)
    return u - np.float32(0.5)


def hash_uniform_torch(n: int, name: str, seed: int, device):
    import torch
    x = torch.arange(n, dtype=torch.int64, device=device)
    x ^= _name_seed(name, seed)
    x ^= x >> 16
    x = (x * _M1) & _MASK
    x ^= x >> 15
    x = (x * _M2) & _MASK
    x ^= x >> 16
    u = (x >> 8).to(torch.float32) * (2.0 ** -24)
    return u - 0.5


def sinusoids_np(length: int, channels: int) -> np.ndarray:
    """Encoder position table (transformers modeling_whisper.py:55-64)."""
    inc = math.log(10000.0) / (channels // 2 - 1)
    inv = np.exp(-inc * np.arange(channels // 2, dtype=np.float32)).astype(np.float32)
    t = np.arange(length, dtype=np.float32)[:, None] * inv[None, :]
    return np.concatenate([np.sin(t), np.cos(t)], axis=1).astype(np.float32)


def tensor_specs(d: WhisperDims):
    """Yield (name, shape, kind, scale).  kind: 'mat' (fp16-rounded), 'vec', 'gamma', 'sin'."""
    D, F = d.d_model, d.ffn_dim
    sq3 = 2.0 * math.sqrt(3.0)          # uniform(-.5,.5)*sq3 has unit variance

    def lin(prefix, out_f, in_f, bias=True, gain=0.8):
        yield prefix + ".weight", (out_f, in_f), "mat", gain * sq3 / math.sqrt(in_f)
        if bias:
            yield prefix + ".bias", (out_f,), "vec", 0.1

    def ln(prefix):
        yield prefix + ".weight", (D,), "gamma", 0.2
        yield prefix + ".bias", (D,), "vec", 0.1

    def attn(prefix, res_gain):
        # q/k gain 2: attention logits of std ~4, peaked enough that outputs depend on WHICH frames
        # or tokens are attended (near-uniform attention makes every chunk decode to the same ids)
        yield from lin(prefix + ".q_proj", D, D, gain=2.0)
        yield from lin(prefix + ".k_proj", D, D, bias=False, gain=2.0)
        yield from lin(prefix + ".v_proj", D, D)
        yield from lin(prefix + ".out_proj", D, D, gain=res_gain)

    # Branch outputs shrink with depth (as in GPT-2 style initialisation): otherwise the residual
    # stream of a 32-layer model is a sum of ~100 nearly input-independent vectors and every chunk
    # decodes to the same ids.
    enc_res = 0.8 * min(1.0, math.sqrt(4.0 / (2 * d.enc_layers)))
    dec_res = 0.8 * min(1.0, math.sqrt(4.0 / (3 * d.dec_layers)))
    e = "model.encoder"
    yield e + ".conv1.weight", (D, d.n_mels, 3), "mat", 2.0 * sq3 / math.sqrt(3 * d.n_mels)
    yield e + ".conv1.bias", (D,), "vec", 0.1
    yield e + ".conv2.weight", (D, D, 3), "mat", 2.0 * sq3 / math.sqrt(3 * D)
    yield e + ".conv2.bias", (D,), "vec", 0.1
    yield e + ".embed_positions.weight", (d.n_audio_ctx, D), "sin", 1.0
    for i in range(d.enc_layers):
        p = f"{e}.layers.{i}"
        yield from attn(p + ".self_attn", enc_res)
        yield from ln(p + ".self_attn_layer_norm")
        yield from lin(p + ".fc1", F, D)
        yield from lin(p + ".fc2", D, F, gain=enc_res)
        yield from ln(p + ".final_layer_norm")
    yield from ln(e + ".layer_norm")

    q = "model.decoder"
    # Tied embeddings give the fed token's own logit a +|E_tok|^2/sigma bonus; keep it near
    # two standard deviations of the other logits so greedy output does not lock onto repeats.
    sigma_x = math.sqrt(1.0 + 0.64 * min(3 * d.dec_layers, 4))
    yield q + ".embed_tokens.weight", (d.vocab, D), "mat", 1.0 * sigma_x * sq3 / math.sqrt(D)
    yield q + ".embed_positions.weight", (d.n_text_ctx, D), "mat", 1.0 * sq3
    for i in range(d.dec_layers):
        p = f"{q}.layers.{i}"
        yield from attn(p + ".self_attn", dec_res)
        yield from ln(p + ".self_attn_layer_norm")
        yield from attn(p + ".encoder_attn", dec_res)
        yield from ln(p + ".encoder_attn_layer_norm")
        yield from lin(p + ".fc1", F, D)
        yield from lin(p + ".fc2", D, F, gain=dec_res)
        yield from ln(p + ".final_layer_norm")
    yield from ln(q + ".layer_norm")


def make_weights_np(d: WhisperDims, seed: int = 1234) -> dict:
    out = {}
    for name, shape, kind, scale in tensor_specs(d):
        n = int(np.prod(shape))
        if kind == "sin":
            w = sinusoids_np(shape[0], shape[1]).astype(np.float16).astype(np.float32)
        else:
            w = hash_uniform_np(n, name, seed) * np.float32(scale)
            if kind == "gamma":
                w = w + np.float32(1.0)
            if kind == "mat":
                w = w.astype(np.float16).astype(np.float32)
        out[name] = w.reshape(shape)
    return out


def make_weights_torch(d: WhisperDims, seed: int = 1234, device="cpu") -> dict:
    import torch
    out = {}
    for name, shape, kind, scale in tensor_specs(d):
        n = int(np.prod(shape))
        if kind == "sin":
            w = torch.from_numpy(sinusoids_np(shape[0], shape[1])).to(device).half()
        else:
            w = hash_uniform_torch(n, name, seed, device) * scale
            if kind == "gamma":
                w = w + 1.0
            if kind == "mat":
                w = w.half()            # kept as fp16: the engine stores matrices in fp16
        out[name] = w.reshape(shape)
    return out


def _voices(b: int, seed: int):
    """Three frequency-swept, amplitude-modulated voices per chunk; parameters depend on (chunk, seed)."""
    out = []
    for v in range(3):
        h = (b * 2654435761 + seed * 40503 + v * 97) & 0xFFFFFFFF
        f0 = 150.0 + (h % 2900)                      # start frequency, Hz
        sweep = ((h >> 8) % 1200) - 600.0            # Hz over the chunk
        am = 0.3 + ((h >> 16) % 50) / 10.0           # envelope rate, Hz
        amp = 0.04 + ((h >> 24) % 8) / 100.0
        out.append((f0, sweep, am, amp))
    return out


def make_pcm_np(batch: int, seed: int = 0, n: int = CHUNK_SAMPLES) -> np.ndarray:
    """Synthetic audio: a low hash-noise floor (std 0.01) plus three swept, amplitude-modulated
    tones whose parameters differ per chunk, so that chunks have clearly different spectrograms."""
    t = np.arange(n, dtype=np.float64) / 16000.0
    rows = []
    for b in range(batch):
        x = (hash_uniform_np(n, f"pcm.{b}", seed) * np.float32(0.01 * 2.0 * math.sqrt(3.0))).astype(np.float64)
        for f0, sweep, am, amp in _voices(b, seed):
            phase = 2 * np.pi * (f0 * t + 0.5 * sweep * t * t / 30.0)
            x = x + amp * (0.5 + 0.5 * np.sin(2 * np.pi * am * t)) * np.sin(phase)
        rows.append(x.astype(np.float32))
    return np.stack(rows)


def make_pcm_torch(batch: int, seed: int = 0, device="cpu", n: int = CHUNK_SAMPLES):
    """Same construction as make_pcm_np, built on `device` (may differ in the last float32 bit)."""
    import torch
    t = torch.arange(n, dtype=torch.float64, device=device) / 16000.0
    rows = []
    for b in range(batch):
        x = (hash_uniform_torch(n, f"pcm.{b}", seed, device) * (0.01 * 2.0 * math.sqrt(3.0))).to(torch.float64)
        for f0, sweep, am, amp in _voices(b, seed):
            phase = 2 * math.pi * (f0 * t + 0.5 * sweep * t * t / 30.0)
            x = x + amp * (0.5 + 0.5 * torch.sin(2 * math.pi * am * t)) * torch.sin(phase)
        rows.append(x.to(torch.float32))
    return torch.stack(rows).contiguous()
