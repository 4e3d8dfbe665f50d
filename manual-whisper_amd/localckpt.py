"""Write a local checkpoint directory in Hugging Face layout from the deterministic synthetic
weights: ``config.json`` + ``model.safetensors`` (or ``pytorch_model.bin``) + ``tokenizer.json``.

``load_model`` reads exactly this layout (the reference's ``whisperx.load_model(MODEL_SIZE, ...)``,
transcribe.py:107-113, resolves a model NAME by downloading; nothing can be downloaded here, so a
real deployment points at such a directory).  No pretrained weights or vocabulary exist offline;
this module gives tests and smoke runs a directory with the real file formats, the real special
token ids of the multilingual Whisper vocabulary, and a byte-level tokenizer whose ordinary tokens
are synthetic (256 byte tokens + numbered fillers), so prompt encoding and detokenisation run
through the same code as with a released tokenizer.json.
"""
from __future__ import annotations

import json
import os

import numpy as np

from .dims import WhisperDims
from . import synth

# language codes in the order of the multilingual vocabulary (<|en|> = 50259, <|zh|> = 50260, ...)
_LANGS = ("en zh de es ru ko fr ja pt tr pl ca nl ar sv it id hi fi vi he uk el ms cs ro da hu ta no th ur hr bg lt la mi "
          "ml cy sk te fa lv bn sr az sl kn et mk br eu is hy ne mn bs kk sq sw gl mr pa si km sn yo so af oc ka be tg sd "
          "gu am yi lo uz fo ht ps tk nn mt sa lb my bo tl mg as tt haw ln ha ba jw su yue").split()


def special_tokens(n_vocab: int) -> list:
    """Special tokens of the multilingual vocabulary from id 50257 upwards (51865 ids, or 51866 with <|yue|>)."""
    n_lang = 100 if n_vocab >= 51866 else 99
    toks = ["<|endoftext|>", "<|startoftranscript|>"] + [f"<|{c}|>" for c in _LANGS[:n_lang]]
    toks += ["<|translate|>", "<|transcribe|>", "<|startoflm|>", "<|startofprev|>", "<|nospeech|>", "<|notimestamps|>"]
    toks += [f"<|{i * 0.02:.2f}|>" for i in range(1501)]
    assert 50257 + len(toks) == n_vocab, (len(toks), n_vocab)
    return toks


def build_synthetic_tokenizer(n_vocab: int):
    """Byte-level BPE without merges: text encodes to one token per UTF-8 byte (ids 0..255); ids
    256..50256 are fillers ``t<id>``; the special tokens sit at their real ids."""
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers
    alphabet = sorted(pre_tokenizers.ByteLevel.alphabet())
    vocab = {ch: i for i, ch in enumerate(alphabet)}
    for i in range(256, 50257):
        vocab[f"t{i}"] = i
    tok = Tokenizer(models.BPE(vocab=vocab, merges=[]))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)
    tok.decoder = decoders.ByteLevel()
    tok.add_special_tokens(special_tokens(n_vocab))
    assert tok.get_vocab_size(with_added_tokens=True) == n_vocab
    return tok


def write_synthetic_checkpoint(path: str, dims: WhisperDims, seed: int = 1234, fmt: str = "safetensors",
                               dtype: str = "float16", tokenizer: bool = True) -> dict:
    """-> the state dict as written (numpy, in the stored precision widened to float32)."""
    os.makedirs(path, exist_ok=True)
    cfg = {"model_type": "whisper", "num_mel_bins": dims.n_mels, "d_model": dims.d_model,
           "encoder_attention_heads": dims.n_heads, "decoder_attention_heads": dims.n_heads,
           "encoder_layers": dims.enc_layers, "decoder_layers": dims.dec_layers,
           "encoder_ffn_dim": dims.ffn_dim, "decoder_ffn_dim": dims.ffn_dim, "vocab_size": dims.vocab,
           "max_source_positions": dims.n_audio_ctx, "max_target_positions": dims.n_text_ctx}
    with open(os.path.join(path, "config.json"), "w", encoding="utf-8") as f:
        json.dump(cfg, f, indent=1)
    w = synth.make_weights_np(dims, seed)
    if fmt == "safetensors":
        if dtype != "float16":
            raise ValueError("safetensors output is written as float16")
        from safetensors.numpy import save_file
        stored = {k: np.ascontiguousarray(v.astype(np.float16)) for k, v in w.items()}
        save_file(stored, os.path.join(path, "model.safetensors"))
        back = {k: v.astype(np.float32) for k, v in stored.items()}
    elif fmt == "bin":
        import torch
        tdt = {"float16": torch.float16, "bfloat16": torch.bfloat16, "float32": torch.float32}[dtype]
        sd = {k: torch.from_numpy(np.ascontiguousarray(v)).to(tdt) for k, v in w.items()}
        sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]          # tied, as torch checkpoints store it
        torch.save(sd, os.path.join(path, "pytorch_model.bin"))
        back = {k: v.float().numpy() for k, v in sd.items() if k != "proj_out.weight"}
    else:
        raise ValueError(fmt)
    if tokenizer:
        build_synthetic_tokenizer(dims.vocab).save(os.path.join(path, "tokenizer.json"))
    return back


# ---------------------------------------------------------------------------------------------
# wav2vec2 CTC checkpoints for the forced-alignment step (reference transcribe.py:130: whisperx.load_align_model
# resolves a model NAME by downloading; here a local directory in Hugging Face layout: config.json,
# model.safetensors, vocab.json).  The variant is the one whisperx loads for Chinese
# (wav2vec2-large-xlsr-53: layer-norm feature extractor with conv bias, stable layer norm).
# ---------------------------------------------------------------------------------------------
W2V_CONFIGS = {
    # real structure (seven convolutions, total stride 320) at toy widths; group width D / groups = 64
    "w2v-micro": dict(vocab_size=40, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                      conv_dim=[64] * 7, conv_kernel=[10, 3, 3, 3, 3, 2, 2], conv_stride=[5, 2, 2, 2, 2, 2, 2],
                      num_conv_pos_embeddings=32, num_conv_pos_embedding_groups=2),
    # wav2vec2-large-xlsr-53 (the architecture behind whisperx's default Chinese alignment model)
    "w2v-large": dict(vocab_size=3503, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                      conv_dim=[512] * 7, conv_kernel=[10, 3, 3, 3, 3, 2, 2], conv_stride=[5, 2, 2, 2, 2, 2, 2],
                      num_conv_pos_embeddings=128, num_conv_pos_embedding_groups=16),
}


def wav2vec2_config(name: str) -> dict:
    cfg = dict(W2V_CONFIGS[name])
    cfg.update(model_type="wav2vec2", conv_bias=True, feat_extract_norm="layer", do_stable_layer_norm=True,
               feat_extract_activation="gelu", hidden_act="gelu", layer_norm_eps=1e-5, num_feat_extract_layers=len(cfg["conv_dim"]),
               architectures=["Wav2Vec2ForCTC"], pad_token_id=0)
    return cfg


def make_wav2vec2_weights_np(cfg: dict, seed: int = 4242) -> dict:
    """Deterministic synthetic state dict (Hugging Face names; matrices fp16-representable, the positional
    convolution in its weight-norm parametrisation)."""
    import math
    sq3 = 2.0 * math.sqrt(3.0)
    D, F, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    out = {}

    def put(name, shape, scale, kind="mat", offset=0.0):
        n = int(np.prod(shape))
        a = synth.hash_uniform_np(n, name, seed) * np.float32(scale) + np.float32(offset)
        if kind == "mat":
            a = a.astype(np.float16).astype(np.float32)
        out[name] = a.reshape(shape).astype(np.float32)

    c_in = 1
    for i, (c, k) in enumerate(zip(cfg["conv_dim"], cfg["conv_kernel"])):
        p = f"wav2vec2.feature_extractor.conv_layers.{i}."
        put(p + "conv.weight", (c, c_in, k), 1.5 * sq3 / math.sqrt(c_in * k))
        put(p + "conv.bias", (c,), 0.1, "vec")
        put(p + "layer_norm.weight", (c,), 0.2, "vec", 1.0)
        put(p + "layer_norm.bias", (c,), 0.1, "vec")
        c_in = c
    put("wav2vec2.feature_projection.layer_norm.weight", (c_in,), 0.2, "vec", 1.0)
    put("wav2vec2.feature_projection.layer_norm.bias", (c_in,), 0.1, "vec")
    put("wav2vec2.feature_projection.projection.weight", (D, c_in), sq3 / math.sqrt(c_in))
    put("wav2vec2.feature_projection.projection.bias", (D,), 0.1, "vec")
    k, G = cfg["num_conv_pos_embeddings"], cfg["num_conv_pos_embedding_groups"]
    pc = "wav2vec2.encoder.pos_conv_embed.conv."
    put(pc + "parametrizations.weight.original1", (D, D // G, k), sq3 / math.sqrt(D // G * k), "vec")
    put(pc + "parametrizations.weight.original0", (1, 1, k), 0.4, "vec", 1.0)
    put(pc + "bias", (D,), 0.1, "vec")
    res = 0.8 * min(1.0, math.sqrt(4.0 / (2 * cfg["num_hidden_layers"])))
    for i in range(cfg["num_hidden_layers"]):
        p = f"wav2vec2.encoder.layers.{i}."
        for proj, gain in (("q_proj", 2.0), ("k_proj", 2.0), ("v_proj", 0.8), ("out_proj", res)):
            put(p + f"attention.{proj}.weight", (D, D), gain * sq3 / math.sqrt(D))
            put(p + f"attention.{proj}.bias", (D,), 0.1, "vec")
        for ln in ("layer_norm", "final_layer_norm"):
            put(p + ln + ".weight", (D,), 0.2, "vec", 1.0)
            put(p + ln + ".bias", (D,), 0.1, "vec")
        put(p + "feed_forward.intermediate_dense.weight", (F, D), 0.8 * sq3 / math.sqrt(D))
        put(p + "feed_forward.intermediate_dense.bias", (F,), 0.1, "vec")
        put(p + "feed_forward.output_dense.weight", (D, F), res * sq3 / math.sqrt(F))
        put(p + "feed_forward.output_dense.bias", (D,), 0.1, "vec")
    put("wav2vec2.encoder.layer_norm.weight", (D,), 0.2, "vec", 1.0)
    put("wav2vec2.encoder.layer_norm.bias", (D,), 0.1, "vec")
    put("lm_head.weight", (V, D), 3.0 * sq3 / math.sqrt(D))
    put("lm_head.bias", (V,), 0.1, "vec")
    return out


def synthetic_ctc_vocab(n_vocab: int) -> dict:
    """A character vocabulary in the layout of the wav2vec2 CTC models: <pad> = 0 (the CTC blank), <s>, </s>, <unk>,
    the word separator "|", then lower-case letters, digits and CJK characters up to n_vocab entries."""
    chars = ["<pad>", "<s>", "</s>", "<unk>", "|"] + list("abcdefghijklmnopqrstuvwxyz'") + list("0123456789")
    code = 0x4E00
    while len(chars) < n_vocab:
        chars.append(chr(code))
        code += 1
    return {c: i for i, c in enumerate(chars[:n_vocab])}


def write_synthetic_wav2vec2(path: str, name: str = "w2v-micro", seed: int = 4242) -> dict:
    """config.json + model.safetensors (float32) + vocab.json; -> the state dict as written."""
    from safetensors.numpy import save_file
    os.makedirs(path, exist_ok=True)
    cfg = wav2vec2_config(name)
    with open(os.path.join(path, "config.json"), "w", encoding="utf-8") as f:
        json.dump(cfg, f, indent=1)
    w = make_wav2vec2_weights_np(cfg, seed)
    save_file({k: np.ascontiguousarray(v) for k, v in w.items()}, os.path.join(path, "model.safetensors"))
    with open(os.path.join(path, "vocab.json"), "w", encoding="utf-8") as f:
        json.dump(synthetic_ctc_vocab(cfg["vocab_size"]), f, ensure_ascii=False)
    return w
