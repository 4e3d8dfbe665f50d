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
