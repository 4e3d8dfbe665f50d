"""load_model from a local Hugging Face-layout directory (config.json + weights + tokenizer.json):
file formats, prompt encoding from ``initial_prompt`` and detokenisation (the reference's
``whisperx.load_model(..., asr_options={"initial_prompt": ...})``, transcribe.py:107-113)."""
import os

import numpy as np
import pytest

import manual_whisper_amd as mw
from manual_whisper_amd import localckpt, synth
from manual_whisper_amd.dims import WhisperDims
from oracle.whisper_ref import WhisperOracle, log_mel

# micro-sized model with the real multilingual vocabulary size, so the real special-token ids apply
DIMS = WhisperDims("micro-ml", 80, 128, 2, 2, 2, 512, 51865, 1500, 448)


def test_synthetic_tokenizer_ids_and_round_trip():
    tok = localckpt.build_synthetic_tokenizer(51865)
    assert tok.token_to_id("<|endoftext|>") == 50257 and tok.token_to_id("<|startoftranscript|>") == 50258
    assert tok.token_to_id("<|zh|>") == 50260 and tok.token_to_id("<|transcribe|>") == 50359
    assert tok.token_to_id("<|startofprev|>") == 50361 and tok.token_to_id("<|notimestamps|>") == 50363
    assert tok.token_to_id("<|30.00|>") == 51864
    text = " 中文会议讨论记录 NFC Node ID"
    ids = tok.encode(text, add_special_tokens=False).ids
    assert len(ids) == len(text.encode("utf-8")) and max(ids) < 256
    assert tok.decode(ids) == text
    assert tok.decode([50258, 50260] + ids + [50257], skip_special_tokens=True) == text
    v3 = localckpt.build_synthetic_tokenizer(51866)                 # large-v3: one more language, later ids move up
    assert v3.token_to_id("<|yue|>") == 50358 and v3.token_to_id("<|notimestamps|>") == 50364


def test_missing_model_is_an_error_not_a_download(tmp_path):
    with pytest.raises(FileNotFoundError):
        mw.load_model("large-v3", "cuda")
    localckpt.write_synthetic_checkpoint(str(tmp_path / "m"), mw.get_dims("micro"), tokenizer=False)
    assert sorted(os.listdir(tmp_path / "m")) == ["config.json", "model.safetensors"]


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,dtype", [("safetensors", "float16"), ("bin", "bfloat16")])
def test_load_model_from_directory(tmp_path, fmt, dtype):
    path = str(tmp_path / "ckpt")
    stored = localckpt.write_synthetic_checkpoint(path, DIMS, seed=77, fmt=fmt, dtype=dtype)
    prompt_text = "微信 APP"
    model = mw.load_model(path, "cuda", compute_type="float16", language="zh",
                          asr_options={"initial_prompt": prompt_text, "max_new_tokens": 8}, max_batch=2)
    try:
        assert model.dims.vocab == 51865 and model.dims.d_model == 128 and model.tokenizer is not None
        opts = model.decode_options("zh")
        text_ids = model.tokenizer.encode(" " + prompt_text, add_special_tokens=False).ids
        assert list(opts.prompt) == [50361] + text_ids + [50258, 50260, 50359, 50363]
        assert opts.eot == 50257 and opts.max_new == 8
        pcm = synth.make_pcm_np(2, seed=5).reshape(-1)[: 16000 * 35]            # 35 s -> two fixed chunks
        res = model.transcribe(pcm, batch_size=2, return_tokens=True)
        assert res["language"] == "zh" and [(s["start"], s["end"]) for s in res["segments"]] == [(0.0, 30.0), (30.0, 35.0)]
        for s in res["segments"]:
            assert s["text"] == model.tokenizer.decode(s["tokens"], skip_special_tokens=True)
            assert all(0 <= t < 51865 for t in s["tokens"])
        # the same ids from the oracle running on the weights as stored in the file
        orc = WhisperOracle(DIMS, stored)
        chunks, _ = model.windows(pcm)
        enc = orc.encode(log_mel(chunks, DIMS.n_mels))
        want, lens, _ = orc.greedy(enc, list(opts.prompt), opts.max_new, opts.eot, opts.effective_suppress(), opts.begin_suppress)
        from test_gpu_parity import assert_ids_match_up_to_ties
        got = np.full_like(want, opts.eot)
        for i, s in enumerate(res["segments"]):
            got[i, : len(s["tokens"])] = s["tokens"]
        assert_ids_match_up_to_ties(got, want, orc.last_margins)
    finally:
        model.close()
