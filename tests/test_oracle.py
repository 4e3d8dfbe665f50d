"""The numpy oracle against fixtures generated from transformers' Whisper (tools/make_golden.py).

This is what pins oracle/whisper_ref.py; the GPU parity tests then compare the HIP path with the
oracle on the same inputs.
"""
import os

import numpy as np
import pytest

import manual_whisper_amd as mw
from manual_whisper_amd import synth
from oracle.whisper_ref import WhisperOracle, log_mel, mel_filter_bank


@pytest.fixture(scope="module", params=["micro", "tiny"])
def setup(request, golden_dir):
    g = np.load(os.path.join(golden_dir, request.param + ".npz"))
    dims = mw.get_dims(request.param)
    w = synth.make_weights_np(dims, int(g["seed"]))
    pcm = synth.make_pcm_np(int(g["batch"]), int(g["pcm_seed"]))
    mel = log_mel(pcm, dims.n_mels)
    orc = WhisperOracle(dims, w)
    enc = orc.encode(mel)
    return dict(g=g, dims=dims, orc=orc, mel=mel, enc=enc)


def test_mel_filter_bank_shape_and_norm():
    for n in (80, 128):
        fb = mel_filter_bank(n)
        assert fb.shape == (201, n)
        assert (fb >= 0).all() and (fb.max(axis=0) > 0).all()


def test_log_mel_matches_fixture(setup):
    g, mel = setup["g"], setup["mel"]
    assert mel.dtype == np.float32
    np.testing.assert_allclose(mel[:, :, ::16], g["mel_sub"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(mel.max(axis=(1, 2)), g["mel_max"], atol=2e-5)
    np.testing.assert_allclose(mel.mean(axis=(1, 2)), g["mel_mean"], atol=2e-5)


def test_encoder_matches_fixture(setup):
    g, enc = setup["g"], setup["enc"]
    np.testing.assert_allclose(enc[:, ::25, :], g["enc_sub"], atol=5e-5, rtol=0)


def test_greedy_ids_and_logits_match_fixture(setup):
    g, orc, enc = setup["g"], setup["orc"], setup["enc"]
    max_new = g["ids"].shape[1]
    ids, lens, logits = orc.greedy(enc, list(g["prompt"]), max_new, int(g["eot"]), g["suppress"], g["begin_suppress"],
                                   return_logits=True)
    assert (ids == g["ids"]).all()
    assert (lens == g["lens"]).all()
    np.testing.assert_allclose(logits[:, :, g["probe_cols"]], g["probe_val"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(np.take_along_axis(logits, g["top_idx"], 2), g["top_val"], atol=1e-4, rtol=0)


def test_forced_logits_match_prompt_fixture(setup):
    g, orc, enc = setup["g"], setup["orc"], setup["enc"]
    B = enc.shape[0]
    toks = np.tile(np.asarray(g["prompt"], np.int64)[None], (B, 1))
    lg = orc.forced_logits(enc, toks)
    np.testing.assert_allclose(lg[:, :, g["probe_cols"]], g["prompt_probe"], atol=1e-4, rtol=0)


def test_eot_stops_a_sequence():
    """EOT handling: force EOT to win by suppressing everything else for one sequence's first token."""
    dims = mw.get_dims("micro")
    w = synth.make_weights_np(dims, 7)
    orc = WhisperOracle(dims, w)
    enc = np.zeros((1, dims.n_audio_ctx, dims.d_model), np.float32)
    eot = 5
    ids, lens, _ = orc.greedy(enc, [1, 2], 6, eot, suppress=[i for i in range(dims.vocab) if i != eot])
    assert (ids == eot).all() and lens[0] == 0


def test_synth_numpy_equals_torch():
    dims = mw.get_dims("micro")
    a = synth.make_weights_np(dims, 99)
    b = synth.make_weights_torch(dims, 99)
    for k in a:
        assert (a[k] == b[k].float().numpy()).all(), k
