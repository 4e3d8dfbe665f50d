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


# ---------------------------------------------------------------------------------------------
# fp8 (e4m3) weight format of the oracle
# ---------------------------------------------------------------------------------------------
def test_e4m3_rounding_agrees_with_torch_float8():
    """The oracle's e4m3 rounding against torch.float8_e4m3fn: every code, every midpoint between
    neighbouring codes (ties to even), random values over the whole range, saturation at 448."""
    import torch
    from oracle.whisper_ref import e4m3_round
    codes = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).float().numpy()
    fin = np.unique(codes[np.isfinite(codes)])
    assert fin.size == 253 and fin.max() == 448.0                 # 254 finite codes, +0 and -0 coincide
    assert (e4m3_round(fin) == fin).all()
    mid = ((fin[1:] + fin[:-1]) / 2).astype(np.float32)
    rng = np.random.default_rng(0)
    x = np.concatenate([mid, np.nextafter(mid, np.float32(1e9)), np.nextafter(mid, np.float32(-1e9)),
                        rng.normal(0, 1, 50000), rng.normal(0, 120, 50000), rng.uniform(-0.03, 0.03, 50000)]).astype(np.float32)
    x = np.clip(x, -448, 448)
    want = torch.from_numpy(x).to(torch.float8_e4m3fn).float().numpy()
    assert (e4m3_round(x) == want).all()
    assert (e4m3_round(np.float32([500.0, -1e6, 448.1])) == np.float32([448, -448, 448])).all()


def test_fp8_row_quantisation_and_weight_selection():
    from oracle.whisper_ref import fake_quant_rows_e4m3, is_fp8_weight, quantize_rows_e4m3
    rng = np.random.default_rng(1)
    w = rng.normal(0, 0.02, (64, 256)).astype(np.float32)
    w[5] = 0
    q, s = quantize_rows_e4m3(w)
    assert s.shape == (64, 1) and s[5, 0] == 1.0 and (q[5] == 0).all()
    assert np.abs(q).max() == 448.0 and (np.abs(q).max(axis=1)[np.arange(64) != 5] == 448.0).all()
    rel = np.abs(fake_quant_rows_e4m3(w) - w)[np.abs(w) > 1e-3] / np.abs(w)[np.abs(w) > 1e-3]
    assert rel.max() <= 2.0 ** -4 + 1e-6                               # half a unit in the last of 3 mantissa bits
    dims = mw.get_dims("micro")
    names = synth.make_weights_np(dims, 1).keys()
    picked = [n for n in names if is_fp8_weight(n)]
    # per encoder layer q,k,v,out,fc1,fc2; per decoder layer those plus the four cross-attention projections
    assert len(picked) == dims.enc_layers * 6 + dims.dec_layers * 10
    assert not any("conv" in n or "embed" in n or "layer_norm" in n or n.endswith(".bias") for n in picked)


def test_fp8_oracle_stays_near_the_unquantised_one():
    from oracle.whisper_ref import WhisperOracle, log_mel
    dims = mw.get_dims("micro")
    w = synth.make_weights_np(dims, 1234)
    mel = log_mel(synth.make_pcm_np(1, 3), dims.n_mels)
    e32 = WhisperOracle(dims, w).encode(mel)
    e8 = WhisperOracle(dims, w, weight_format="fp8").encode(mel)
    rms = float(np.sqrt(np.mean((e8 - e32) ** 2)))
    assert 0.01 < rms < 0.12, rms                                      # quantisation is in effect and is a perturbation
