"""fp8 (OCP e4m3) weight format, BASELINE config 5: the HIP engine against the oracle's restatement
of the same scheme (oracle/whisper_ref.py: per-row scales = max|.|/448, round to nearest even).

The oracle models every quantisation the engine performs (weights of the layers' Linears; LayerNorm
rows ahead of the encoder's QKV / MLP-in products; encoder output rows ahead of the cross K/V
projection).  Two kinds of check follow from how rounding to 3 mantissa bits behaves:

  * where engine and oracle quantise THE SAME fp32 numbers the codes are identical and the fp16
    tolerances of test_gpu_parity.py apply.  That is the case for all weights, and for the whole
    decoder once the oracle is handed the engine's own encoder output (mw_encode returns exactly the
    fp32 rows it quantised for the cross K/V projection): logits within LOGIT_ATOL, greedy ids
    identical up to oracle near-ties.
  * inside the encoder the two residual streams differ by fp16 rounding (~1e-3), and an activation
    that sits within that distance of an e4m3 rounding boundary lands on the other code: a 6 % jump
    of that element.  The oracle shows the same sensitivity to a 2e-4 perturbation of its own input
    (rms 0.013-0.021, max 0.13 on the encoder output, against 4e-4 for the unquantised model), so the
    encoder output is held to a fraction of the quantisation noise itself, measured in the same
    test as the distance between the fp8 oracle and the unquantised oracle (rms ~0.05, max ~0.3).
The fp8 products themselves are checked exactly (integer-valued sums of e4m3 products against a
software decode) by tools/gemm_bench ... f8.
"""
import numpy as np
import pytest

import manual_whisper_amd as mw
from manual_whisper_amd import synth
from manual_whisper_amd.decode_options import DecodeOptions, synthetic_options
from oracle.whisper_ref import WhisperOracle, log_mel

from test_gpu_parity import ENC_ATOL, LOGIT_ATOL, TIE_GAP, assert_ids_match_up_to_ties

pytestmark = pytest.mark.gpu


def _fp8_engine(dims, weights, max_batch):
    from manual_whisper_amd.engine import Engine
    eng = Engine(dims, device=0, max_batch=max_batch, weight_format="fp8")
    eng.load_weights(weights)
    return eng


@pytest.fixture(scope="module", params=["micro", "tiny"])
def small(request):
    dims = mw.get_dims(request.param)
    w = synth.make_weights_np(dims, 1234)
    B = 3
    pcm = synth.make_pcm_np(B, 77)
    orc = WhisperOracle(dims, w, weight_format="fp8")
    mel_ref = log_mel(pcm, dims.n_mels)
    enc_ref = orc.encode(mel_ref)
    eng = _fp8_engine(dims, w, 4)
    yield dict(dims=dims, w=w, B=B, pcm=pcm, orc=orc, mel_ref=mel_ref, enc_ref=enc_ref, eng=eng)
    eng.close()


def _rms(a):
    return float(np.sqrt(np.mean(np.square(a, dtype=np.float64))))


def test_fp8_encoder_within_quantisation_noise(small):
    eng, B, dims = small["eng"], small["B"], small["dims"]
    enc = eng.encode(B, mel=small["mel_ref"])
    enc_f32 = WhisperOracle(dims, small["w"]).encode(small["mel_ref"])
    noise = small["enc_ref"] - enc_f32                     # what quantisation itself does to the encoder output
    d = enc - small["enc_ref"]
    assert _rms(noise) > 0.02, "the fp8 oracle does not differ from the unquantised one: quantisation not in effect?"
    assert _rms(d) < 0.5 * _rms(noise), (_rms(d), _rms(noise))
    assert np.abs(d).max() < np.abs(noise).max(), (np.abs(d).max(), np.abs(noise).max())
    # and the engine really quantises: it is closer to the fp8 oracle than to the unquantised one
    assert _rms(enc - enc_f32) > 1.5 * _rms(d)


def test_fp8_decoder_logits_vs_oracle_on_engine_encoder_output(small):
    eng, orc, B, dims = small["eng"], small["orc"], small["B"], small["dims"]
    opts = synthetic_options(dims, max_new=4)
    enc = eng.encode(B, mel=small["mel_ref"])
    rng = np.random.default_rng(6)
    toks = np.concatenate([np.tile(np.asarray(opts.prompt)[None], (B, 1)), rng.integers(0, dims.vocab, size=(B, 6))], axis=1).astype(np.int32)
    got = eng.decode_forced(toks)
    want = orc.forced_logits(enc, toks)
    assert np.abs(got - want).max() < LOGIT_ATOL
    # the weights are quantised: the unquantised oracle on the same encoder output is measurably further away
    far = WhisperOracle(dims, small["w"]).forced_logits(enc, toks)
    assert np.abs(got - far).max() > 4 * np.abs(got - want).max()


@pytest.mark.parametrize("graph", [0, 1])
def test_fp8_greedy_ids_vs_oracle(small, graph):
    eng, orc, B, dims = small["eng"], small["orc"], small["B"], small["dims"]
    opts = synthetic_options(dims, max_new=12)
    eng.set_option("graph", graph)
    eng.set_decode_options(opts)
    eng.logmel(small["pcm"], want_output=False)
    enc = eng.encode(B)
    ids, lens, _ = eng.decode_greedy(B)
    wi, wl, _ = orc.greedy(enc, list(opts.prompt), opts.max_new, opts.eot, opts.effective_suppress(), opts.begin_suppress)
    assert_ids_match_up_to_ties(ids, wi, orc.last_margins)
    whole, wlens = eng.transcribe_batch(small["pcm"])          # the one-call path runs the same kernels
    assert (whole == ids).all() and (wlens == lens).all()
    eng.set_option("graph", 1)


def test_fp8_batch_sizes_micro():
    """Batches of 1, 16, 17 and 32 (one and two decode tiles) give the ids each chunk gets alone."""
    dims = mw.get_dims("micro")
    w = synth.make_weights_np(dims, 4321)
    eng = _fp8_engine(dims, w, 32)
    try:
        opts = synthetic_options(dims, max_new=10)
        eng.set_decode_options(opts)
        pcm = synth.make_pcm_np(32, 31)
        full, _ = eng.transcribe_batch(pcm)
        for n in (1, 16, 17):
            part, _ = eng.transcribe_batch(pcm[:n])
            assert (part == full[:n]).all(), n
        fork = eng.fork()
        try:
            again, _ = fork.transcribe_batch(pcm)
            assert (again == full).all()
        finally:
            fork.close()
    finally:
        eng.close()


def test_fp8_rejects_bad_width():
    """d_model / ffn_dim must be multiples of 128 for the fp8 products (K-tiles of 128 bytes)."""
    from manual_whisper_amd.engine import Engine, EngineError
    from manual_whisper_amd.dims import WhisperDims
    bad = WhisperDims("odd", 80, 192, 3, 1, 1, 768, 512, 1500, 448)
    with pytest.raises(EngineError, match="128"):
        Engine(bad, 0, 1, weight_format="fp8")


def test_config5_large_v3_fp8_one_chunk_vs_oracle():
    """Full-size architecture with fp8 weights: logits and ids of one chunk against the oracle on the engine's
    encoder output, the encoder output against the oracle within the activation-rounding noise, batch 32."""
    import torch
    from manual_whisper_amd.engine import Engine
    dims = mw.get_dims("large-v3")
    wt = synth.make_weights_torch(dims, 1234, torch.device("cuda", 0))
    eng = Engine(dims, 0, 32, weight_format="fp8")
    eng.load_weights(wt)
    orc = WhisperOracle(dims, {k: v.float().cpu().numpy() for k, v in wt.items()}, weight_format="fp8")
    del wt
    torch.cuda.empty_cache()
    try:
        opts = synthetic_options(dims, max_new=6, suppress_eot=True)
        eng.set_decode_options(opts)
        pcm = synth.make_pcm_np(1, seed=303)
        eng.logmel(pcm, want_output=False)
        enc = eng.encode(1)
        ids, lens, logits = eng.decode_greedy(1, want_logits=True)
        wi, wl, wlog = orc.greedy(enc, list(opts.prompt), opts.max_new, opts.eot, opts.effective_suppress(),
                                  opts.begin_suppress, return_logits=True)
        assert np.abs(logits[0] - wlog[0]).max() < 2 * LOGIT_ATOL
        assert_ids_match_up_to_ties(ids, wi, orc.last_margins)
        # encoder: the yardstick is the activation-rounding noise (see the module docstring), i.e. the fp8 oracle
        # against the same dequantised weights without activation rounding.  32 layers deep the rounding histories
        # of engine and oracle decorrelate: two independent histories would sit at sqrt(2) x the noise, identical
        # ones at 0; measured 0.9.
        mel_ref = log_mel(pcm, dims.n_mels)
        enc_ref = orc.encode(mel_ref)
        noise_free = WhisperOracle(dims, {k: v for k, v in orc.w.items()}).encode(mel_ref)    # same (dequantised) weights, no activation rounding
        assert _rms(enc_ref - noise_free) > 0.01
        assert _rms(enc - enc_ref) < 1.2 * _rms(enc_ref - noise_free), (_rms(enc - enc_ref), _rms(enc_ref - noise_free))
        assert _rms(enc - noise_free) < 1.2 * _rms(enc_ref - noise_free), (_rms(enc - noise_free), _rms(enc_ref - noise_free))
        # batch 32 (the per-GPU batch of the configuration): both halves decode as they do alone
        opts = synthetic_options(dims, max_new=12, suppress_eot=True)
        eng.set_decode_options(opts)
        pcm = synth.make_pcm_np(32, seed=505)
        a, la = eng.transcribe_batch(pcm)
        lo, _ = eng.transcribe_batch(pcm[:16])
        hi, _ = eng.transcribe_batch(pcm[16:])
        assert (la == 12).all() and (a[:16] == lo).all() and (a[16:] == hi).all()
        assert len({tuple(r) for r in a}) > 1
        # configs[4] as the benchmark runs it (32 chunks in one pass): 2 of the 32 rows x 32 sampled tokens against
        # the oracle on the engine's own encoder output (identical fp32 inputs -> identical codes)
        from test_gpu_configs import _rows_vs_oracle
        first_err, path_err, forced_err, _ = _rows_vs_oracle(eng, orc, dims, pcm, [3, 28], 32, enc_rows="engine")
        assert first_err < LOGIT_ATOL, first_err
        assert path_err < 2 * LOGIT_ATOL and forced_err < 2 * LOGIT_ATOL, (path_err, forced_err)
    finally:
        eng.close()


def test_config5_large_v3_fp8_decode_group_rows_vs_oracle():
    """configs[4]'s decode shape as the benchmark runs it (`fp8_b32`): seven encoder passes of 32 fp8 chunks written into
    one decode engine's cache (mw_encode_into), one greedy loop over the 224 sequences (gemm_dec_kernel<..., W8> in two
    blocks of 128, cross_attn_stream_kernel).  3 of the 224 rows x 16 sampled tokens against the fp8 oracle on the
    engine's own encoder output of those chunks (identical fp32 inputs -> identical codes), and the group's ids against
    the same chunks decoded 32 at a time."""
    import torch
    from manual_whisper_amd.engine import Engine
    dims = mw.get_dims("large-v3")
    wt = synth.make_weights_torch(dims, 1234, torch.device("cuda", 0))
    eng = Engine(dims, 0, 32, weight_format="fp8")
    eng.load_weights(wt)
    orc = WhisperOracle(dims, {k: v.float().cpu().numpy() for k, v in wt.items()}, weight_format="fp8")
    del wt
    torch.cuda.empty_cache()
    n_new, rows = 16, [0, 100, 223]
    opts = synthetic_options(dims, max_new=n_new, suppress_eot=True)
    eng.set_decode_options(opts)
    pcm = synth.make_pcm_np(224, seed=919)
    dec = eng.fork(224)
    try:
        dec.set_decode_options(opts)
        for i in range(7):
            eng.encode_into(pcm[32 * i: 32 * i + 32], dec, 32 * i)
        dec.set_encoded(224)
        dec.set_option("dec_gemm_fat", 1)          # the projections' form the pools select when engines share the GPU
        ids, lens, logits = dec.decode_greedy(224, want_logits=True)        # logits [n_new][224][vocab]
        assert (lens == n_new).all()
        # a chunk's encoder output does not depend on what shares its pass: the rows' own output, from a pass of their own
        eng.logmel(pcm[rows], want_output=False)
        enc = eng.encode(len(rows))
        wi, wl, wlog = orc.greedy(enc, list(opts.prompt), n_new, opts.eot, opts.effective_suppress(), opts.begin_suppress,
                                  return_logits=True)
        assert_ids_match_up_to_ties(ids[rows], wi, orc.last_margins)
        first_err = float(np.abs(logits[0][rows] - wlog[0]).max())
        path_err = 0.0
        for j, r in enumerate(rows):
            diff = np.nonzero(ids[r] != wi[j])[0]
            upto = int(diff[0]) + 1 if diff.size else n_new
            path_err = max(path_err, float(np.abs(logits[:upto, r] - wlog[:upto, j]).max()))
        same = int((ids[rows] == wi).all(axis=1).sum())
        by32 = np.concatenate([eng.transcribe_batch(pcm[32 * i: 32 * i + 32])[0] for i in range(7)])
        agree = int((by32 == ids).all(axis=1).sum())
        print(f"fp8 decode group of 224, rows {rows}: first-position logits {first_err:.2e}, on-path logits {path_err:.2e}, "
              f"{same}/{len(rows)} rows identical to the oracle over {n_new} tokens, {agree}/224 identical to passes of 32")
        assert first_err < LOGIT_ATOL and path_err < 2 * LOGIT_ATOL, (first_err, path_err)
        assert same == 3 and agree >= 215          # measured: 3/3 and 221/224
        # a sequence may part from its pass of 32 (other summation order in the projections) only where its own logits
        # hold the two choices within a near-tie
        for r in np.nonzero((by32 != ids).any(axis=1))[0]:
            n = int(np.nonzero(by32[r] != ids[r])[0][0])
            gap = float(logits[n, r, ids[r, n]] - logits[n, r, by32[r, n]])
            assert abs(gap) < TIE_GAP, (int(r), n, gap)
        # the projections' form of an engine that has the GPU to itself: the same sums split differently
        dec.set_option("dec_gemm_fat", 0)
        thin, _, thin_logits = dec.decode_greedy(224, want_logits=True)
        assert float(np.abs(thin_logits[0] - logits[0]).max()) < LOGIT_ATOL / 2
        apart = np.nonzero((thin != ids).any(axis=1))[0]
        assert len(apart) <= 9, apart
        for r in apart:
            n = int(np.nonzero(thin[r] != ids[r])[0][0])
            gap = float(logits[n, r, ids[r, n]] - logits[n, r, thin[r, n]])
            assert abs(gap) < TIE_GAP, (int(r), n, gap)
    finally:
        dec.close()
        eng.close()


# ---------------------------------------------------------------------------------------------
# e4m3 cross-attention K/V cache (kv_format="fp8")
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["micro", "tiny"])
def test_kv8_decoder_vs_oracle(name):
    """The projection's epilogue quantises each (sequence, head, frame) row of 64 keys / values; the decode and
    prefill cross-attention read the codes.  With the oracle on the engine's own encoder output the projected
    values agree to fp32 rounding, so the codes are the same and the fp16 tolerances apply."""
    from manual_whisper_amd.engine import Engine
    dims = mw.get_dims(name)
    w = synth.make_weights_np(dims, 1234)
    eng = Engine(dims, 0, 32, weight_format="fp8", kv_format="fp8")
    eng.load_weights(w)
    try:
        B = 3
        pcm = synth.make_pcm_np(B, 91)
        orc = WhisperOracle(dims, w, weight_format="fp8", kv_format="fp8")
        base = synthetic_options(dims, max_new=10)
        eng.set_decode_options(base)
        eng.logmel(pcm, want_output=False)
        enc = eng.encode(B)
        rng = np.random.default_rng(8)
        toks = np.concatenate([np.tile(np.asarray(base.prompt)[None], (B, 1)), rng.integers(0, dims.vocab, size=(B, 6))], axis=1).astype(np.int32)
        got = eng.decode_forced(toks)
        want = orc.forced_logits(enc, toks)
        assert np.abs(got - want).max() < LOGIT_ATOL
        # the cache itself (codes x row scales read back through the C ABI) against the oracle's quantised projection of
        # the engine's encoder output: the same codes except where a value sits on a rounding boundary (one e4m3 step = 1/8
        # of the value at most; a row's scale follows its largest element)
        ckv = orc.cross_kv(enc)
        for layer in range(dims.dec_layers):
            for b in range(B):
                k, v = eng.read_cross_kv(layer, b)
                for have, ref in ((k, ckv[layer][0][b]), (v, ckv[layer][1][b])):
                    d = np.abs(have - ref)
                    rowmax = np.abs(ref).max(axis=-1, keepdims=True)
                    # same code: the values then differ only by the fp32 rounding of the row scale (one e4m3 step is >= 1/16 of a value)
                    assert (d > 1e-4 * rowmax).mean() < 0.02, "more than 2 % of the cached values differ from the oracle's codes"
                    assert (d <= 0.14 * rowmax + 1e-6).all()
        far = WhisperOracle(dims, w, weight_format="fp8").forced_logits(enc, toks)          # fp16 cache: measurably further away
        assert np.abs(got - far).max() > 3 * np.abs(got - want).max()
        # greedy, stepping and prefilled, a long prompt (the grouped form of the kernel: query rows sharing a cache)
        prompt = tuple(int(t) for t in rng.integers(0, dims.vocab, size=29)) + (base.prompt[-1],)
        opts = DecodeOptions(prompt, base.eot, base.suppress, base.begin_suppress, 10)
        eng.set_decode_options(opts)
        wi, wl, wlog = orc.greedy(enc, list(prompt), 10, opts.eot, opts.effective_suppress(), opts.begin_suppress, return_logits=True)
        for prefill in (0, 1):
            eng.set_option("prefill", prefill)
            ids, lens, logits = eng.decode_greedy(B, want_logits=True)
            assert np.abs(logits[0] - wlog[0]).max() < LOGIT_ATOL
            assert_ids_match_up_to_ties(ids, wi, orc.last_margins)
        # a beam of one is greedy; five beams share each recording's cache rows
        b1 = eng.decode_beam(B, 1)
        assert (b1[0] == ids).all()
        b5 = eng.decode_beam(B, 5, 1.0, want_hypotheses=True)
        w5 = orc.beam_search(enc, list(prompt), 10, opts.eot, 5, 1.0, opts.suppress, opts.begin_suppress)
        for a in range(B):
            if orc.last_beam_margins[a] > 4e-2:
                assert {tuple(int(x) for x in b5[3][a, c, : b5[4][a, c]]) for c in range(5)} == {tuple(t) for t, _ in w5[3][a]}
        # batch position and size do not matter
        eng.set_decode_options(base)
        pcm32 = synth.make_pcm_np(32, 17)
        full, _ = eng.transcribe_batch(pcm32)
        for n in (1, 17):
            part, _ = eng.transcribe_batch(pcm32[:n])
            assert (part == full[:n]).all(), n
    finally:
        eng.close()


def test_kv8_needs_fp8_weights():
    from manual_whisper_amd.engine import Engine, EngineError
    with pytest.raises(EngineError, match="MW_KV_FP8"):
        Engine(mw.get_dims("micro"), 0, 1, weight_format="f16", kv_format="fp8")


def test_config5_large_v3_kv8_one_chunk_vs_oracle():
    import torch
    from manual_whisper_amd.engine import Engine
    dims = mw.get_dims("large-v3")
    wt = synth.make_weights_torch(dims, 1234, torch.device("cuda", 0))
    eng = Engine(dims, 0, 32, weight_format="fp8", kv_format="fp8")
    eng.load_weights(wt)
    orc = WhisperOracle(dims, {k: v.float().cpu().numpy() for k, v in wt.items()}, weight_format="fp8", kv_format="fp8")
    del wt
    torch.cuda.empty_cache()
    try:
        opts = synthetic_options(dims, max_new=6, suppress_eot=True)
        eng.set_decode_options(opts)
        pcm = synth.make_pcm_np(1, seed=303)
        eng.logmel(pcm, want_output=False)
        enc = eng.encode(1)
        ids, lens, logits = eng.decode_greedy(1, want_logits=True)
        wi, wl, wlog = orc.greedy(enc, list(opts.prompt), opts.max_new, opts.eot, opts.effective_suppress(),
                                  opts.begin_suppress, return_logits=True)
        assert np.abs(logits[0] - wlog[0]).max() < 2 * LOGIT_ATOL
        assert_ids_match_up_to_ties(ids, wi, orc.last_margins)
        opts = synthetic_options(dims, max_new=12, suppress_eot=True)
        eng.set_decode_options(opts)
        pcm = synth.make_pcm_np(32, seed=505)
        a, la = eng.transcribe_batch(pcm)
        lo, _ = eng.transcribe_batch(pcm[:16])
        hi, _ = eng.transcribe_batch(pcm[16:])
        assert (la == 12).all() and (a[:16] == lo).all() and (a[16:] == hi).all()
    finally:
        eng.close()
