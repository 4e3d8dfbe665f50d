"""Parity cases at the sizes of the target configurations (beyond the micro/tiny fixtures).

config 2: whisper-small fp16, batch 8          -> logits and ids against the oracle
config 3: whisper-large-v3 fp16, batch 16      -> one chunk against the oracle at full size, and
                                                  size-independent properties at batch 16
config 4: long audio sharded over ranks        -> 120-chunk bookkeeping (CPU side is in test_host.py)
Weights are generated on the GPU with the same hash the numpy generator uses.
"""
import numpy as np
import pytest

import manual_whisper_amd as mw
from manual_whisper_amd import synth
from manual_whisper_amd.decode_options import synthetic_options
from oracle.whisper_ref import WhisperOracle, log_mel

from test_gpu_parity import ENC_ATOL, LOGIT_ATOL, TIE_GAP, assert_ids_match_up_to_ties

pytestmark = pytest.mark.gpu


def _engine_and_oracle(name, max_batch, with_oracle=True):
    import torch
    from manual_whisper_amd.engine import Engine
    dims = mw.get_dims(name)
    wt = synth.make_weights_torch(dims, 1234, torch.device("cuda", 0))
    eng = Engine(dims, 0, max_batch)
    eng.load_weights(wt)
    orc = WhisperOracle(dims, {k: v.float().cpu().numpy() for k, v in wt.items()}) if with_oracle else None
    del wt
    torch.cuda.empty_cache()
    return dims, eng, orc


def test_config2_small_batch8():
    dims, eng, orc = _engine_and_oracle("small", 8)
    try:
        opts = synthetic_options(dims, max_new=6)
        eng.set_decode_options(opts)
        pcm = synth.make_pcm_np(8, seed=202)
        mel_ref = log_mel(pcm, dims.n_mels)
        enc_ref = orc.encode(mel_ref)
        enc = eng.encode(8, mel=mel_ref)
        assert np.abs(enc - enc_ref).max() < ENC_ATOL
        toks = np.tile(np.asarray(opts.prompt + (11, 22, 33), np.int32)[None], (8, 1))
        got = eng.decode_forced(toks)
        want = orc.forced_logits(enc_ref, toks)
        assert np.abs(got - want).max() < LOGIT_ATOL
        ids, lens = eng.transcribe_batch(pcm)
        wi, wl = orc.greedy(enc_ref, list(opts.prompt), opts.max_new, opts.eot, opts.suppress, opts.begin_suppress)[:2]
        assert_ids_match_up_to_ties(ids, wi, orc.last_margins)
        assert (ids == wi).all(axis=1).sum() >= 6
    finally:
        eng.close()


@pytest.fixture(scope="module")
def large():
    dims, eng, orc = _engine_and_oracle("large-v3", 16)
    yield dims, eng, orc
    eng.close()


def test_config3_large_v3_one_chunk_vs_oracle(large):
    """Full-size architecture, one 30 s chunk, 6 sampled tokens: encoder output, logits and ids."""
    dims, eng, orc = large
    opts = synthetic_options(dims, max_new=6, suppress_eot=True)
    eng.set_decode_options(opts)
    pcm = synth.make_pcm_np(1, seed=303)
    mel_ref = log_mel(pcm, dims.n_mels)
    mel = eng.logmel(pcm)
    assert np.abs(mel - mel_ref).max() < 1e-4
    enc_ref = orc.encode(mel_ref)
    enc = eng.encode(1)
    assert np.abs(enc - enc_ref).max() < 2 * ENC_ATOL              # 32 layers deep
    ids, lens, logits = eng.decode_greedy(1, want_logits=True)
    wi, wl, wlog = orc.greedy(enc_ref, list(opts.prompt), opts.max_new, opts.eot, opts.effective_suppress(),
                              opts.begin_suppress, return_logits=True)
    assert np.abs(logits[0] - wlog[0]).max() < 2 * LOGIT_ATOL
    assert_ids_match_up_to_ties(ids, wi, orc.last_margins)
    # the cached cross-attention K/V of the first, a middle and the last decoder layer, read back through the C ABI
    want = orc.cross_kv(enc_ref)
    near = orc.cross_kv(enc)
    for layer in (0, 15, 31):
        k, v = eng.read_cross_kv(layer, 0)
        for got, ref, ref_near in ((k, want[layer][0][0], near[layer][0][0]), (v, want[layer][1][0], near[layer][1][0])):
            assert np.abs(got - ref).max() < 2 * ENC_ATOL, (layer, float(np.abs(got - ref).max()))
            assert np.abs(got - ref_near).max() < 2.5e-3 + 1e-3 * np.abs(ref_near).max()     # fp16 rounding of the oracle's value


def _rows_vs_oracle(eng, orc, dims, pcm, rows, n_new, enc_rows=None, enc_atol=None):
    """The benchmark's own workload shape: the whole batch through the engine, `rows` of it through the oracle,
    `n_new` sampled tokens.  Checks (rows only): encoder output, the logits of every sampled position while the
    engine is on the oracle's path, ids identical up to oracle near-ties, and teacher-forced logits along the
    oracle's ids at every position whatever the greedy paths did.  Returns the measured maxima."""
    B = pcm.shape[0]
    opts = synthetic_options(dims, max_new=n_new, suppress_eot=True)
    eng.set_decode_options(opts)
    eng.logmel(pcm, want_output=False)
    enc = eng.encode(B)
    ids, lens, logits = eng.decode_greedy(B, want_logits=True)            # logits [n_new][B][vocab]
    assert (lens == n_new).all()
    if enc_rows is None:                                                    # fp16: the oracle runs its own encoder
        enc_ref = orc.encode(log_mel(pcm[rows], dims.n_mels))
        enc_err = float(np.abs(enc[rows] - enc_ref).max())
        assert enc_err < enc_atol, enc_err
    else:                                                                   # fp8: identical fp32 inputs -> identical codes
        enc_ref, enc_err = enc[rows], 0.0
    wi, wl, wlog = orc.greedy(enc_ref, list(opts.prompt), n_new, opts.eot, opts.effective_suppress(), opts.begin_suppress,
                              return_logits=True)
    assert_ids_match_up_to_ties(ids[rows], wi, orc.last_margins)
    first_err = float(np.abs(logits[0][rows] - wlog[0]).max())
    path_err, compared = 0.0, 0
    for j, r in enumerate(rows):
        diff = np.nonzero(ids[r] != wi[j])[0]
        upto = int(diff[0]) + 1 if diff.size else n_new                    # position upto-1 was still fed the same history
        path_err = max(path_err, float(np.abs(logits[:upto, r] - wlog[:upto, j]).max()))
        compared += upto
    # teacher-forced along the oracle's ids: every row of the batch needs tokens, the other rows repeat row 0's
    P = len(opts.prompt)
    toks = np.tile(np.concatenate([np.asarray(opts.prompt, np.int32), wi[0][: n_new - 1]])[None], (B, 1)).astype(np.int32)
    for j, r in enumerate(rows):
        toks[r, P:] = wi[j][: n_new - 1]
    got = eng.decode_forced(toks)[rows][:, P - 1:]                          # logits that chose each sampled token
    forced_err = float(np.abs(got - np.transpose(wlog, (1, 0, 2))).max())
    same = int((ids[rows] == wi).all(axis=1).sum())
    print(f"rows {rows}: encoder {enc_err:.2e}, first-position logits {first_err:.2e}, on-path logits {path_err:.2e} "
          f"({compared} positions), teacher-forced logits {forced_err:.2e}, {same}/{len(rows)} rows identical over {n_new} tokens")
    return first_err, path_err, forced_err, same


def test_config3_large_v3_batch16_rows_vs_oracle(large):
    """configs[2] as the benchmark runs it (large-v3 fp16, 16 chunks in one pass): 4 of the 16 rows x 32 sampled
    tokens against the fp32 oracle.  Measured on MI355X: see the printed line; the bounds are 1x LOGIT_ATOL at the
    first position and 2x along the path (32 layers deep, 35 positions of accumulated cache rounding)."""
    dims, eng, orc = large
    pcm = synth.make_pcm_np(16, seed=808)
    first_err, path_err, forced_err, same = _rows_vs_oracle(eng, orc, dims, pcm, [0, 5, 10, 15], 32, enc_atol=2 * ENC_ATOL)
    assert first_err < LOGIT_ATOL, first_err
    assert path_err < 2 * LOGIT_ATOL and forced_err < 2 * LOGIT_ATOL, (path_err, forced_err)
    assert same == 4           # measured: 4/4 (a row may only part at an oracle near-tie, which these rows do not have)


@pytest.mark.parametrize("fat", [1, 0])
def test_config3_large_v3_decode_group_rows_vs_oracle(large, fat):
    """The benchmark's decode shape: seven encoder passes of 16 chunks written into one decode engine's cache
    (mw_encode_into), one greedy loop over the 112 sequences (the projections then run as gemm_dec_kernel, the
    LayerNorms as launches of their own, the cross-attention as a launch of seven rounds).  4 of the 112 rows x 16
    sampled tokens against the fp32 oracle, and the group's ids against the same chunks decoded 16 at a time.
    fat = 1: the projections' form that the pools select when engines share the GPU, as the benchmark's do (option
    "dec_gemm_fat": 128 weight rows per workgroup and, at 112 sequences, K ranges of 640 columns)."""
    dims, eng, orc = large
    n_new = 16
    opts = synthetic_options(dims, max_new=n_new, suppress_eot=True)
    eng.set_decode_options(opts)
    pcm = synth.make_pcm_np(112, seed=909)
    rows = [0, 37, 75, 111]
    dec = eng.fork(112)
    try:
        dec.set_decode_options(opts)
        dec.set_option("dec_gemm_fat", fat)
        for i in range(7):
            eng.encode_into(pcm[16 * i: 16 * i + 16], dec, 16 * i)
        dec.set_encoded(112)
        ids, lens, logits = dec.decode_greedy(112, want_logits=True)        # logits [n_new][112][vocab]
        assert (lens == n_new).all()
        enc_ref = orc.encode(log_mel(pcm[rows], dims.n_mels))
        wi, wl, wlog = orc.greedy(enc_ref, list(opts.prompt), n_new, opts.eot, opts.effective_suppress(), opts.begin_suppress,
                                  return_logits=True)
        assert_ids_match_up_to_ties(ids[rows], wi, orc.last_margins)
        first_err = float(np.abs(logits[0][rows] - wlog[0]).max())
        path_err = 0.0
        for j, r in enumerate(rows):
            diff = np.nonzero(ids[r] != wi[j])[0]
            upto = int(diff[0]) + 1 if diff.size else n_new
            path_err = max(path_err, float(np.abs(logits[:upto, r] - wlog[:upto, j]).max()))
        same = int((ids[rows] == wi).all(axis=1).sum())
        # the same chunks 16 at a time (fused-LayerNorm GEMV kernels: other summation order, so near-ties may part)
        by16 = np.concatenate([eng.transcribe_batch(pcm[16 * i: 16 * i + 16])[0] for i in range(7)])
        agree = int((by16 == ids).all(axis=1).sum())
        print(f"decode group of 112, rows {rows}: first-position logits {first_err:.2e}, on-path logits {path_err:.2e}, "
              f"{same}/4 rows identical to the oracle over {n_new} tokens, {agree}/112 identical to passes of 16")
        assert first_err < LOGIT_ATOL and path_err < 2 * LOGIT_ATOL, (first_err, path_err)
        # measured: 4/4 rows on the oracle's path and 111/112 identical to passes of 16 (the 112th parts at a near-tie)
        assert same == 4 and agree >= 110
        for r in np.nonzero((by16 != ids).any(axis=1))[0]:      # parting from the pass of 16 only at a near-tie of its own logits
            n = int(np.nonzero(by16[r] != ids[r])[0][0])
            gap = float(logits[n, r, ids[r, n]] - logits[n, r, by16[r, n]])
            assert abs(gap) < TIE_GAP, (int(r), n, gap)
    finally:
        dec.close()


def test_config3_large_v3_batch16_properties(large):
    """Batch 16 at full size: run-to-run identical, independent of batch position and batch size,
    identical from a forked engine."""
    dims, eng, _ = large
    opts = synthetic_options(dims, max_new=24, suppress_eot=True)
    eng.set_decode_options(opts)
    pcm = synth.make_pcm_np(16, seed=404)
    a, la = eng.transcribe_batch(pcm)
    b, _ = eng.transcribe_batch(pcm)
    assert (a == b).all() and (la == 24).all()
    rev, _ = eng.transcribe_batch(np.ascontiguousarray(pcm[::-1]))
    assert (rev[::-1] == a).all()
    one, _ = eng.transcribe_batch(pcm[7:8])
    assert (one[0] == a[7]).all()
    fork = eng.fork()
    try:
        c, _ = fork.transcribe_batch(pcm)
        assert (c == a).all()
    finally:
        fork.close()
    assert len({tuple(r) for r in a}) > 1, "all 16 chunks decoded to the same ids"


def test_large_v3_beam_search_properties(large):
    """Beam search at full size (3 recordings x 5 hypotheses = 15 sequences): a beam of one is greedy,
    results repeat run to run and do not depend on which other recordings share the pass, every
    hypothesis list is ranked and the winner maximises the per-token score."""
    dims, eng, _ = large
    opts = synthetic_options(dims, max_new=10, suppress_eot=True)
    eng.set_decode_options(opts)
    pcm = synth.make_pcm_np(3, seed=606)
    eng.logmel(pcm, want_output=False)
    eng.encode(3, want_output=False)
    g_ids, g_lens = eng.decode_greedy(3)[:2]
    b1 = eng.decode_beam(3, 1)
    assert (b1[0] == g_ids).all() and (b1[1] == g_lens).all()
    ids, lens, avg, hi, hl, hs = eng.decode_beam(3, 5, 1.0, want_hypotheses=True)
    again = eng.decode_beam(3, 5, 1.0)
    assert (again[0] == ids).all() and (again[2] == avg).all()
    assert (hl == 10).all() and (lens == 10).all()                      # EOT suppressed: every hypothesis runs to max_new
    for a in range(3):
        assert len({tuple(hi[a, c]) for c in range(5)}) == 5           # five distinct hypotheses
        best = int(np.argmax(hs[a] / hl[a]))
        assert (ids[a] == hi[a, best]).all() and abs(avg[a] - hs[a, best] / 11) < 1e-5
        assert hs[a].max() >= -1e-3 + np.float32(avg[a]) * 11
    eng.logmel(pcm[1:2], want_output=False)
    eng.encode(1, want_output=False)
    one = eng.decode_beam(1, 5, 1.0)
    assert (one[0][0] == ids[1]).all() and abs(one[2][0] - avg[1]) < 1e-6


def test_config4_long_audio_chunk_bookkeeping():
    """One hour = 120 chunks: spans, batching and rank shards tile the recording exactly."""
    from manual_whisper_amd import parallel, scheduler
    n = 16000 * 3600
    chunks, spans = scheduler.split_chunks(np.zeros(n, np.float32))
    assert chunks.shape == (120, 480000) and spans[0] == (0.0, 30.0) and spans[-1] == (3570.0, 3600.0)
    shards = [parallel.shard_range(120, r, 8) for r in range(8)]
    assert [b - a for a, b in shards] == [15] * 8
    assert scheduler.batches_of(15, 16) == [(0, 15)]


def test_config5_shape_large_v3_batch32():
    """Batch 32 at full size (the per-GPU batch of the fp8 configuration, here with fp16 weights):
    both halves decode exactly as they do alone at batch 16."""
    dims, eng, _ = _engine_and_oracle("large-v3", 32, with_oracle=False)
    try:
        opts = synthetic_options(dims, max_new=12, suppress_eot=True)
        eng.set_decode_options(opts)
        pcm = synth.make_pcm_np(32, seed=505)
        a, la = eng.transcribe_batch(pcm)
        lo, _ = eng.transcribe_batch(pcm[:16])
        hi, _ = eng.transcribe_batch(pcm[16:])
        assert (la == 12).all() and (a[:16] == lo).all() and (a[16:] == hi).all()
    finally:
        eng.close()


def test_large_v3_batch48_runs_as_groups():
    """48 chunks in one pass at full size (three groups of 16 inside every decode projection launch, LayerNorm as
    its own launch, two weight-row tiles per workgroup for the wide outputs): every chunk decodes as it does in a
    batch of 16.  The two routes round differently (fused against separate LayerNorm), so a sequence may part at a
    near-tie; the first tokens and nearly all sequences must agree."""
    dims, eng, _ = _engine_and_oracle("large-v3", 48, with_oracle=False)
    try:
        opts = synthetic_options(dims, max_new=12, suppress_eot=True)
        eng.set_decode_options(opts)
        pcm = synth.make_pcm_np(48, seed=707)
        a, la = eng.transcribe_batch(pcm)
        parts = np.concatenate([eng.transcribe_batch(pcm[i:i + 16])[0] for i in (0, 16, 32)])
        assert (la == 12).all()
        assert (a[:, :3] == parts[:, :3]).all()
        assert (a == parts).all(axis=1).sum() >= 45
        b, _ = eng.transcribe_batch(pcm)
        assert (a == b).all()                                   # run-to-run identical
        ragged, _ = eng.transcribe_batch(pcm[:37])
        assert (ragged == a[:37]).all()                         # group count and a ragged last group do not matter
    finally:
        eng.close()
