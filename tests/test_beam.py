"""Beam search (the decoding whisperx runs by default; the reference overrides only initial_prompt,
transcribe.py:111): oracle self-checks on the CPU, the HIP path against the oracle on the GPU.
Hypothesis scores are cumulative log-probabilities; fp16 weights move them by a few 1e-3, so a GPU
winner may differ from the oracle's only where the oracle's own ranking is that close."""
import numpy as np
import pytest

import manual_whisper_amd as mw
from manual_whisper_amd import synth
from manual_whisper_amd.decode_options import DecodeOptions, synthetic_options
from oracle.whisper_ref import WhisperOracle, log_mel

SCORE_ATOL = 2e-2       # cumulative log-probability over <= 16 tokens


@pytest.fixture(scope="module")
def micro_case():
    dims = mw.get_dims("micro")
    w = synth.make_weights_np(dims, 1234)
    orc = WhisperOracle(dims, w)
    pcm = synth.make_pcm_np(3, seed=1)
    mel = log_mel(pcm, dims.n_mels)
    return dims, w, orc, pcm, mel, orc.encode(mel)


def test_oracle_beam_of_one_is_greedy(micro_case):
    dims, _, orc, _, _, enc = micro_case
    o = synthetic_options(dims, max_new=8)
    gi, gl, _ = orc.greedy(enc, list(o.prompt), 8, o.eot, o.suppress, o.begin_suppress)
    bi, bl, _, hyps = orc.beam_search(enc, list(o.prompt), 8, o.eot, 1, 1.0, o.suppress, o.begin_suppress)
    assert (gi == bi).all() and (gl == bl).all() and all(len(h) == 1 for h in hyps)


def test_oracle_beam_scores_are_consistent(micro_case):
    """Every hypothesis' score equals the teacher-forced log-probability of its tokens, the winner's
    per-token score is the best of its list, and a wider beam never finds a worse winner here."""
    dims, _, orc, _, _, enc = micro_case
    o = synthetic_options(dims, max_new=6)
    sup = np.zeros(dims.vocab, bool)
    sup[list(o.suppress)] = True
    bsup = sup.copy()
    bsup[list(o.begin_suppress)] = True
    ids, lens, avg, hyps = orc.beam_search(enc, list(o.prompt), 6, o.eot, 4, 1.0, o.suppress, o.begin_suppress)
    for a in range(enc.shape[0]):
        for toks, lp in hyps[a]:
            seq = np.asarray(list(o.prompt) + toks, np.int32)[None]
            lg = orc.forced_logits(enc[a:a + 1], seq)[0].astype(np.float64)
            total = 0.0
            for n, t in enumerate(toks):
                row = np.where(bsup if n == 0 else sup, -np.inf, lg[len(o.prompt) - 1 + n])
                total += row[t] - (row.max() + np.log(np.exp(row - row.max()).sum()))
            assert abs(total - lp) < 1e-3
        best = max(lp / max(len(t), 1) for t, lp in hyps[a])
        assert abs(avg[a] * (lens[a] + 1) / max(lens[a], 1) - best) < 1e-5
    _, l1, a1, _ = orc.beam_search(enc, list(o.prompt), 6, o.eot, 1, 1.0, o.suppress, o.begin_suppress)
    assert (avg * (lens + 1) / np.maximum(lens, 1) >= a1 * (l1 + 1) / np.maximum(l1, 1) - 1e-6).all()


def _forced_score(orc, enc_a, opts, toks, ends_with_eot):
    """Log-probability of `toks` (+ end-of-text) for one recording under the oracle, teacher-forced."""
    sup = np.zeros(orc.d.vocab, bool)
    sup[list(opts.effective_suppress())] = True
    bsup = sup.copy()
    bsup[list(opts.begin_suppress)] = True
    full = list(toks) + ([opts.eot] if ends_with_eot else [])
    seq = np.asarray(list(opts.prompt) + full, np.int32)[None]
    lg = orc.forced_logits(enc_a[None], seq)[0].astype(np.float64)
    total = 0.0
    for n, t in enumerate(full):
        row = np.where(bsup if n == 0 else sup, -np.inf, lg[len(opts.prompt) - 1 + n])
        total += row[t] - (row.max() + np.log(np.exp(row - row.max()).sum()))
    return total


def _compare(got, want, margins, orc, enc, opts, max_new):
    """got: (ids, lens, avg, hyp_ids, hyp_len, hyp_lp) from the engine; want: (ids, lens, avg, hyps) from the oracle.

    A beam search prunes on score ORDER, so where the oracle's own run had a proposal within rounding
    distance of the beam's edge (margins[a] small) the two searches may legitimately walk apart.  There
    only the invariant is checked: every engine hypothesis carries the score the oracle gives the same
    tokens.  Everywhere else the hypothesis lists and the winner must be the oracle's."""
    ids, lens, avg, hi, hl, hs = got
    wi, wl, wavg, hyps = want
    for a in range(ids.shape[0]):
        oracle = {tuple(t): lp for t, lp in hyps[a]}
        mine = {tuple(int(x) for x in hi[a, c, : hl[a, c]]): float(hs[a, c]) for c in range(hi.shape[1]) if hl[a, c] >= 0}
        assert (ids[a, lens[a]:] == opts.eot).all()
        assert tuple(int(x) for x in ids[a, : lens[a]]) in mine
        for k, v in mine.items():
            if k in oracle:
                assert abs(v - oracle[k]) < SCORE_ATOL
            else:
                assert abs(v - _forced_score(orc, enc[a], opts, k, len(k) < max_new)) < SCORE_ATOL
        if margins[a] > 2 * SCORE_ATOL:
            assert set(mine) == set(oracle), f"recording {a}: hypothesis lists differ although the oracle margin is {margins[a]:.3f}"
            if (ids[a] != wi[a]).any():      # only a photo finish between two hypotheses may change the winner
                assert abs(float(avg[a]) - float(wavg[a])) < SCORE_ATOL / 4
            else:
                assert lens[a] == wl[a]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["micro", "tiny"])
def test_beam_search_matches_oracle(name):
    from manual_whisper_amd.engine import Engine, EngineError
    dims = mw.get_dims(name)
    w = synth.make_weights_np(dims, 1234)
    orc = WhisperOracle(dims, w)
    pcm = synth.make_pcm_np(3, seed=1)
    mel = log_mel(pcm, dims.n_mels)
    enc = orc.encode(mel)
    eng = Engine(dims, 0, 32)
    try:
        eng.load_weights(w)
        o = synthetic_options(dims, max_new=10)
        eng.set_decode_options(o)
        eng.encode(3, mel=mel, want_output=False)
        g_ids, g_lens = eng.decode_greedy(3)[:2]
        b_ids, b_lens, _ = eng.decode_beam(3, 1)
        assert (b_ids == g_ids).all() and (b_lens == g_lens).all()                 # a beam of one is greedy
        for graph in (0, 1):
            eng.set_option("graph", graph)
            got = eng.decode_beam(3, 5, 1.0, want_hypotheses=True)
            want = orc.beam_search(enc, list(o.prompt), 10, o.eot, 5, 1.0, o.suppress, o.begin_suppress)
            _compare(got, want, orc.last_beam_margins, orc, enc, o, 10)
        # end-of-text inside the search: declare a token that greedy decoding emits early to be EOT,
        # so hypotheses finish at different lengths and the finished lists, patience and early stop all run
        eot2 = int(g_ids[0, 3])
        o2 = DecodeOptions(o.prompt, eot2, tuple(t for t in o.suppress if t != eot2), (), 12)
        eng.set_decode_options(o2)
        for beam, patience in ((4, 1.0), (3, 2.0), (8, 1.0)):
            got = eng.decode_beam(3, beam, patience, want_hypotheses=True)
            want = orc.beam_search(enc, list(o2.prompt), 12, eot2, beam, patience, o2.suppress, o2.begin_suppress)
            _compare(got, want, orc.last_beam_margins, orc, enc, o2, 12)
            assert any(len(t) < 12 for h in want[3] for t, _ in h), "the crafted end-of-text never fired: the test lost its point"
        # early stop: one recording whose finished list fills long before max_new ends the loop
        eng.set_option("check_every", 1)
        for a in range(3):
            w1 = orc.beam_search(enc[a:a + 1], list(o2.prompt), 12, eot2, 3, 1.0, o2.suppress, o2.begin_suppress)
            eng.encode(1, mel=mel[a:a + 1], want_output=False)
            _compare(eng.decode_beam(1, 3, 1.0, want_hypotheses=True), w1, orc.last_beam_margins, orc, enc[a:a + 1], o2, 12)
            longest = max(len(t) for t, _ in w1[3][0])
            if longest < 10:
                assert eng.stage_ms()["decode_steps"] <= len(o2.prompt) + longest + 2     # a check is read one check later
        eng.set_option("check_every", 4)
        # the same hypotheses whatever else shares the batch
        eng.encode(1, mel=mel[1:2], want_output=False)
        one = eng.decode_beam(1, 4, 1.0)
        eng.encode(3, mel=mel, want_output=False)
        three = eng.decode_beam(3, 4, 1.0)
        assert (one[0][0] == three[0][1]).all() and one[1][0] == three[1][1]
        for bad in (dict(n_audio=3, beam=9), dict(n_audio=3, beam=0), dict(n_audio=3, beam=5, patience=0.5), dict(n_audio=2, beam=4)):
            with pytest.raises(EngineError):
                eng.decode_beam(**bad)                                              # beam range, patience, n_audio != last encode
        eng.encode(3, mel=mel, want_output=False)
        eng.set_option("graph", 1)
    finally:
        eng.close()


@pytest.mark.gpu
def test_model_transcribe_with_beam_size():
    """asr_options={"beam_size": 5}: Model.transcribe groups max_batch // 5 chunks per pass and returns
    what the engine's beam search returns for each chunk."""
    model = mw.load_model("synthetic:micro", "cuda", asr_options={"beam_size": 5, "max_new_tokens": 8}, max_batch=12, n_engines=2)
    try:
        audio = synth.make_pcm_np(5, seed=44).reshape(-1)[: 16000 * 140]           # 140 s -> 5 chunks -> passes of 2, 2, 1
        res = model.transcribe(audio, batch_size=16, return_tokens=True)
        assert len(res["segments"]) == 5
        chunks, _ = model.windows(audio)
        eng = model.engine
        for a in range(5):
            eng.logmel(chunks[a:a + 1], want_output=False)
            eng.encode(1, want_output=False)
            ids, lens, _ = eng.decode_beam(1, 5)
            assert res["segments"][a]["tokens"] == [int(t) for t in ids[0, : lens[0]]]
    finally:
        model.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,n_prompt", [("micro", 41), ("tiny", 70)])
def test_beam_search_with_a_long_prompt_prefilled(name, n_prompt):
    """whisperx's default decoding with the reference's initial_prompt: a beam of 5 behind a prompt of several
    dozen ids.  The prompt goes through the decoder once per recording (its k/v rows shared by the beams through
    the cache-row table); results must be the oracle's, and those of stepping through the prompt."""
    from manual_whisper_amd.engine import Engine
    dims = mw.get_dims(name)
    w = synth.make_weights_np(dims, 1234)
    orc = WhisperOracle(dims, w)
    pcm = synth.make_pcm_np(3, seed=2)
    mel = log_mel(pcm, dims.n_mels)
    enc = orc.encode(mel)
    eng = Engine(dims, 0, 16)
    try:
        eng.load_weights(w)
        base = synthetic_options(dims, max_new=8)
        rng = np.random.default_rng(n_prompt)
        prompt = tuple(int(t) for t in rng.integers(0, dims.vocab, size=n_prompt - 1)) + (base.prompt[-1],)
        o = DecodeOptions(prompt, base.eot, base.suppress, base.begin_suppress, 8)
        eng.set_decode_options(o)
        eng.encode(3, mel=mel, want_output=False)
        want = orc.beam_search(enc, list(prompt), 8, o.eot, 5, 1.0, o.suppress, o.begin_suppress)
        runs = {}
        for prefill in (0, 1):
            eng.set_option("prefill", prefill)
            got = eng.decode_beam(3, 5, 1.0, want_hypotheses=True)
            _compare(got, want, orc.last_beam_margins, orc, enc, o, 8)
            runs[prefill] = got
        for a in range(3):
            if orc.last_beam_margins[a] > 2 * SCORE_ATOL:
                assert (runs[0][0][a] == runs[1][0][a]).all() or abs(float(runs[0][2][a]) - float(runs[1][2][a])) < SCORE_ATOL / 4
        assert eng.stage_ms()["decode_steps"] == n_prompt + 8 - 1          # positions processed, prefilled ones included
    finally:
        eng.close()


@pytest.mark.gpu
def test_fork_with_a_cache_for_the_recordings_only():
    """mw_fork_cached: an engine for n recordings x b hypotheses holds n cached recordings.  Beam search and best-of-n on
    it give what an engine with a slot per sequence gives; what needs a slot per sequence is refused beyond n."""
    from manual_whisper_amd.engine import Engine, EngineError
    dims = mw.get_dims("micro")
    w = synth.make_weights_np(dims, 1234)
    pcm = synth.make_pcm_np(4, seed=5)
    full = Engine(dims, 0, 20)
    try:
        full.load_weights(w)
        o = synthetic_options(dims, max_new=10)
        full.set_decode_options(o)
        lean = full.fork(20, cache_sequences=4)
        assert lean.max_batch == 20 and lean.cache_sequences == 4
        lean.set_decode_options(o)
        got = []
        for e in (full, lean):
            e.logmel(pcm, want_output=False)
            e.encode(4, want_output=False)
            got.append(e.decode_beam(4, 5, 1.0, want_hypotheses=True))
        for a, b in zip(got[0], got[1]):
            assert np.array_equal(np.asarray(a), np.asarray(b))
        ids_g, lens_g = lean.decode_greedy(4)[:2]
        want_g, want_l = full.decode_greedy(4)[:2]
        assert (ids_g == want_g).all() and (lens_g == want_l).all()
        five = synth.make_pcm_np(5, seed=6)
        lean.logmel(five, want_output=False)                      # log-mel has no cache to fill: 5 <= max_batch is fine
        with pytest.raises(EngineError):
            lean.encode(5, want_output=False)
        with pytest.raises(EngineError):
            lean.decode_beam(5, 4)
        with pytest.raises(EngineError):
            full.fork(8, cache_sequences=9)
        lean.close()
    finally:
        full.close()
