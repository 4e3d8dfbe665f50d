"""Temperature fallback (whisperx's asr_options temperatures / best_of / compression_ratio_threshold /
log_prob_threshold; the reference passes asr_options at transcribe.py:107-113): the counter-based draw and the
host loop on the CPU, the sampling launches against the oracle on the GPU.  The upstream decoding code is not in
the reference tree: parity with the reference is unpinned; the oracle restates the published algorithm."""
import numpy as np
import pytest

import manual_whisper_amd as mw
from manual_whisper_amd import fallback as fb
from manual_whisper_amd import synth
from manual_whisper_amd.decode_options import DecodeOptions, synthetic_options
from oracle.whisper_ref import WhisperOracle, gumbel_noise, log_mel, sample_from_logits, sample_uniform

TIE = 1e-4              # perturbed-score gaps below this may resolve either way (float32 log, fp16 weights aside)
MODEL_TIE = 4e-2        # GPU logits differ from the oracle's by a few 1e-3; a draw is compared only above this gap / temperature
LP_ATOL = 3e-2          # summed log-probability over <= 12 tokens


# ------------------------------------------------------------------------------------------------ CPU
def test_uniform_of_the_library_is_the_oracles():
    from manual_whisper_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    for _ in range(3000):
        seed, seq, pos, tok = (int(x) for x in rng.integers(0, 2 ** 32, 4))
        seq, pos, tok = seq % 4096, pos % 448, tok % 60000
        got = lib.mw_sample_uniform(seed, seq, pos, tok)
        want = float(sample_uniform(seed, np.array([seq]), pos, np.array([tok]))[0])
        assert got == want and 0.0 < got < 1.0
    # the extremes of the 23-bit value stay inside the open interval in float32
    assert np.float32((0 + 0.5) / 8388608.0) > 0 and np.float32((8388607 + 0.5) / 8388608.0) < 1


def test_gumbel_argmax_draws_from_the_tempered_softmax():
    """Frequencies of the draw over many seeds follow softmax(logits / T) (the categorical draw of the published code)."""
    rng = np.random.default_rng(3)
    V, T, n = 12, 0.7, 6000
    logits = rng.normal(0, 1.5, V).astype(np.float32)
    masked = np.zeros(V, bool)
    masked[[2, 7]] = True
    counts = np.zeros(V)
    for seed in range(n):
        tok, lp, _ = sample_from_logits(logits[None], masked, T, seed, 5)
        counts[tok[0]] += 1
    p = np.where(masked, 0.0, np.exp(logits.astype(np.float64) / T))
    p /= p.sum()
    assert counts[masked].sum() == 0
    assert np.abs(counts / n - p).max() < 4 * np.sqrt(p.max() * (1 - p.max()) / n) + 2e-3
    # the reported log-probability is that of the UNtempered distribution
    q = np.where(masked, -np.inf, logits.astype(np.float64))
    q = q - (q.max() + np.log(np.exp(q - q.max()).sum()))
    tok, lp, _ = sample_from_logits(logits[None], masked, T, 11, 0)
    assert abs(lp[0] - q[tok[0]]) < 1e-9
    tok0, _, _ = sample_from_logits(logits[None], masked, 0.0, 11, 0)
    assert tok0[0] == int(np.argmax(q))


@pytest.fixture(scope="module")
def micro_case():
    dims = mw.get_dims("micro")
    w = synth.make_weights_np(dims, 1234)
    orc = WhisperOracle(dims, w)
    pcm = synth.make_pcm_np(3, seed=1)
    mel = log_mel(pcm, dims.n_mels)
    return dims, w, orc, pcm, mel, orc.encode(mel)


def test_oracle_sampling_at_temperature_zero_is_greedy(micro_case):
    dims, _, orc, _, _, enc = micro_case
    o = synthetic_options(dims, max_new=8)
    gi, gl, _ = orc.greedy(enc, list(o.prompt), 8, o.eot, o.suppress, o.begin_suppress)
    si, sl, avg, (smp, _, slp) = orc.sample(enc, list(o.prompt), 8, o.eot, 0.0, 1, 5, o.suppress, o.begin_suppress)
    assert (gi == si).all() and (gl == sl).all() and (smp[:, 0] == si).all()
    _, _, bavg, _ = orc.beam_search(enc, list(o.prompt), 8, o.eot, 1, 1.0, o.suppress, o.begin_suppress)
    assert np.abs(avg - bavg).max() < 1e-5                  # a beam of one scores the same tokens the same way
    # inactive recordings come back empty, the others unchanged
    ai, al, _, _ = orc.sample(enc, list(o.prompt), 8, o.eot, 0.0, 1, 5, o.suppress, o.begin_suppress, active=[True, False, True])
    assert al[1] == 0 and (ai[1] == o.eot).all() and (ai[[0, 2]] == si[[0, 2]]).all()


def test_oracle_samples_differ_and_the_best_is_chosen(micro_case):
    dims, _, orc, _, _, enc = micro_case
    o = synthetic_options(dims, max_new=6)
    ids, lens, avg, (smp, sl, slp) = orc.sample(enc, list(o.prompt), 6, o.eot, 1.0, 4, 9, o.suppress, o.begin_suppress)
    assert len({tuple(r) for r in smp[0]}) > 1              # best_of draws are independent
    for a in range(3):
        j = int(np.argmax(slp[a] / np.maximum(sl[a], 1)))
        assert (ids[a] == smp[a, j]).all() and abs(avg[a] - slp[a, j] / (sl[a, j] + 1)) < 1e-6
    again = orc.sample(enc, list(o.prompt), 6, o.eot, 1.0, 4, 9, o.suppress, o.begin_suppress)
    assert (again[0] == ids).all()                           # same seed, same ids
    other = orc.sample(enc, list(o.prompt), 6, o.eot, 1.0, 4, 10, o.suppress, o.begin_suppress)
    assert (other[3][0] != smp).any()


def test_compression_ratio():
    assert fb.compression_ratio("abc" * 200) > 10
    assert fb.compression_ratio("The quick brown fox jumps over the lazy dog.") < 1.2
    assert fb.compression_ratio("") == 0.0
    assert abs(fb.compression_ratio("今天天气很好" * 30) - len(("今天天气很好" * 30).encode()) / len(__import__("zlib").compress(("今天天气很好" * 30).encode()))) < 1e-12


def _fake_decoder(table, calls):
    """decode_at over a table {temperature: [(ids, avg_logprob) per window]}; records (temperature, active)."""
    def decode_at(t, active):
        calls.append((t, active.copy()))
        rows = table[t]
        return [np.asarray(r[0], np.int32) for r in rows], [r[1] for r in rows]
    return decode_at


def test_walk_temperatures():
    LOOP, FINE, FINE2 = [7] * 60, list(range(40)), list(range(40, 70))
    detok = lambda ids: " ".join(f"w{int(i)}" for i in ids)      # noqa: E731
    assert fb.compression_ratio(detok(LOOP)) > 2.4 > fb.compression_ratio(detok(FINE))
    table = {0.0: [(FINE, -0.3), (LOOP, -0.2), (FINE, -1.5), (LOOP, -0.2)],
             0.2: [(None, 0), (FINE2, -0.4), (FINE, -1.2), (LOOP, -0.6)],
             0.4: [(None, 0), (None, 0), (FINE2, -0.9), (LOOP, -0.5)]}
    table = {t: [(r[0] if r[0] is not None else [], r[1]) for r in rows] for t, rows in table.items()}
    calls = []
    o = fb.FallbackOptions(temperatures=(0.0, 0.2, 0.4))
    res = fb.walk_temperatures(_fake_decoder(table, calls), 4, detok, o)
    assert [c[0] for c in calls] == [0.0, 0.2, 0.4]
    assert calls[0][1].tolist() == [True] * 4 and calls[1][1].tolist() == [False, True, True, True] and calls[2][1].tolist() == [False, False, True, True]
    assert res[0].temperature == 0.0 and list(res[0].ids) == FINE and len(res[0].tried) == 1           # passes at once
    assert res[1].temperature == 0.2 and list(res[1].ids) == FINE2 and len(res[1].tried) == 2          # repetition loop, then fine
    assert res[2].temperature == 0.4 and abs(res[2].avg_logprob + 0.9) < 1e-9                          # low probability twice
    # window 3 fails everywhere (compression ratio): "best" = highest avg_logprob among all, "last" = the last attempt
    assert res[3].temperature == 0.0 and len(res[3].tried) == 3
    last = fb.walk_temperatures(_fake_decoder(table, []), 4, detok, fb.FallbackOptions(temperatures=(0.0, 0.2, 0.4), final="last"))
    assert last[3].temperature == 0.4
    # results below the compression threshold are preferred to better-scoring ones above it
    table2 = {0.0: [(LOOP, -0.1)], 0.5: [(FINE, -1.4)]}
    r2 = fb.walk_temperatures(_fake_decoder(table2, []), 1, detok, fb.FallbackOptions(temperatures=(0.0, 0.5)))
    assert r2[0].temperature == 0.5 and list(r2[0].ids) == FINE
    # thresholds switched off: the first decode is final
    calls = []
    r3 = fb.walk_temperatures(_fake_decoder(table, calls), 4, detok,
                              fb.FallbackOptions(temperatures=(0.0, 0.2), compression_ratio_threshold=None, log_prob_threshold=None))
    assert len(calls) == 1 and all(r.temperature == 0.0 for r in r3)
    # a single temperature is "no fallback"
    assert not fb.FallbackOptions(temperatures=(0.0,)).enabled and fb.FallbackOptions.from_asr_options({}).enabled is False
    assert fb.FallbackOptions.from_asr_options({"temperatures": [0, 0.4], "best_of": 3, "beam_size": 5}).rows_per_window() == 5
    with pytest.raises(ValueError):
        fb.FallbackOptions(temperatures=())
    with pytest.raises(ValueError):
        fb.FallbackOptions(temperatures=(0.0, -0.1))


def test_walk_temperatures_equals_a_plain_loop_restatement():
    """Against a per-window restatement written the way the published code reads (one window at a time)."""
    rng = np.random.default_rng(5)
    temps = (0.0, 0.2, 0.4, 0.6)
    detok = lambda ids: " ".join(f"w{int(i)}" for i in ids)      # noqa: E731
    for trial in range(40):
        n = int(rng.integers(1, 6))
        table = {}
        for t in temps:
            rows = []
            for _ in range(n):
                ids = [7] * 50 if rng.random() < 0.4 else [int(x) for x in rng.integers(0, 500, 30)]
                rows.append((ids, float(-rng.random() * 2)))
            table[t] = rows
        for final in ("best", "last"):
            o = fb.FallbackOptions(temperatures=temps, final=final)
            got = fb.walk_temperatures(_fake_decoder(table, []), n, detok, o)
            for i in range(n):
                attempts, chosen = [], None
                for t in temps:
                    ids, avg = table[t][i]
                    cr = fb.compression_ratio(detok(ids))
                    attempts.append((t, ids, avg, cr))
                    if not (cr > 2.4 or avg < -1.0):
                        chosen = attempts[-1]
                        break
                if chosen is None:
                    if final == "last":
                        chosen = attempts[-1]
                    else:
                        below = [a for a in attempts if a[3] < 2.4]
                        chosen = max(below or attempts, key=lambda a: a[2])
                assert got[i].temperature == chosen[0] and list(got[i].ids) == chosen[1], (trial, i, final)


# ------------------------------------------------------------------------------------------------ GPU
def _engine(name, max_batch, max_new, eot_from=None):
    from manual_whisper_amd.engine import Engine
    dims = mw.get_dims(name)
    w = synth.make_weights_np(dims, 1234)
    eng = Engine(dims, 0, max_batch)
    eng.load_weights(w)
    return dims, w, eng


def _check_against_own_logits(eng, o, n_audio, best_of, T, seed, smp, sl, slp, logits):
    """Every draw of the engine, redone on the host from the engine's own logits: same ids (gaps above TIE), same sums."""
    sup = np.zeros(eng.dims.vocab, bool)
    sup[list(o.effective_suppress())] = True
    bsup = sup.copy()
    bsup[list(o.begin_suppress)] = True
    R = n_audio * best_of
    ids = smp.reshape(R, -1)
    lens = sl.reshape(R)
    total = np.zeros(R)
    for i in range(ids.shape[1]):
        live = i <= np.minimum(lens, ids.shape[1] - 1)          # positions up to and including the end-of-text draw
        if not live.any():
            break
        tok, lp, margin = sample_from_logits(logits[i], bsup if i == 0 else sup, T, seed, i)
        for r in np.flatnonzero(live):
            if margin[r] > TIE:
                assert tok[r] == ids[r, i], (r, i, tok[r], ids[r, i], margin[r])
            row = np.where(bsup if i == 0 else sup, -np.inf, logits[i][r].astype(np.float64))
            total[r] += row[ids[r, i]] - (row.max() + np.log(np.exp(row - row.max()).sum()))
    assert np.abs(total - slp.reshape(R)).max() < 2e-3, np.abs(total - slp.reshape(R)).max()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["micro", "tiny"])
def test_sampling_matches_oracle(name):
    from manual_whisper_amd.engine import EngineError
    dims, w, eng = _engine(name, 32, 10)
    try:
        orc = WhisperOracle(dims, w)
        pcm = synth.make_pcm_np(3, seed=1)
        mel = log_mel(pcm, dims.n_mels)
        enc = orc.encode(mel)
        o = synthetic_options(dims, max_new=10)
        eng.set_decode_options(o)
        eng.encode(3, mel=mel, want_output=False)
        # temperature 0 = greedy decoding with its log-probabilities
        g_ids, g_lens = eng.decode_greedy(3)[:2]
        z_ids, z_lens, z_avg = eng.decode_sample(3, 0.0)
        assert (z_ids == g_ids).all() and (z_lens == g_lens).all()
        _, _, b_avg = eng.decode_beam(3, 1)
        assert np.abs(z_avg - b_avg).max() < 1e-4
        _, _, o_avg, _ = orc.sample(enc, list(o.prompt), 10, o.eot, 0.0, 1, 0, o.suppress, o.begin_suppress)
        assert np.abs(z_avg - o_avg).max() < LP_ATOL
        # declare a token that greedy decoding emits early to be end-of-text, so that samples end at different lengths
        eot2 = int(g_ids[0, 3])
        o2 = DecodeOptions(o.prompt, eot2, tuple(t for t in o.suppress if t != eot2), (), 12)
        eng.set_decode_options(o2)
        ended_early = 0
        for graph, prefill in ((0, 1), (1, 1), (1, 0)):
            eng.set_option("graph", graph)
            eng.set_option("prefill", prefill)
            for T, best_of, seed in ((0.2, 1, 1), (0.6, 5, 2), (1.0, 8, 77)):
                ids, lens, avg, smp, sl, slp, logits = eng.decode_sample(3, T, best_of, seed, want_samples=True, want_logits=True)
                _check_against_own_logits(eng, o2, 3, best_of, T, seed, smp, sl, slp, logits)
                ended_early += int((sl < 12).sum())
                # without the logits (ended sequences skipped) the ids are the same
                ids2, lens2, avg2, smp2, sl2, slp2 = eng.decode_sample(3, T, best_of, seed, want_samples=True)
                assert (smp2 == smp).all() and (sl2 == sl).all() and np.abs(slp2 - slp).max() < 1e-4
                for a in range(3):                                  # the recording's result is its best sample per token
                    j = int(np.argmax(slp[a] / np.maximum(sl[a], 1)))
                    assert (ids[a] == smp[a, j]).all() and lens[a] == sl[a, j] and abs(avg[a] - slp[a, j] / (sl[a, j] + 1)) < 1e-5
                    assert (ids[a, lens[a]:] == eot2).all()
                # against the oracle's own run: identical draws until the first one the model noise could flip
                w_ids, w_lens, w_avg, (w_smp, w_sl, w_slp) = orc.sample(enc, list(o2.prompt), 12, eot2, T, best_of, seed, o2.suppress, o2.begin_suppress)
                margins = orc.last_sample_margins.reshape(3, best_of, 12)
                whole = 0
                for a in range(3):
                    for j in range(best_of):
                        close = np.flatnonzero(margins[a, j] < MODEL_TIE / T)
                        upto = int(close[0]) if len(close) else 12
                        assert (smp[a, j, :upto] == w_smp[a, j, :upto]).all(), (T, a, j, upto)
                        if upto == 12:
                            whole += 1
                            assert sl[a, j] == w_sl[a, j] and abs(slp[a, j] - w_slp[a, j]) < LP_ATOL
                assert whole > 0 or best_of == 1, "no sample could be compared in full: the test lost its point"
        assert ended_early > 0, "the crafted end-of-text never fired"
        eng.set_option("graph", 1)
        eng.set_option("prefill", 1)
        # recordings switched off come back empty; the others do not notice
        full = eng.decode_sample(3, 0.8, 4, 5, want_samples=True)
        part = eng.decode_sample(3, 0.8, 4, 5, active=[True, False, True], want_samples=True)
        assert part[1][1] == 0 and (part[0][1] == eot2).all() and (part[4][1] == 0).all()
        assert (part[3][[0, 2]] == full[3][[0, 2]]).all() and (part[0][[0, 2]] == full[0][[0, 2]]).all()
        none = eng.decode_sample(3, 0.8, 4, 5, active=[False, False, False])
        assert (none[1] == 0).all()
        # same seed, same ids; the samples of a recording do not depend on what else shares the batch only through seeds:
        again = eng.decode_sample(3, 0.8, 4, 5, want_samples=True)
        assert (again[3] == full[3]).all()
        for bad in (dict(n_audio=3, temperature=0.5, best_of=9), dict(n_audio=3, temperature=0.5, best_of=0),
                    dict(n_audio=3, temperature=-1.0), dict(n_audio=2, temperature=0.5), dict(n_audio=3, temperature=0.5, best_of=8, active=[1, 1])):
            with pytest.raises(EngineError):
                eng.decode_sample(**bad)
        eng.encode(3, mel=mel, want_output=False)
        with pytest.raises(EngineError):
            eng_small = eng.fork(8)
            try:
                eng_small.set_decode_options(o2)
                eng_small.encode(3, mel=mel, want_output=False)
                eng_small.decode_sample(3, 0.5, 4)                  # 12 sequences do not fit 8
            finally:
                eng_small.close()
    finally:
        eng.close()


@pytest.mark.gpu
def test_sampling_behind_a_long_prompt():
    """The reference's initial_prompt makes the prompt dozens of ids long: the prefilled prompt (one pass, every sample
    reading its recording's cross K/V) and the stepped one give the same draws, the oracle's."""
    dims, w, eng = _engine("micro", 24, 8)
    try:
        orc = WhisperOracle(dims, w)
        pcm = synth.make_pcm_np(2, seed=2)
        mel = log_mel(pcm, dims.n_mels)
        enc = orc.encode(mel)
        base = synthetic_options(dims, max_new=8)
        rng = np.random.default_rng(41)
        prompt = tuple(int(t) for t in rng.integers(0, dims.vocab, size=40)) + (base.prompt[-1],)
        o = DecodeOptions(prompt, base.eot, base.suppress, base.begin_suppress, 8)
        eng.set_decode_options(o)
        eng.encode(2, mel=mel, want_output=False)
        runs = {}
        for prefill in (0, 1):
            eng.set_option("prefill", prefill)
            out = eng.decode_sample(2, 0.5, 3, 13, want_samples=True, want_logits=True)
            _check_against_own_logits(eng, o, 2, 3, 0.5, 13, out[3], out[4], out[5], out[6])
            runs[prefill] = out
        w_smp = orc.sample(enc, list(prompt), 8, o.eot, 0.5, 3, 13, o.suppress, o.begin_suppress)[3][0]
        margins = orc.last_sample_margins.reshape(2, 3, 8)
        for a in range(2):
            for j in range(3):
                close = np.flatnonzero(margins[a, j] < MODEL_TIE / 0.5)
                upto = int(close[0]) if len(close) else 8
                for prefill in (0, 1):
                    assert (runs[prefill][3][a, j, :upto] == w_smp[a, j, :upto]).all()
    finally:
        eng.close()


@pytest.mark.gpu
def test_model_transcribe_with_temperature_fallback():
    """asr_options naming several temperatures: Model.transcribe walks them per window.  Thresholds that nothing
    passes force every temperature to run and the final rule to pick; thresholds off leave greedy decoding."""
    audio = synth.make_pcm_np(5, seed=44).reshape(-1)[: 16000 * 140]           # 140 s -> 5 chunks
    plain = mw.load_model("synthetic:micro", "cuda", asr_options={"max_new_tokens": 8}, max_batch=16)
    try:
        want = plain.transcribe(audio, batch_size=16, return_tokens=True)
    finally:
        plain.close()
    off = mw.load_model("synthetic:micro", "cuda", max_batch=16, n_engines=2,
                        asr_options={"max_new_tokens": 8, "temperatures": [0.0, 0.5], "best_of": 3,
                                     "compression_ratio_threshold": None, "log_prob_threshold": None})
    try:
        assert off.fallback is not None and not off.grouped
        got = off.transcribe(audio, batch_size=16, return_tokens=True)          # passes of 16 // 3 = 5 windows
        assert [s["tokens"] for s in got["segments"]] == [s["tokens"] for s in want["segments"]]
    finally:
        off.close()
    hot = mw.load_model("synthetic:micro", "cuda", max_batch=16, n_engines=1,
                        asr_options={"max_new_tokens": 8, "temperatures": [0.0, 0.5, 1.0], "best_of": 3, "log_prob_threshold": 0.0})
    try:
        chunks, _ = hot.windows(audio)
        eng = hot.engine
        eng.set_decode_options(hot.decode_options("zh"))
        ids, lens, res = fb.transcribe_batch(eng, chunks[:5], hot.detokenize, hot.fallback)
        for i, r in enumerate(res):
            assert [t for t, _, _ in r.tried] == [0.0, 0.5, 1.0]               # an average log-probability is never >= 0
            below = [a for a in r.tried if a[2] < 2.4] or r.tried
            assert abs(r.avg_logprob - max(a[1] for a in below)) < 1e-6
            assert (ids[i, : lens[i]] == r.ids).all()
        got = hot.transcribe(audio, batch_size=16, return_tokens=True)
        assert [s["tokens"] for s in got["segments"]] == [[int(t) for t in r.ids] for r in res]
        # the temperature-0 attempt is the greedy result
        for i, r in enumerate(res):
            eng.logmel(chunks[i:i + 1], want_output=False)
            eng.encode(1, want_output=False)
            _, _, avg = eng.decode_sample(1, 0.0)
            assert abs(r.tried[0][1] - float(avg[0])) < 1e-4
    finally:
        hot.close()
