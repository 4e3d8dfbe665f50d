"""Forced alignment (reference transcribe.py:127-135).

CPU: the numpy oracle of the wav2vec2 CTC model against vectors made by transformers itself
(tests/golden/w2v_micro.npz, tools/make_align_golden.py); the host alignment logic against the oracle's loop
restatement, an exhaustive search and hand-computed cases.  GPU: mw_align_emissions against the oracle.
Parity with whisperx itself is unpinned (it is not available offline and the reference holds no alignment vectors).
"""
import ctypes as C
import math
import os

import numpy as np
import pytest

import manual_whisper_amd as mw  # noqa: F401
from manual_whisper_amd import _lib, alignment as al, localckpt, synth
from oracle import wav2vec2_ref as ref

CFG = localckpt.wav2vec2_config("w2v-micro")


def _wave(tag, n, seed=9):
    return synth.hash_uniform_np(n, f"wave.{tag}", seed) * np.float32(0.6)


@pytest.fixture(scope="module")
def micro_weights():
    return localckpt.make_wav2vec2_weights_np(CFG, 4242)


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "w2v_micro.npz"))


# ---------------------------------------------------------------- oracle vs transformers' own output
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_oracle_matches_transformers_vectors(tag, micro_weights, golden):
    n = int(golden[f"n_{tag}"])
    o = ref.Wav2Vec2Oracle(CFG, micro_weights)
    logits = o.logits(_wave(tag, n))
    assert logits.shape == golden[f"logits_{tag}"].shape == (ref.n_frames(CFG, n), CFG["vocab_size"])
    np.testing.assert_allclose(logits, golden[f"logits_{tag}"], atol=2e-4, rtol=0)
    np.testing.assert_allclose(o.emissions(_wave(tag, n)), golden[f"emissions_{tag}"], atol=2e-4, rtol=0)


def test_frame_count_of_the_library_matches_the_oracle():
    lib = _lib.load()
    d = _lib.AlignDesc()
    d.n_conv = 7
    for i, (k, s) in enumerate(zip(CFG["conv_kernel"], CFG["conv_stride"])):
        d.conv_kernel[i], d.conv_stride[i] = k, s
    for n in (0, 399, 400, 719, 720, 6000, 19200, 480000):
        assert lib.mw_align_frames(C.byref(d), n) == ref.n_frames(CFG, n)
    assert ref.n_frames(CFG, 400) == 1 and ref.n_frames(CFG, 480000) == 1499


# ---------------------------------------------------------------- trellis / backtracking
def _random_emission(T, V, seed):
    rng = np.random.default_rng(seed)
    z = rng.normal(size=(T, V)).astype(np.float32) * 2
    z = z - z.max(-1, keepdims=True)
    return (z - np.log(np.exp(z).sum(-1, keepdims=True))).astype(np.float32)


@pytest.mark.parametrize("T,tokens", [(6, [3, 1, 2]), (9, [1, 1, 4, 2]), (7, [2]), (8, [3, -1, 2, -1]), (5, [1, 2, 3, 4, 1]), (40, [1, 2, -1, 3, 3, 4, 1, 2])])
def test_path_is_the_exhaustive_optimum_and_matches_the_loop_restatement(T, tokens):
    """mw_align_path (C, host side of the library) against the oracle's loop restatement and, on the small cases,
    an exhaustive search over every alignment."""
    em = _random_emission(T, 5, seed=T * 31 + len(tokens))
    tr = ref.get_trellis(em, tokens, 0)
    path = al.ctc_path(em, tokens, 0)
    assert [p.time_index for p in path] == list(range(T))                      # one point per frame
    assert path[0].token_index == 0 and path[-1].token_index == len(tokens) - 1
    steps = np.diff([p.token_index for p in path])
    assert set(steps.tolist()) <= {0, 1}
    if T <= 9:
        best, changes = ref.best_alignment_bruteforce(em, tokens, 0)
        assert tr[-1, -1] == pytest.approx(best, abs=1e-4)
        # the walk returns the optimum: token j is entered at frame c_j + 1 (the move is paid at frame c_j)
        assert tuple(int(t) for t in np.nonzero(steps)[0]) == tuple(changes)
    want = ref.backtrack(tr, em, tokens, 0)
    assert [(p.token_index, p.time_index) for p in path] == [(p.token_index, p.time_index) for p in want]
    assert [p.score for p in path] == pytest.approx([p.score for p in want], rel=1e-5)


def test_path_fails_when_there_are_fewer_frames_than_tokens():
    em = _random_emission(2, 4, 1)
    assert al.ctc_path(em, [1, 2, 3], 0) is None
    assert ref.backtrack(ref.get_trellis(em, [1, 2, 3], 0), em, [1, 2, 3], 0) is None
    with pytest.raises(mw.EngineError):
        al.ctc_path(em, [1, 9], 0)                       # id outside the vocabulary


def test_wildcard_scores_the_best_non_blank_entry():
    frame = np.log(np.array([[0.7, 0.1, 0.15, 0.05], [0.2, 0.1, 0.6, 0.1], [0.7, 0.1, 0.15, 0.05]], np.float32))
    path = al.ctc_path(frame, [1, -1], 0)
    # the wildcard is entered where the best non-blank entry (0.6, label 2) is: paid at frame 1
    assert [p.token_index for p in path] == [0, 0, 1] and path[1].score == pytest.approx(0.6, rel=1e-5)
    assert ref.token_logp(frame[0], -1, 0) == pytest.approx(math.log(0.15))


def test_merge_repeats_known_answer():
    P = al.Point
    path = [P(0, 0, 0.5), P(0, 1, 0.7), P(1, 2, 0.9), P(2, 3, 0.2), P(2, 4, 0.4), P(2, 5, 0.6)]
    got = al.merge_repeats(path, "abc")
    assert [(s.label, s.start, s.end) for s in got] == [("a", 0, 2), ("b", 2, 3), ("c", 3, 6)]
    assert [s.score for s in got] == pytest.approx([0.6, 0.9, 0.4])
    want = ref.merge_repeats([ref.Point(p.token_index, p.time_index, p.score) for p in path], "abc")
    assert [(s.label, s.start, s.end, s.score) for s in want] == [(s.label, s.start, s.end, pytest.approx(s.score)) for s in got]


# ---------------------------------------------------------------- text handling
def test_sentence_spans():
    assert al.sentence_spans("今天开会讨论支付宝") == [(0, 9)]
    assert al.sentence_spans(" Hello there. How are you? Fine") == [(1, 13), (14, 26), (27, 31)]
    assert al.sentence_spans("Wait... what?! ok ") == [(0, 7), (8, 14), (15, 17)]
    assert al.sentence_spans("") == [] and al.sentence_spans("   ") == []


def test_interpolate_nans():
    nan = math.nan
    assert al.interpolate_nans([nan, 1.0, nan, nan, 4.0, nan]) == [1.0, 1.0, 1.0, 4.0, 4.0, 4.0]
    assert al.interpolate_nans([nan, 1.0, nan, nan, 4.0], "linear") == [1.0, 1.0, 2.0, 3.0, 4.0]
    assert al.interpolate_nans([nan, 2.5, nan]) == [2.5, 2.5, 2.5]
    assert all(math.isnan(v) for v in al.interpolate_nans([nan, nan]))


class _ScriptedModel:
    """Stands in for the acoustic model: emissions with the scripted label near-certain in every frame."""

    def __init__(self, vocab, frames):
        self.vocab, self.frames = vocab, frames

    def emissions(self, wave):
        em = np.full((len(self.frames), len(self.vocab)), math.log(1e-4), np.float32)
        for t, ch in enumerate(self.frames):
            em[t, self.vocab[ch]] = math.log(1 - 1e-4 * (len(self.vocab) - 1))
        return em


def test_align_chinese_segment_characters_become_words():
    vocab = {"<pad>": 0, "<unk>": 1, "你": 2, "好": 3, "吗": 4}
    meta = {"language": "zh", "dictionary": vocab, "type": "huggingface"}
    # 10 frames over a 2 s segment that starts at 10 s: 0.2 s per ... (T - 1) = 9 intervals
    model = _ScriptedModel(vocab, ["<pad>", "<pad>", "好", "<pad>", "<pad>", "吗", "<pad>", "你", "<pad>", "<pad>"])
    audio = np.zeros(16000 * 13, np.float32)
    out = al.align([{"text": " 你好吗？", "start": 10.0, "end": 12.0}], model, meta, audio, "cuda")
    assert list(out) == ["segments", "word_segments"]
    seg = out["segments"][0]
    assert seg["text"] == "你好吗？"
    words = seg["words"]
    assert [w["word"] for w in words] == ["你", "好", "吗", "？"] and out["word_segments"] == words
    r = 2.0 / 9
    # 好 is paid at frame 2 (entered at 3), 吗 at frame 5 (entered at 6); "？" is outside the vocabulary: a wildcard
    # that takes the best non-blank entry (frame 7, entered at 8); the path ends on it
    assert words[0]["start"] == pytest.approx(10.0) and words[0]["end"] == pytest.approx(round(10 + 3 * r, 3))
    assert words[1]["start"] == pytest.approx(round(10 + 3 * r, 3)) and words[1]["end"] == pytest.approx(round(10 + 6 * r, 3))
    assert words[2]["start"] == pytest.approx(round(10 + 6 * r, 3)) and words[2]["end"] == pytest.approx(round(10 + 8 * r, 3))
    assert words[3]["end"] == pytest.approx(round(10 + 10 * r, 3))
    assert seg["start"] == words[0]["start"] and seg["end"] == words[3]["end"]
    assert all(0.0 <= w["score"] <= 1.0 for w in words)


def test_align_spaced_language_words_and_sentences():
    vocab = {"<pad>": 0, "|": 1, "a": 2, "b": 3, "c": 4}
    meta = {"language": "en", "dictionary": vocab, "type": "huggingface"}
    frames = ["<pad>", "b", "a", "|", "c", "<pad>", "|", "a", "<pad>", "<pad>", "<pad>"]
    model = _ScriptedModel(vocab, frames)
    out = al.align([{"text": "Ab. c a", "start": 0.0, "end": 1.0}], model, meta, np.zeros(16000, np.float32), "cuda",
                   return_char_alignments=True)
    # two sentences ("Ab." and "c a") -> two segments; "." is a wildcard character inside the first word
    assert [s["text"] for s in out["segments"]] == ["Ab.", "c a"]
    assert [w["word"] for w in out["word_segments"]] == ["Ab.", "c", "a"]
    assert out["segments"][0]["end"] <= out["segments"][1]["start"] + 1e-9
    assert [c["char"] for c in out["segments"][0]["chars"]] == ["A", "b", ".", " "]


def test_align_falls_back_to_the_segment_when_nothing_can_be_aligned():
    vocab = {"<pad>": 0, "a": 1}
    meta = {"language": "en", "dictionary": vocab, "type": "huggingface"}
    model = _ScriptedModel(vocab, ["<pad>"] * 4)
    audio = np.zeros(16000, np.float32)
    out = al.align([{"text": "   ", "start": 0.0, "end": 0.5}, {"text": "a", "start": 5.0, "end": 6.0},
                    {"text": "aaaaaaaa", "start": 0.0, "end": 0.5}], model, meta, audio, "cuda")
    assert [s["words"] for s in out["segments"]] == [[], [], []]             # blank text; starts past the audio; 8 tokens in 4 frames
    assert [(s["start"], s["end"]) for s in out["segments"]] == [(0.0, 0.5), (5.0, 6.0), (0.0, 0.5)]


def test_align_keeps_the_other_segments_when_the_model_refuses_one(capsys):
    """A segment the acoustic model cannot take (61 s with an aligner built for 60 s) falls back to its original
    times like every other per-segment failure; the segments around it are aligned."""
    from manual_whisper_amd.engine import EngineError
    vocab = {"<pad>": 0, "a": 1, "b": 2}
    meta = {"language": "en", "dictionary": vocab, "type": "huggingface"}

    class _Limited(_ScriptedModel):
        def emissions(self, wave):
            if len(wave) > 60 * 16000:
                raise EngineError("mw_align_segment failed (-1): n_samples exceeds max_samples")
            return super().emissions(wave)

    model = _Limited(vocab, ["<pad>", "a", "<pad>", "b", "<pad>"])
    audio = np.zeros(16000 * 70, np.float32)
    out = al.align([{"text": "ab", "start": 0.0, "end": 1.0}, {"text": "ab", "start": 2.0, "end": 63.0},
                    {"text": "ab", "start": 64.0, "end": 65.0}], model, meta, audio, "cuda")
    assert [len(s["words"]) for s in out["segments"]] == [1, 0, 1]
    assert (out["segments"][1]["start"], out["segments"][1]["end"]) == (2.0, 63.0)
    assert "max_samples" in capsys.readouterr().out


def test_load_align_model_needs_a_local_directory(monkeypatch):
    monkeypatch.delenv("MW_ALIGN_MODEL", raising=False)
    with pytest.raises(FileNotFoundError, match="MW_ALIGN_MODEL"):
        al.load_align_model("zh", "cuda")


# ---------------------------------------------------------------- GPU: the acoustic model against the oracle
# fp16 operands with fp32 accumulation against an fp32 oracle: log-probabilities agree to 0.03 on the toy model
# (logit spread about 1) and the best label of a frame is the same wherever the oracle's margin exceeds that.
EMIT_ATOL = 0.03


def _check_emissions(got, want, atol):
    assert got.shape == want.shape
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= atol
    np.testing.assert_allclose(np.exp(got.astype(np.float64)).sum(-1), 1.0, atol=1e-4)
    top2 = np.sort(want, -1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 2 * atol
    assert (got.argmax(-1)[clear] == want.argmax(-1)[clear]).all()


@pytest.fixture(scope="module")
def micro_aligner(micro_weights):
    a = al.Aligner(CFG, device=0, max_seconds=4, n_streams=8, pack_seconds=32)
    a.load_weights(micro_weights)
    yield a
    a.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tag,n", [("a", 19200), ("b", 6000), ("c", 400), ("ragged", 12345), ("long", 63999)])
def test_gpu_emissions_match_the_oracle(tag, n, micro_aligner, micro_weights, golden):
    wave = _wave(tag, n)
    got = micro_aligner.emissions(wave)
    want = ref.Wav2Vec2Oracle(CFG, micro_weights).emissions(wave)
    _check_emissions(got, want, EMIT_ATOL)
    if f"emissions_{tag}" in golden.files:
        _check_emissions(got, golden[f"emissions_{tag}"], EMIT_ATOL)


@pytest.mark.gpu
def test_gpu_emissions_are_repeatable_and_independent_of_the_previous_call(micro_aligner):
    a = micro_aligner.emissions(_wave("a", 19200))
    micro_aligner.emissions(_wave("long", 63999))
    b = micro_aligner.emissions(_wave("a", 19200))
    np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
def test_gpu_concurrent_segments_on_forked_workspaces(micro_aligner):
    """Up to n_streams calls run at once, each on its own stream and workspaces: the results are those of one call
    after another."""
    from concurrent.futures import ThreadPoolExecutor
    assert micro_aligner.n_streams == 8
    waves = [_wave(f"s{i}", 4000 + 3777 * i) for i in range(12)]
    one_by_one = [micro_aligner.emissions(w) for w in waves]
    with ThreadPoolExecutor(max_workers=6) as pool:
        together = list(pool.map(micro_aligner.emissions, waves))
    for a, b in zip(one_by_one, together):
        np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
def test_gpu_segment_path_equals_emissions_then_search(micro_aligner):
    """mw_align_segment (columns gathered on the GPU, search on the host) against mw_align_emissions + mw_align_path,
    with wildcards, a single token, and more tokens than frames."""
    wave = _wave("a", 19200)
    em = micro_aligner.emissions(wave)
    for tokens in ([5, 9, -1, 7, 7, 30, -1], [12], list(range(1, 40)) + [-1] * 10, [3] * 70):
        want = al.ctc_path(em, tokens, 0)
        got, n = micro_aligner.segment_path(wave, tokens, 0)
        assert n == em.shape[0]
        if want is None:
            assert got is None
            continue
        assert [(p.token_index, p.time_index) for p in got] == [(p.token_index, p.time_index) for p in want]
        assert [p.score for p in got] == pytest.approx([p.score for p in want], rel=1e-6)
    with pytest.raises(mw.EngineError, match="outside the vocabulary"):
        micro_aligner.segment_path(wave, [40], 0)


@pytest.mark.gpu
def test_gpu_pack_of_segments_equals_one_segment_at_a_time(micro_aligner):
    """mw_align_segments: segments of different lengths through the model in ONE pass (slices of the longest one, the
    positional convolution and the attention on each segment's own frames) give every segment the path and the scores
    it gets alone, whatever shares its pack and wherever it stands in it."""
    lens = [63999, 19200, 400, 12345, 30001, 6000, 47777]
    waves = [_wave(f"p{i}", n) for i, n in enumerate(lens)]
    tokens = [[5, 9, -1, 7, 7, 30, -1], [12], [3], list(range(1, 30)), [3] * 120, [8, -1, 8], [2, 4, 6, 8, 10] * 4]
    alone = [micro_aligner.segment_path(w, t, 0) for w, t in zip(waves, tokens)]
    assert alone[2][0] is not None and alone[4][0] is None          # one frame, one token: a path; 120 tokens in 93 frames: none
    for order in (list(range(7)), [6, 2, 0, 5, 3, 1, 4], [1, 3], [4, 0]):
        ws, ts = [waves[i] for i in order], [tokens[i] for i in order]
        assert micro_aligner.pack_fits([len(w) for w in ws], [len(t) for t in ts]) == (len(order) * max(len(w) for w in ws) <= 32 * 16000)
        if not micro_aligner.pack_fits([len(w) for w in ws], [len(t) for t in ts]):
            continue
        got = micro_aligner.segments_path(ws, ts, 0)
        for i, (path, n) in zip(order, got):
            want, n_want = alone[i]
            assert n == n_want
            if want is None:
                assert path is None
                continue
            assert [(p.token_index, p.time_index) for p in path] == [(p.token_index, p.time_index) for p in want], i
            assert [p.score for p in path] == pytest.approx([p.score for p in want], rel=2e-3, abs=1e-6)
    assert not micro_aligner.pack_fits([63999] * 9, [3] * 9)         # 9 slices of 4 s do not fit 32 s
    with pytest.raises(mw.EngineError, match="max_samples"):
        micro_aligner.segments_path([np.zeros(64001, np.float32)], [[3]], 0)


@pytest.mark.gpu
def test_gpu_align_packs_segments_and_keeps_their_order(micro_aligner):
    """align() over segments of mixed lengths: packed passes (longest first) and one segment at a time (the model
    stripped of its pack interface) give the same result, in the segments' order."""
    vocab = localckpt.synthetic_ctc_vocab(CFG["vocab_size"])
    meta = {"language": "zh", "dictionary": vocab, "type": "huggingface"}
    chars = [c for c in vocab if len(c) == 1 and c.isalpha()]
    rng = np.random.default_rng(5)
    audio = _wave("mix", 16000 * 30, seed=9)
    bounds = [0.0, 3.9, 4.3, 8.2, 9.0, 12.5, 12.9, 16.8, 20.0, 23.5, 27.4, 30.0]
    segs = [{"text": "".join(rng.choice(chars, 1 + int(3 * (b - a)))), "start": a, "end": b} for a, b in zip(bounds[:-1], bounds[1:])]
    packed = al.align(segs, micro_aligner, meta, audio, "cuda")

    class _OneAtATime:
        n_streams = 1

        def segment_path(self, wave, tokens, blank_id=0):
            return micro_aligner.segment_path(wave, tokens, blank_id)

    single = al.align(segs, _OneAtATime(), meta, audio, "cuda")
    assert [s["text"] for s in packed["segments"]] == [s["text"] for s in single["segments"]] == [s["text"] for s in segs]
    assert [(w["word"], w.get("start"), w.get("end")) for w in packed["word_segments"]] == \
           [(w["word"], w.get("start"), w.get("end")) for w in single["word_segments"]]
    assert [w["score"] for w in packed["word_segments"]] == pytest.approx([w["score"] for w in single["word_segments"]], abs=2e-3)


@pytest.mark.gpu
def test_gpu_argument_errors(micro_aligner, micro_weights):
    with pytest.raises(mw.EngineError, match="max_samples"):
        micro_aligner.emissions(np.zeros(16000 * 4 + 1, np.float32))
    with pytest.raises(mw.EngineError, match="shorter than one frame"):
        micro_aligner.emissions(np.zeros(399, np.float32))
    fresh = al.Aligner(CFG, device=0, max_seconds=1)
    try:
        with pytest.raises(mw.EngineError, match="finalize"):
            fresh.emissions(np.zeros(400, np.float32))
        partial = {k: v for k, v in micro_weights.items() if k != "lm_head.bias"}
        with pytest.raises(mw.EngineError, match="lm_head.bias"):
            fresh.load_weights(partial)
    finally:
        fresh.close()
    bad = dict(CFG, hidden_size=96)
    with pytest.raises(mw.EngineError, match="head_dim"):
        al.Aligner(bad, device=0)


@pytest.mark.gpu
def test_gpu_alignment_from_a_checkpoint_directory(tmp_path, micro_weights):
    path = str(tmp_path / "w2v")
    localckpt.write_synthetic_wav2vec2(path, "w2v-micro")
    model, meta = al.load_align_model("zh", "cuda:0", model_name=path)
    try:
        assert meta["type"] == "huggingface" and meta["language"] == "zh" and meta["dictionary"]["<pad>"] == 0
        audio = np.concatenate([_wave("a", 19200), _wave("long", 63999)])
        chars = [c for c in meta["dictionary"] if len(c) == 1 and c.isalpha()][:6]
        assert len(chars) == 6
        segs = [{"text": "".join(chars[:4]) + "，", "start": 0.0, "end": 1.2}, {"text": "".join(chars[2:6]), "start": 1.2, "end": 5.1}]
        out = al.align(segs, model, meta, audio, "cuda:0")

        class _Oracle:
            def __init__(self):
                self.o = ref.Wav2Vec2Oracle(CFG, micro_weights)

            def emissions(self, wave):
                return self.o.emissions(wave)

        want = al.align(segs, _Oracle(), meta, audio, "cuda:0")
        assert [s["text"] for s in out["segments"]] == [s["text"] for s in want["segments"]]
        assert [w["word"] for w in out["word_segments"]] == [w["word"] for w in want["word_segments"]] == list(segs[0]["text"] + segs[1]["text"])
        # the same path wherever the emissions decide clearly: on these near-flat synthetic emissions at least
        # 8 of the 9 words start within two frames (0.04 s) of where the oracle's emissions put them
        close = sum(abs(a["start"] - b["start"]) <= 0.0401 for a, b in zip(out["word_segments"], want["word_segments"]))
        assert close >= len(want["word_segments"]) - 1
        for seg, src in zip(out["segments"], segs):
            assert src["start"] <= seg["start"] <= seg["end"] <= src["end"] + 0.03        # last frame's end: duration * T / (T - 1)
    finally:
        model.close()


@pytest.mark.gpu
def test_gpu_full_size_model_matches_the_oracle():
    """wav2vec2-large-xlsr-53 geometry (24 layers of 1024, vocabulary 3503) with synthetic weights on 3 s of noise."""
    cfg = localckpt.wav2vec2_config("w2v-large")
    w = localckpt.make_wav2vec2_weights_np(cfg, 77)
    wave = _wave("full", 48000, seed=3)
    a = al.Aligner(cfg, device=0, max_seconds=30)
    try:
        a.load_weights(w)
        got = a.emissions(wave)
        got30 = a.emissions(np.tile(wave, 10))                         # the longest segment the path produces: 30 s
        # a pack at full size: 30 s, 3 s and 11 s in one pass, each as it is alone
        waves = [np.tile(wave, 10), wave, np.tile(wave, 4)[: 16000 * 11]]
        toks = [list(range(5, 105)), [7, 8, 9], list(range(200, 240))]
        alone = [a.segment_path(w_, t_, 0) for w_, t_ in zip(waves, toks)]
        packed = a.segments_path(waves, toks, 0)
        for (p1, n1), (p2, n2) in zip(alone, packed):
            assert n1 == n2 and [(p.token_index, p.time_index) for p in p1] == [(p.token_index, p.time_index) for p in p2]
            assert [p.score for p in p2] == pytest.approx([p.score for p in p1], rel=5e-3, abs=1e-6)
    finally:
        a.close()
    want = ref.Wav2Vec2Oracle(cfg, w).emissions(wave)
    assert got.shape == want.shape == (149, 3503)
    # 24 layers deep and a vocabulary of 3503: agreement to 0.1 in log-probability, best label where the margin is clear
    _check_emissions(got, want, 0.1)
    assert got30.shape == (1499, 3503) and np.isfinite(got30).all()
    np.testing.assert_allclose(np.exp(got30.astype(np.float64)).sum(-1), 1.0, atol=1e-4)


@pytest.mark.gpu
def test_gpu_cli_aligns_segments_when_a_model_directory_is_given(tmp_path):
    """`python transcribe.py x.wav` with MW_ALIGN_MODEL set (reference transcribe.py:127-135): every segment of the
    JSON carries words whose times lie inside the segment's speech window, and the result keeps word_segments."""
    import json
    import subprocess
    import sys
    from manual_whisper_amd import audio as aud
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sr = 16000
    x = synth.make_pcm_np(3, seed=71).reshape(-1)[: sr * 70].copy()      # speech 0-25 s, silence, speech 40-65 s, silence
    x[25 * sr: 40 * sr] = 0.0
    x[65 * sr:] = 0.0
    wav = tmp_path / "meeting.wav"
    aud.write_wav(str(wav), x)
    w2v = str(tmp_path / "w2v")
    localckpt.write_synthetic_wav2vec2(w2v, "w2v-micro")
    env = dict(os.environ, WHISPER_MODEL="synthetic:micro", BATCH_SIZE="2", MW_ENGINES="2", MW_ALIGN_MODEL=w2v)
    r = subprocess.run([sys.executable, os.path.join(root, "transcribe.py"), str(wav)], cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "对齐时间戳" in r.stdout and "对齐耗时" in r.stdout
    js = json.loads((tmp_path / "meeting_transcript.json").read_text(encoding="utf-8"))
    assert len(js["segments"]) == 2 and js["word_segments"]
    for seg, (wa, wb) in zip(js["segments"], [(0.0, 25.0), (40.0, 65.0)]):
        assert seg["words"], seg
        assert wa - 0.2 <= seg["start"] <= seg["end"] <= wb + 0.3
        for w in seg["words"]:
            assert seg["start"] - 1e-6 <= w["start"] <= w["end"] <= seg["end"] + 1e-6 and 0.0 <= w["score"] <= 1.0
        # the synthetic Whisper checkpoint has no tokenizer: the text is "<|id|>..." and every character is a word for "zh"
        assert "".join(w["word"] for w in seg["words"]) == seg["text"].replace(" ", "")
