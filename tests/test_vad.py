"""Speech-activity windows (the step before the hot path): host logic on known answers, the HIP
frame-energy kernel against its numpy restatement, and the windows end to end on the GPU."""
import numpy as np
import pytest

import manual_whisper_amd as mw
from manual_whisper_amd import synth, vad
from oracle.vad_ref import frame_energy_db


# ---------------------------------------------------------------- host logic (no GPU)
def test_binarize_hysteresis_known_answer():
    y = [0.1, 0.6, 0.7, 0.4, 0.3, 0.2, 0.8, 0.9]
    got = vad.binarize(y, onset=0.5, offset=0.363, hop_s=0.01)
    assert got == pytest.approx([(0.015, 0.045), (0.065, 0.075)])
    # 0.4 lies between offset and onset: it neither opens nor closes a region
    assert vad.binarize([0.4, 0.4, 0.4], 0.5, 0.363, 0.01) == []
    assert vad.binarize([0.9, 0.4, 0.4], 0.5, 0.363, 0.01) == pytest.approx([(0.005, 0.025)])
    assert vad.binarize([], 0.5, 0.363) == []


def test_binarize_cuts_long_regions_at_the_weakest_frame_of_the_second_half():
    y = np.full(40, 0.9)
    y[25] = 0.6
    y[5] = 0.55                       # weaker, but in the first half: not a cut candidate
    got = vad.binarize(y, 0.5, 0.363, hop_s=1.0, max_duration=30.0)
    assert got == pytest.approx([(0.5, 25.5), (25.5, 39.5)])
    assert all(b - a <= 30.0 for a, b in got)


def test_merge_chunks_known_answer():
    regions = [(0, 10), (12, 20), (25, 33), (50, 55)]
    got = vad.merge_chunks(regions, 30.0)
    assert [(w["start"], w["end"]) for w in got] == [(0.0, 20.0), (25.0, 55.0)]
    assert got[0]["segments"] == [(0.0, 10.0), (12.0, 20.0)] and got[1]["segments"] == [(25.0, 33.0), (50.0, 55.0)]
    assert vad.merge_chunks([], 30.0) == []
    one = vad.merge_chunks([(3.5, 4.0)], 30.0)
    assert [(w["start"], w["end"]) for w in one] == [(3.5, 4.0)]


def test_speech_score_and_cut_windows():
    e = np.full(3000, -70.0, np.float32)
    e[500:1500] = -20.0               # 10 s of speech inside 30 s of near silence
    s = vad.speech_score(e)
    assert s[:400].max() < 0.05 and s[600:1400].min() > 0.95 and s[1700:].max() < 0.05
    regions = vad.binarize(s)
    assert len(regions) == 1 and abs(regions[0][0] - 5.0) < 0.1 and abs(regions[0][1] - 15.0) < 0.1
    assert vad.speech_score(np.full(100, -80.0, np.float32)).max() == 0.0      # flat and quiet: silence
    assert vad.speech_score(np.full(100, -20.0, np.float32)).min() == 1.0      # flat and loud: speech
    audio = np.arange(16000 * 40, dtype=np.float32)
    chunks, spans = vad.cut_windows(audio, [{"start": 1.0, "end": 3.0}, {"start": 5.0, "end": 39.0}])
    assert chunks.shape == (2, 480000) and spans == [(1.0, 3.0), (5.0, 39.0)]
    assert chunks[0, 0] == 16000 and chunks[0, 31999] == 47999 and chunks[0, 32000] == 0
    assert chunks[1, 0] == 80000 and chunks[1, -1] == 80000 + 479999           # clipped to 30 s


def test_oracle_energy_known_answer():
    x = np.full(1000, 0.1, np.float32)
    e = frame_energy_db(x, 400, 160)
    assert e.shape == (7,) and np.allclose(e, -20.0, atol=1e-4)                # ragged last frames average what exists
    assert np.allclose(frame_energy_db(np.zeros(320, np.float32), 400, 160), -100.0)


# ---------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def micro_engine():
    from manual_whisper_amd.engine import Engine
    dims = mw.get_dims("micro")
    eng = Engine(dims, 0, 4)
    eng.load_weights(synth.make_weights_np(dims, 1234))
    yield eng
    eng.close()


@pytest.mark.gpu
def test_frame_energy_matches_oracle(micro_engine):
    rng = np.random.default_rng(5)
    for n in (1, 159, 160, 161, 400, 48000 + 37, 480000):
        x = (rng.standard_normal(n) * np.exp(rng.uniform(-6, 0))).astype(np.float32)
        got = micro_engine.frame_energy_db(x)
        want = frame_energy_db(x)
        assert got.shape == want.shape
        assert np.abs(got - want).max() < 1e-3, n
    assert micro_engine.frame_energy_db(np.zeros(0, np.float32)).shape == (0,)
    x = (rng.standard_normal(20000) * 0.1).astype(np.float32)                   # other frame / hop
    assert np.abs(micro_engine.frame_energy_db(x, 1024, 256) - frame_energy_db(x, 1024, 256)).max() < 1e-3


@pytest.mark.gpu
def test_frame_energy_long_recording_and_device_input(micro_engine):
    """18 M samples (19 min): host input is staged in pieces, device input runs in one launch."""
    import torch
    n = 18_000_000 + 123
    t = torch.arange(n, dtype=torch.float32, device="cuda")
    x_dev = (torch.sin(t * 0.01) * (0.05 + 0.4 * (torch.sin(t * 1e-5) > 0))).contiguous()
    x = x_dev.cpu().numpy()
    want = frame_energy_db(x)
    got_host = micro_engine.frame_energy_db(x)
    got_dev = micro_engine.frame_energy_db(x_dev)
    assert np.abs(got_host - want).max() < 1e-3
    assert np.array_equal(got_host, got_dev)                                    # piece seams change nothing


@pytest.mark.gpu
def test_frame_energy_rejects_bad_arguments(micro_engine):
    from manual_whisper_amd.engine import EngineError
    x = np.zeros(1000, np.float32)
    with pytest.raises(EngineError):
        micro_engine.frame_energy_db(x, frame=100, hop=160)                     # frame < hop


@pytest.mark.gpu
def test_transcribe_with_speech_windows():
    """A recording with two bursts separated by silence: two windows, each transcribed exactly as
    the same audio handed to the engine directly; silence produces no segment.  The second burst is the longer
    one, so the scheduler (longest windows first) hands them to the engine in the other order and must put the
    results back."""
    model = mw.load_model("synthetic:micro", "cuda", compute_type="float16", language="zh",
                          vad_options={"vad_onset": 0.5, "vad_offset": 0.363}, max_batch=4)
    try:
        sr = 16000
        burst = synth.make_pcm_np(2, seed=11)[:, : 8 * sr]
        audio = np.zeros(40 * sr, np.float32)
        audio[2 * sr: 7 * sr] = burst[0][: 5 * sr]
        audio[24 * sr: 32 * sr] = burst[1]
        res = model.transcribe(audio, batch_size=4, return_tokens=True)
        segs = res["segments"]
        assert len(segs) == 2
        assert abs(segs[0]["start"] - 2.0) < 0.2 and abs(segs[0]["end"] - 7.0) < 0.2
        assert abs(segs[1]["start"] - 24.0) < 0.2 and abs(segs[1]["end"] - 32.0) < 0.2
        chunks, spans = model.windows(audio)
        assert spans == [(s["start"], s["end"]) for s in segs]
        ids, lens = model.engine.transcribe_batch(chunks)
        for i, s in enumerate(segs):
            assert s["tokens"] == [int(t) for t in ids[i, : lens[i]]]
        assert model.transcribe(np.zeros(5 * sr, np.float32))["segments"] == []
        model.vad_options = {"vad_method": "fixed"}
        fixed = model.transcribe(audio, batch_size=4)
        assert len(fixed["segments"]) == 2         # ceil(40 s / 30 s)
        # the reference's result shape (transcribe.py:162-176,221-223): nothing but text, start, end
        assert set(fixed) == {"segments", "language"} and all(set(s) == {"text", "start", "end"} for s in fixed["segments"])
    finally:
        model.close()
