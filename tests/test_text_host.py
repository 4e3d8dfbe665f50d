"""Host text path against the reference's own outputs (tests/golden/text_cases.json, produced by
tools/make_text_golden.py from reference transcribe.py:64-95,186-226)."""
import json
import os

import pytest

from manual_whisper_amd import emit, postprocess


@pytest.fixture(scope="module")
def cases(golden_dir):
    with open(os.path.join(golden_dir, "text_cases.json"), encoding="utf-8") as f:
        return json.load(f)


def test_remove_hallucination_loops(cases):
    for src, want in cases["remove_hallucination_loops"]:
        assert postprocess.remove_hallucination_loops(src) == want, src
    for src, want in cases["remove_hallucination_loops_max2"]:
        assert postprocess.remove_hallucination_loops(src, 2) == want, src


def test_loop_rule_known_answers():
    f = postprocess.remove_hallucination_loops
    assert f("那个吧" * 5) == "那个吧"
    assert f("好的" * 3) == "好的" * 3          # needs >= 4 copies
    assert f("好的" * 4) == "好的"
    assert f("哈" * 6) == "哈" * 6              # unit must be >= 2 characters
    assert f("我们 我们 我们 我们 开会") == "我们 开会"


def test_clean_segment_text(cases):
    for src, want in cases["clean_segment_text"]:
        assert postprocess.clean_segment_text(src) == want, repr(src)


def test_format_timestamp(cases):
    for sec, want in cases["format_timestamp"]:
        assert emit.format_timestamp(sec) == want, sec


def test_format_transcript_files(cases, tmp_path):
    c = cases["constants"]
    for case in cases["format_transcript"]:
        a = case["args"]
        out = tmp_path / "out_transcript.md"
        emit.format_transcript(json.loads(json.dumps(a["result"])), a["audio_file"], str(out), a["audio_duration"],
                               a["total_time"], model_size=c["MODEL_SIZE"], compute_type=c["COMPUTE_TYPE"],
                               device=c["DEVICE"], quiet=True)
        assert out.read_text(encoding="utf-8") == case["md"]
        assert (tmp_path / "out_transcript.json").read_text(encoding="utf-8") == case["json"]


def test_clean_segments_drop_and_count():
    segs = [{"text": "请使用简体中文。", "start": 0, "end": 30}, {"text": "，，搜码二维码！！", "start": 30, "end": 60},
            {"text": "正常", "start": 60, "end": 90}, {"start": 90, "end": 120}]
    kept, changed, dropped = postprocess.clean_segments(segs)
    assert [s["text"] for s in kept] == ["收款二维码", "正常"]
    assert (changed, dropped) == (1, 2)
