"""Segment-list post-processing of the reference's web service (web/api/lib/processor.js:52-113).

The expected values of the first tests are worked out by hand from the cited lines; the last test compares with what
the reference's JavaScript itself returned (tests/golden/segment_filter_cases.json, made by tools/make_filter_golden.py).
"""
import copy
import importlib

import pytest

sf = importlib.import_module("manual-whisper_amd.segment_filter")
OPTS = sf.FilterOptions()


def run(segments, **kw):
    return sf.post_process_segments(segments, sf.FilterOptions(**kw))


def test_meeting_excerpt_end_to_end():
    segments = [
        {"text": "  请使用简体中文。 ", "start": 0, "end": 2.5},
        {"text": "嗯。", "start": 2.5, "end": 3.2},
        {"text": "我们 我们 我们 我们 开会", "start": 3.2, "end": 6.0, "speaker": "SPEAKER_00"},
        {"text": "打开AAP里面的二维码，，对吧？？", "start": 6.3, "end": 9.0, "speaker": "SPEAKER_00"},
        {"text": "请不吝点赞 订阅 转发 打赏支持明镜与点点栏目", "start": 9.5, "end": 22.0, "speaker": "SPEAKER_01"},
        {"text": "好的，收到。", "start": 22.4, "end": 24.0, "speaker": "SPEAKER_01"},
        {"text": "哦", "start": 24.1, "end": 24.3},
    ]
    before = copy.deepcopy(segments)
    out = sf.post_process_segments(segments, OPTS)
    assert segments == before                                   # the input is left alone
    assert out["segments"] == [
        {"text": "我们 开会 打开APP里面的二维码，对吧", "start": 3.2, "end": 9.0, "speaker": "SPEAKER_00"},
        {"text": "好的，收到", "start": 22.4, "end": 24.0, "speaker": "SPEAKER_01"},
    ]
    assert out["stats"] == {
        "input_segments": 7, "output_segments": 2, "removed_prompt_only_segments": 1,
        "removed_hallucination_segments": 1, "removed_noise_segments": 2, "cleaned_prompt_fragments": 1,
        "cleaned_hallucination_fragments": 1, "merged_segments": 1,
    }
    assert [(r["start"], r["end"], r["reason"]) for r in out["removedRanges"]] == [
        (0.0, 2.5, "prompt"), (2.5, 3.2, "noise"), (9.5, 22.0, "hallucination"), (24.1, 24.3, "noise")]
    q = out["qualityReport"]
    assert q["input_speech_sec"] == 23.0 and q["output_speech_sec"] == 7.4 and q["removed_speech_sec"] == 15.6
    assert q["coverage_ratio"] == 0.3217
    assert q["suspicious_ranges"] == [{"start": 9.5, "end": 22.0, "duration": 12.5, "reason": "hallucination"}]
    assert q["warnings"] == ["检测并移除了 1 条疑似幻觉片段", "清理后语音覆盖率偏低 (32.2%)，建议人工复核",
                             "检测到可疑丢失时间段：00:00:09-00:00:22"]


def test_clean_text_counts_patterns_not_occurrences():
    text, n_leak, n_hall = sf.clean_text("请使用简体中文。请使用简体中文！会议开始，术语参考：微信、支付宝。好", OPTS)
    assert (text, n_leak, n_hall) == ("会议开始，。好", 2, 0)     # two patterns fired; "，。" are different marks
    text, n_leak, n_hall = sf.clean_text("本期明镜与点点栏目 结束", OPTS)
    assert (text, n_leak, n_hall) == ("本期 结束", 0, 1)
    assert sf.clean_text(None, OPTS) == ("", 0, 0) and sf.clean_text(12, OPTS) == ("", 0, 0)
    assert sf.clean_text("a\n\n b\t c", OPTS)[0] == "a b c"
    assert sf.clean_text("唯一表识和文艺标识", OPTS)[0] == "唯一标识和唯一标识"


def test_whole_segment_prompt_leak_is_dropped_even_when_fragments_are_kept():
    seg = [{"text": "术语参考：微信、支付宝", "start": 1, "end": 4}]
    out = run(seg)
    assert out["segments"] == [] and out["stats"]["removed_prompt_only_segments"] == 1
    assert out["stats"]["cleaned_prompt_fragments"] == 1
    out = run(seg, strip_prompt_leak=False)                      # the whole-segment test does not look at the switch
    assert out["segments"] == [] and out["stats"]["removed_prompt_only_segments"] == 1
    assert out["stats"]["cleaned_prompt_fragments"] == 0
    assert sf.is_prompt_leak("请用简体中文！") and not sf.is_prompt_leak("请用简体中文回答")


def test_hallucination_switch():
    seg = [{"text": "明镜与点点栏目", "start": 0, "end": 3}]
    out = run(seg)
    assert out["segments"] == [] and out["stats"]["removed_hallucination_segments"] == 1
    assert out["removedRanges"][0]["reason"] == "hallucination"
    out = run(seg, strip_hallucination=False)
    assert [s["text"] for s in out["segments"]] == ["明镜与点点栏目"]
    assert sf.is_hallucination("点赞，订阅，转发，打赏支持", OPTS)   # matched on the text without punctuation
    assert not sf.is_hallucination("请大家点赞", OPTS)


def test_noise_rules():
    keep = lambda text, a, b, **kw: len(run([{"text": text, "start": a, "end": b}], **kw)["segments"]) == 1
    assert not keep("好", 0, 0.3)          # one character within 0.3 s
    assert keep("好", 0, 0.5)
    assert not keep("啊", 5, 6.2 - 1e-9)   # a filler syllable within 1.2 s
    assert keep("啊", 5, 6.5)
    assert keep("嗯嗯", 0, 1.0)            # two characters are not in the filler set
    assert keep("啊", 5, 5.5, drop_short_noise=False)
    out = run([{"text": "。。。", "start": 0, "end": 1}])           # nothing left after the edge strip: counted as prompt-only
    assert out["segments"] == [] and out["stats"]["removed_prompt_only_segments"] == 1


def test_merging_neighbours():
    segs = [{"text": "一二三", "start": 0, "end": 1}, {"text": "四五六", "start": 1.5, "end": 2},
            {"text": "七八九", "start": 2.5, "end": 3}, {"text": "十十十", "start": 3.7, "end": 4},
            {"text": "甲乙丙", "start": 4.1, "end": 5, "speaker": "A"}]
    out = run(segs)
    assert [(s["text"], s["start"], s["end"]) for s in out["segments"]] == [
        ("一二三 四五六 七八九", 0.0, 3.0), ("十十十", 3.7, 4.0), ("甲乙丙", 4.1, 5.0)]
    assert out["stats"]["merged_segments"] == 2
    out = run(segs, merge_adjacent_segments=False)
    assert len(out["segments"]) == 5 and out["stats"]["merged_segments"] == 0
    out = run(segs, max_merge_gap_sec=0.0)
    assert len(out["segments"]) == 5


def test_times_follow_javascript_number_conversion():
    out = run([{"text": "你好啊", "start": "1.5"}, {"text": "再见啦", "start": 9, "end": None}, {"text": "没有时间"},
               {"text": "坏的时间", "start": "abc", "end": float("inf")}], merge_adjacent_segments=False)
    assert [(s["start"], s["end"]) for s in out["segments"]] == [(1.5, 1.5), (9.0, 0.0), (0.0, 0.0), (0.0, 0.0)]
    assert out["qualityReport"]["input_speech_sec"] == 0 and out["qualityReport"]["coverage_ratio"] == 1


def test_ranges_fuse_within_the_gap_and_keep_their_reasons():
    fused = sf.merge_ranges([{"start": 30, "end": 41, "reason": "prompt"}, {"start": 0, "end": 5, "reason": "hallucination"},
                             {"start": 6, "end": 12, "reason": "prompt"}, {"start": 12.5, "end": 13, "reason": "hallucination"},
                             {"start": 20, "end": 19}], 1.2)
    assert [(r["start"], r["end"], r["duration"], r["reasons"]) for r in fused] == [
        (0.0, 13.0, 13.0, ["hallucination", "prompt"]), (20.0, 20.0, 0.0, ["unknown"]), (30.0, 41.0, 11.0, ["prompt"])]
    assert sf.merge_ranges([], 1.2) == []


def test_long_removed_spans_are_reported():
    segs = [{"text": "中文会议讨论记录", "start": 0, "end": 6}, {"text": "点赞订阅转发打赏支持", "start": 7, "end": 12.5},
            {"text": "正常内容在这里", "start": 13, "end": 113}]
    q = run(segs)["qualityReport"]
    assert q["suspicious_ranges"] == [{"start": 0.0, "end": 12.5, "duration": 12.5, "reason": "prompt+hallucination"}]
    assert q["coverage_ratio"] == 0.8969 and q["removed_speech_sec"] == 11.5       # 100 / 111.5
    assert q["warnings"] == ["检测并移除了 1 条疑似幻觉片段", "检测到可疑丢失时间段：00:00:00-00:00:12"]


def test_format_timestamp():
    want = {0: "00:00:00", 59.9: "00:00:59", 60: "00:01:00", 3599.99: "00:59:59", 3600: "01:00:00",
            86399: "23:59:59", 90061.5: "25:01:01"}
    assert {k: sf.format_timestamp(k) for k in want} == want


def test_markdown():
    md = sf.format_to_markdown([{"text": " 大家好 ", "start": 0, "end": 2, "speaker": "A"}, {"text": "  ", "start": 2, "end": 3},
                                {"text": "继续", "start": 3, "end": 4, "speaker": "A"}, {"text": "好", "start": 61, "end": 3661, "speaker": "B"},
                                {"text": "无名", "start": 3661, "end": 3662}])
    assert md == ("# 录音转写\n\n\n### A\n\n[00:00:00 - 00:00:02] 大家好\n\n[00:00:03 - 00:00:04] 继续\n\n"
                  "\n### B\n\n[00:01:01 - 01:01:01] 好\n\n[01:01:01 - 01:01:02] 无名\n\n")


def test_options_from_environment():
    env = {"MERGE_ADJACENT_SEGMENTS": " Off ", "STRIP_PROMPT_LEAK": "maybe", "MAX_MERGE_GAP_SEC": "0", "DROP_NOISE_MAX_SEC": "",
           "MIN_WARN_REMOVED_SPAN_SEC": "2.5", "TERM_REPLACEMENTS_JSON": '{"微心": " 微信 ", "AAP": "应用", "x": 3, " ": "y"}',
           "TERM_REPLACEMENTS": "支负宝=支付宝; bad ;=x;y=; a = b=c "}
    o = sf.FilterOptions.from_env(env)
    assert (o.merge_adjacent_segments, o.strip_prompt_leak, o.max_merge_gap_sec, o.drop_noise_max_sec,
            o.min_warn_removed_span_sec) == (False, True, 0.0, 1.2, 2.5)
    keys = list(o.term_replacements)
    assert keys[:8] == list(sf.DEFAULT_TERM_REPLACEMENTS) and keys[8:] == ["微心", "支负宝", "a"]
    assert o.term_replacements["AAP"] == "应用" and o.term_replacements["微心"] == "微信" and o.term_replacements["a"] == "b=c"
    assert sf.FilterOptions.from_env({"TERM_REPLACEMENTS_JSON": "[1]"}).term_replacements == sf.DEFAULT_TERM_REPLACEMENTS
    assert sf.FilterOptions.from_env({}) == sf.FilterOptions()


# ---------------------------------------------------------------------------------- the reference's own outputs
def _same(a, b, path=""):
    """JSON values equal; numbers to 1e-9 (both sides compute in IEEE doubles, JSON prints them shortest-round-trip)."""
    if isinstance(a, dict) and isinstance(b, dict):
        assert set(a) == set(b), (path, set(a) ^ set(b))
        for k in a:
            _same(a[k], b[k], path + "/" + str(k))
    elif isinstance(a, list) and isinstance(b, list):
        assert len(a) == len(b), (path, len(a), len(b))
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, path + "/" + str(i))
    elif isinstance(a, (int, float)) and isinstance(b, (int, float)) and not isinstance(a, bool) and not isinstance(b, bool):
        assert abs(a - b) < 1e-9, (path, a, b)
    else:
        assert a == b, (path, a, b)


def test_against_outputs_of_the_reference_javascript():
    """tests/golden/segment_filter_cases.json holds what web/api/lib/processor.js itself returned (run under node by
    tools/make_filter_golden.py) for hand-made and random segment lists under eight environments."""
    import json
    import os
    golden = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "segment_filter_cases.json"), encoding="utf-8"))
    assert len(golden["cases"]) >= 50 and len(golden["sets"]) >= 8
    n = 0
    for s in golden["sets"]:
        opts = sf.FilterOptions.from_env(s["env"])
        for case, want in zip(golden["cases"], s["outputs"]):
            got = sf.post_process_segments(copy.deepcopy(case), opts)
            _same(json.loads(json.dumps(got)), want["result"], f"env={s['env']} case={n % len(golden['cases'])}")
            assert sf.format_to_markdown(got["segments"]) == want["markdown"]
            n += 1
    assert n == len(golden["cases"]) * len(golden["sets"])
