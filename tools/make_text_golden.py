#!/usr/bin/env python3
"""Generate tests/golden/text_cases.json from the reference's own host functions.

Build-container only: imports /root/reference/transcribe.py with empty stand-in
modules for ``whisperx`` and ``dotenv`` (its other imports are stdlib + torch),
calls format_timestamp / remove_hallucination_loops / clean_segment_text /
format_transcript on the inputs below and stores inputs + outputs as data.
"""
import importlib.util
import json
import os
import sys
import tempfile
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/transcribe.py"


def load_reference():
    sys.modules.setdefault("whisperx", types.ModuleType("whisperx"))
    dotenv = types.ModuleType("dotenv")
    dotenv.load_dotenv = lambda *a, **k: None
    sys.modules.setdefault("dotenv", dotenv)
    for k in ("BATCH_SIZE", "WHISPER_MODEL", "HF_TOKEN"):
        os.environ.pop(k, None)
    spec = importlib.util.spec_from_file_location("ref_transcribe", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


LOOP_CASES = [
    "那个吧" * 5, "好的" * 3, "好的" * 4, "abababab", "哈" * 6, "我们 我们 我们 我们 开会", "",
    "今天开会讨论二维码二维码二维码二维码二维码的问题", "嗯嗯嗯嗯嗯嗯嗯嗯", "12" * 10 + "34",
    "对对对对 对对对对 对对对对 对对对对", "abcabcabcabcabcabcxyz", "是的是的是的", "x" * 50,
    "支付宝支付宝支付宝支付宝。微信微信微信微信。", "谢谢大家 谢谢大家 谢谢大家 谢谢大家 谢谢大家 再见",
]
CLEAN_CASES = [
    "请使用简体中文。", "，，搜码二维码！！", "  AAP里面的AP里 。。", "", None, "请用简体中文！！好的", "文艺标识是UID",
    "不信支付法，，，收码二维码。。。", "、、开始吧？？", "AP里面有小程序", "正常的一句话。", "   ", "！？！？",
    "那个吧那个吧那个吧那个吧那个吧，，对", "请使用简体中文请使用简体中文请使用简体中文请使用简体中文", "NFC 和 Node ID,,,,",
    "AAP AAP AAP AAP", "hello... world!!!", "。开头和结尾，", "搜码二维码搜码二维码搜码二维码搜码二维码",
]
TS_CASES = [0, 0.4, 59.9, 60, 61.5, 3599.99, 3600, 86399, 86400, 90061.5, 359999.9, 12.000001]
TRANSCRIPTS = [
    {"result": {"segments": [
        {"start": 0.0, "end": 30.0, "text": " 第一段 ", "speaker": "SPEAKER_00"},
        {"start": 30.0, "end": 60.0, "text": "第二段", "speaker": "SPEAKER_00"},
        {"start": 60.0, "end": 90.0, "text": "", "speaker": "SPEAKER_01"},
        {"start": 90.0, "end": 120.5, "text": "第四段", "speaker": "SPEAKER_01"},
        {"start": 120.5, "end": 150.0, "text": "无说话人"},
        {"start": 3700.0, "end": 3725.0, "text": "回到零号", "speaker": "SPEAKER_00"}], "language": "zh"},
     "audio_file": "meeting.m4a", "audio_duration": 3725.0, "total_time": 41.6},
    {"result": {"segments": [], "language": "zh"}, "audio_file": "empty.wav", "audio_duration": 0.0, "total_time": 0.4},
    {"result": {"segments": [{"start": 0, "end": 29.99, "text": "只有一段"}]},
     "audio_file": "a b/c.wav", "audio_duration": 29.99, "total_time": 1.49},
]


def main():
    ref = load_reference()
    out = {
        "source": "reference transcribe.py functions run in the build container",
        "constants": {"DEVICE": ref.DEVICE, "BATCH_SIZE": ref.BATCH_SIZE, "COMPUTE_TYPE": ref.COMPUTE_TYPE,
                      "MODEL_SIZE": ref.MODEL_SIZE, "INITIAL_PROMPT": ref.INITIAL_PROMPT,
                      "VAD_OPTIONS": ref.VAD_OPTIONS},
        "remove_hallucination_loops": [[c, ref.remove_hallucination_loops(c)] for c in LOOP_CASES],
        "remove_hallucination_loops_max2": [[c, ref.remove_hallucination_loops(c, 2)] for c in LOOP_CASES],
        "clean_segment_text": [[c, ref.clean_segment_text(c)] for c in CLEAN_CASES],
        "format_timestamp": [[c, ref.format_timestamp(c)] for c in TS_CASES],
        "format_transcript": [],
    }
    cwd = os.getcwd()
    for case in TRANSCRIPTS:
        with tempfile.TemporaryDirectory() as td:
            os.chdir(td)
            try:
                ref.format_transcript(json.loads(json.dumps(case["result"])), case["audio_file"], "out_transcript.md",
                                      case["audio_duration"], case["total_time"])
                md = open("out_transcript.md", encoding="utf-8").read()
                js = open("out_transcript.json", encoding="utf-8").read()
            finally:
                os.chdir(cwd)
        out["format_transcript"].append({"args": case, "md": md, "json": js})
    path = os.path.join(ROOT, "tests", "golden", "text_cases.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
