"""Markdown / JSON transcript emitters (reference transcribe.py:64-71, :186-226)."""
from __future__ import annotations

import json
from datetime import timedelta


def format_timestamp(seconds) -> str:
    """HH:MM:SS, truncating fractions; hours keep counting past 24."""
    whole = int(timedelta(seconds=seconds).total_seconds())
    h, rem = divmod(whole, 3600)
    m, s = divmod(rem, 60)
    return f"{h:02d}:{m:02d}:{s:02d}"


def render_markdown(result: dict, audio_file, audio_duration: float, total_time: float,
                    model_size: str, compute_type: str, device: str) -> str:
    parts = [
        "# 会议录音转写\n\n",
        f"**源文件**: {audio_file}  \n",
        f"**音频时长**: {audio_duration / 60:.1f} 分钟  \n",
        f"**模型**: {model_size} | **精度**: {compute_type} | **设备**: {device}  \n",
        f"**转写总耗时**: {total_time:.0f}s\n\n",
        "---\n\n",
    ]
    last_speaker = None
    for seg in result.get("segments", []):
        text = seg.get("text", "").strip()
        if not text:
            continue
        speaker = seg.get("speaker", "")
        if speaker and speaker != last_speaker:
            parts.append(f"\n### {speaker}\n\n")
            last_speaker = speaker
        stamp = f"[{format_timestamp(seg.get('start', 0))} - {format_timestamp(seg.get('end', 0))}]"
        parts.append(f"{stamp} {text}\n\n")
    return "".join(parts)


def format_transcript(result: dict, audio_file, output_file: str, audio_duration: float, total_time: float,
                      model_size: str = "large-v3", compute_type: str = "int8", device: str = "cpu",
                      quiet: bool = False) -> str:
    """Write ``output_file`` (Markdown) and its ``.json`` twin; returns the JSON path."""
    if not quiet:
        print(f"📄 生成文档: {output_file}")
    with open(output_file, "w", encoding="utf-8") as f:
        f.write(render_markdown(result, audio_file, audio_duration, total_time, model_size, compute_type, device))
    json_file = output_file.replace(".md", ".json")
    with open(json_file, "w", encoding="utf-8") as f:
        json.dump(result, f, ensure_ascii=False, indent=2)
    if not quiet:
        print(f"✅ 转写稿: {output_file}")
        print(f"✅ JSON: {json_file}")
    return json_file
