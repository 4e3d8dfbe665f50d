"""Segment text clean-up applied after transcription.

Behaviour follows reference transcribe.py:48-61 (tables), :74-79 (loop
collapse), :82-95 (per-segment clean-up) and :156-181 (drop / count loop);
tests/golden/text_cases.json holds the reference's own outputs.
"""
from __future__ import annotations

import re
from typing import Iterable, List, Tuple

# reference transcribe.py:48-51
PROMPT_LEAK_PATTERNS = (
    r"请使用简体中文[。.!！?？]*",
    r"请用简体中文[。.!！?？]*",
)

# reference transcribe.py:53-61 (order matters: applied first to last)
TERM_REPLACEMENTS = (
    ("搜码二维码", "收款二维码"),
    ("收码二维码", "收款二维码"),
    ("文艺标识", "唯一标识"),
    ("AAP", "APP"),
    ("AP里面", "APP里面"),
    ("AP里", "APP里"),
    ("不信支付法", "支付宝支付法"),
)

_PUNCT = "，。！？,.!?"
_RE_REPEAT_PUNCT = re.compile("([" + _PUNCT + r"])\1+")
_EDGE = "[" + _PUNCT + r"、\s]+"
_RE_EDGES = re.compile("^" + _EDGE + "|" + _EDGE + "$")
_LEAKS = tuple(re.compile(p) for p in PROMPT_LEAK_PATTERNS)


def remove_hallucination_loops(text: str, max_repeat: int = 3) -> str:
    """Collapse a 2-20 character unit followed by >= max_repeat copies of itself to one copy."""
    return re.sub(r"(.{2,20}?)\1{%d,}" % max_repeat, r"\1", text)


def clean_segment_text(text) -> str:
    s = (text or "").strip()
    s = remove_hallucination_loops(s)
    for rx in _LEAKS:
        s = rx.sub("", s)
    for wrong, right in TERM_REPLACEMENTS:
        s = s.replace(wrong, right)
    s = _RE_REPEAT_PUNCT.sub(r"\1", s)
    return _RE_EDGES.sub("", s).strip()


def clean_segments(segments: Iterable[dict]) -> Tuple[List[dict], int, int]:
    """Clean every segment in place; drop those that become empty.

    Returns (kept segments, number whose text changed, number dropped).
    """
    kept, changed, dropped = [], 0, 0
    for seg in segments:
        before = seg.get("text", "")
        after = clean_segment_text(before)
        if not after:
            dropped += 1
            continue
        if after != before:
            changed += 1
        seg["text"] = after
        kept.append(seg)
    return kept, changed, dropped
