"""Segment-list post-processing of the reference's web service: clean, classify, drop, merge, report.

Behaviour follows reference web/api/lib/processor.js:52-113 (the pass over the segments), :138-168 (text
clean-up), :185-208 (prompt-leak / hallucination / filler-noise tests), :210-232 (merging neighbours),
:234-266 (quality report), :279-299 (range merging), :301-307 (time stamps) and :331-368 (replacement table
from the environment).  It is the richer sibling of ``postprocess.clean_segments`` (the command-line script's
clean-up): besides rewriting text it drops filler noise, joins neighbours of one speaker and says how much
speech the clean-up removed.  Host code; the ids it works on come from the GPU engine unchanged.

Differences from the JavaScript that cannot matter for BMP text: lengths count code points, not UTF-16 units.
Pinned by the reference's own outputs: tests/golden/segment_filter_cases.json holds what processor.js returned,
run under node, for hand-made and random segment lists under eight environments (tools/make_filter_golden.py).
"""
from __future__ import annotations

import json
import math
import os
import re
from dataclasses import dataclass, field
from decimal import ROUND_HALF_UP, Decimal
from typing import Dict, Iterable, List, Mapping, Optional

# processor.js:16-22
_LEAK_FRAGMENTS = tuple(re.compile(p) for p in (
    r"请使用简体中文[。.!！?？]*",
    r"请用简体中文[。.!！?？]*",
    r"术语参考[:：][^。!?！？\n]*",
    r"中文会议讨论记录[:：]?",
    r"语文会议讨论记录[:：]?",
))
# processor.js:24-28
_HALLUCINATION_FRAGMENTS = tuple(re.compile(p) for p in (
    r"请不吝点赞\s*订阅\s*转发\s*打赏支持[^。!?！？\n]*",
    r"点赞\s*订阅\s*转发\s*打赏支持[^。!?！？\n]*",
    r"明镜与点点栏目",
))
# processor.js:30-35
_HALLUCINATION_WHOLE = tuple(re.compile(p) for p in (
    r"^(术语参考|中文会议讨论记录|语文会议讨论记录)",
    r"^请不吝点赞",
    r"^点赞订阅转发打赏支持",
    r"明镜与点点栏目",
))
_LEAK_WHOLE = re.compile(r"^(术语参考|中文会议讨论记录|语文会议讨论记录)")
# processor.js:37
NOISE_FILLERS = frozenset("嗯啊哦呃额哈哎")
# processor.js:39-48 (insertion order is the order of application)
DEFAULT_TERM_REPLACEMENTS = {
    "搜码二维码": "收款二维码",
    "收码二维码": "收款二维码",
    "文艺标识": "唯一标识",
    "唯一表识": "唯一标识",
    "AAP": "APP",
    "AP里面": "APP里面",
    "AP里": "APP里",
    "不信支付法": "支付宝支付法",
}

_PUNCT = "，。！？,.!?"
_RE_SPACES = re.compile(r"\s+")
_RE_LOOP = re.compile(r"(.{2,20}?)\1{3,}")
_RE_REPEAT_PUNCT = re.compile("([" + _PUNCT + r"])\1{1,}")
_RE_EDGES = re.compile("^[" + _PUNCT + r"、\s]+|[" + _PUNCT + r"、\s]+$")
_RE_COMPACT = re.compile("[" + _PUNCT + r"、\s]")


def _flag(raw, fallback: bool) -> bool:
    """processor.js:323-329."""
    if not isinstance(raw, str):
        return fallback
    v = raw.strip().lower()
    if v in ("1", "true", "yes", "on"):
        return True
    if v in ("0", "false", "no", "off"):
        return False
    return fallback


def _number(raw, fallback: float) -> float:
    """``Number(raw || fallback)`` of processor.js:10-14 for an environment string: empty and unset fall
    back, "0" is zero, anything unparsable is NaN (every comparison with it is then false)."""
    if raw is None or raw.strip() == "":
        return fallback
    try:
        return float(raw)
    except ValueError:
        return math.nan


def replacements_from_env(env: Mapping[str, str]) -> Dict[str, str]:
    """Defaults, then TERM_REPLACEMENTS_JSON (an object of strings), then TERM_REPLACEMENTS
    (``from=to;from=to``); later sources override, new keys go last (processor.js:331-368)."""
    table = dict(DEFAULT_TERM_REPLACEMENTS)
    raw = env.get("TERM_REPLACEMENTS_JSON")
    if isinstance(raw, str) and raw.strip():
        try:
            parsed = json.loads(raw)
        except ValueError:
            parsed = None
        if isinstance(parsed, dict):
            for k, v in parsed.items():
                if isinstance(k, str) and isinstance(v, str) and k.strip() and v.strip():
                    table[k.strip()] = v.strip()
    raw = env.get("TERM_REPLACEMENTS")
    if isinstance(raw, str) and raw.strip():
        for pair in (p.strip() for p in raw.split(";")):
            cut = pair.find("=")
            if cut <= 0:
                continue
            a, b = pair[:cut].strip(), pair[cut + 1:].strip()
            if a and b:
                table[a] = b
    return table


@dataclass
class FilterOptions:
    """The switches of processor.js:6-14 with their defaults."""
    strip_prompt_leak: bool = True
    strip_hallucination: bool = True
    merge_adjacent_segments: bool = True
    drop_short_noise: bool = True
    max_merge_gap_sec: float = 0.6
    drop_noise_max_sec: float = 1.2
    drop_noise_max_chars: float = 2
    min_warn_removed_span_sec: float = 10
    min_warn_coverage_ratio: float = 0.85
    term_replacements: Dict[str, str] = field(default_factory=lambda: dict(DEFAULT_TERM_REPLACEMENTS))

    @classmethod
    def from_env(cls, env: Optional[Mapping[str, str]] = None) -> "FilterOptions":
        env = os.environ if env is None else env
        return cls(
            strip_prompt_leak=_flag(env.get("STRIP_PROMPT_LEAK"), True),
            strip_hallucination=_flag(env.get("STRIP_HALLUCINATION"), True),
            merge_adjacent_segments=_flag(env.get("MERGE_ADJACENT_SEGMENTS"), True),
            drop_short_noise=_flag(env.get("DROP_SHORT_NOISE"), True),
            max_merge_gap_sec=_number(env.get("MAX_MERGE_GAP_SEC"), 0.6),
            drop_noise_max_sec=_number(env.get("DROP_NOISE_MAX_SEC"), 1.2),
            drop_noise_max_chars=_number(env.get("DROP_NOISE_MAX_CHARS"), 2),
            min_warn_removed_span_sec=_number(env.get("MIN_WARN_REMOVED_SPAN_SEC"), 10),
            min_warn_coverage_ratio=_number(env.get("MIN_WARN_COVERAGE_RATIO"), 0.85),
            term_replacements=replacements_from_env(env),
        )


_MISSING = object()


def _finite(value, fallback: float) -> float:
    """JavaScript ``Number(v)`` when finite, else the fallback (processor.js:315-318): a missing field falls
    back, ``None`` (JSON null) and "" are 0, numeric strings parse."""
    if value is _MISSING:
        return fallback
    if value is None:
        return 0.0
    if isinstance(value, bool):
        return float(value)
    if isinstance(value, (int, float)):
        return float(value) if math.isfinite(value) else fallback
    if isinstance(value, str):
        s = value.strip()
        if not s:
            return 0.0
        try:
            v = float(s)
        except ValueError:
            return fallback
        return v if math.isfinite(v) else fallback
    return fallback


def _bounds(seg) -> tuple:
    get = seg.get if isinstance(seg, Mapping) else (lambda k, d: d)
    start = _finite(get("start", _MISSING), 0.0)
    return start, _finite(get("end", _MISSING), start)


def _js_round(v: float, scale: int) -> float:
    """``Math.round(v * scale) / scale``: halves go up."""
    x = v * scale
    f = math.floor(x)
    return (f + (1 if x - f >= 0.5 else 0)) / scale


def format_timestamp(seconds: float) -> str:
    """HH:MM:SS as processor.js:301-307 builds it: hours from the seconds, minutes and seconds from the
    millisecond count truncated to an integer (what the JavaScript ``Date`` keeps)."""
    ms = math.trunc(seconds * 1000)
    return "%02d:%02d:%02d" % (math.floor(seconds / 3600), (ms // 60000) % 60, (ms // 1000) % 60)


def remove_hallucination_loops(text: str) -> str:
    """processor.js:170-174."""
    return _RE_LOOP.sub(r"\1", text) if text else ""


def compact(text: str) -> str:
    """The text without punctuation and white space: what the whole-segment tests look at."""
    return _RE_COMPACT.sub("", text)


def clean_text(raw, opts: FilterOptions) -> tuple:
    """(text, prompt fragments removed, hallucination fragments removed); counts are per pattern that
    changed the text, not per occurrence (processor.js:138-168)."""
    text = raw if isinstance(raw, str) else ""
    text = _RE_SPACES.sub(" ", text).strip()
    text = remove_hallucination_loops(text).strip()
    n_leak = n_hall = 0
    if opts.strip_prompt_leak:
        for rx in _LEAK_FRAGMENTS:
            after = rx.sub("", text)
            n_leak += after != text
            text = after
    if opts.strip_hallucination:
        for rx in _HALLUCINATION_FRAGMENTS:
            after = rx.sub("", text)
            n_hall += after != text
            text = after
    for wrong, right in opts.term_replacements.items():
        if wrong and right:
            text = text.replace(wrong, right)
    text = _RE_REPEAT_PUNCT.sub(r"\1", text)
    text = _RE_EDGES.sub("", text)
    return _RE_SPACES.sub(" ", text).strip(), n_leak, n_hall


def is_prompt_leak(text: str) -> bool:
    """processor.js:185-189 (applied whatever STRIP_PROMPT_LEAK says, as there)."""
    c = compact(text)
    return c in ("请使用简体中文", "请用简体中文") or bool(_LEAK_WHOLE.search(c))


def is_hallucination(text: str, opts: FilterOptions) -> bool:
    """processor.js:191-199."""
    if not opts.strip_hallucination:
        return False
    c = compact(text)
    if not c:
        return True
    return any(rx.search(c) or rx.search(text) for rx in _HALLUCINATION_WHOLE)


def is_noise(text: str, duration: float, opts: FilterOptions) -> bool:
    """processor.js:201-208: nothing but punctuation; a lone filler syllable in a short segment; a single
    character in 0.3 s."""
    if not opts.drop_short_noise:
        return False
    c = compact(text)
    if not c:
        return True
    if duration <= opts.drop_noise_max_sec and len(c) <= opts.drop_noise_max_chars and c in NOISE_FILLERS:
        return True
    return duration <= 0.3 and len(c) <= 1


def merge_adjacent(segments: List[dict], max_gap_sec: float) -> tuple:
    """Join a segment to its predecessor when the speaker (absent counts as a value) is the same and the
    gap is at most ``max_gap_sec``; texts are joined by a space (processor.js:210-232)."""
    if len(segments) <= 1:
        return segments, 0
    out, cur, merged = [], segments[0], 0
    for nxt in segments[1:]:
        if cur.get("speaker") == nxt.get("speaker") and nxt["start"] - cur["end"] <= max_gap_sec:
            cur["text"] += " " + nxt["text"]
            cur["end"] = nxt["end"]
            merged += 1
        else:
            out.append(cur)
            cur = nxt
    out.append(cur)
    return out, merged


def merge_ranges(ranges: Iterable[dict], max_gap_sec: float) -> List[dict]:
    """Sort by start, fuse ranges whose gap is at most ``max_gap_sec``, collect their distinct reasons in
    order of appearance (processor.js:279-299)."""
    items = []
    for r in ranges:
        s = _finite(r.get("start", _MISSING), 0.0)
        items.append({"start": s, "end": max(s, _finite(r.get("end", _MISSING), s)),
                      "reasons": [r.get("reason") or "unknown"]})
    if not items:
        return []
    items.sort(key=lambda r: r["start"])
    fused = [items[0]]
    for cur in items[1:]:
        prev = fused[-1]
        if cur["start"] - prev["end"] <= max_gap_sec:
            prev["end"] = max(prev["end"], cur["end"])
            prev["reasons"] += [x for x in cur["reasons"] if x not in prev["reasons"]]
        else:
            fused.append(cur)
    for r in fused:
        r["duration"] = max(0.0, r["end"] - r["start"])
    return fused


def _speech_seconds(segments) -> float:
    if not isinstance(segments, (list, tuple)):
        return 0.0
    total = 0.0
    for seg in segments:
        a, b = _bounds(seg)
        total += max(0.0, b - a)
    return total


def quality_report(input_segments, output_segments, removed_ranges, stats, opts: FilterOptions) -> dict:
    """How much speech time survived, which long spans were removed as prompt leak or hallucination, and
    the warnings the web page shows (processor.js:234-266)."""
    t_in, t_out = _speech_seconds(input_segments), _speech_seconds(output_segments)
    coverage = t_out / t_in if t_in > 0 else 1.0
    suspicious = [r for r in merge_ranges((r for r in removed_ranges if r["reason"] in ("hallucination", "prompt")), 1.2)
                  if r["duration"] >= opts.min_warn_removed_span_sec]
    warnings = []
    if stats["removed_hallucination_segments"] > 0:
        warnings.append("检测并移除了 %d 条疑似幻觉片段" % stats["removed_hallucination_segments"])
    if coverage < opts.min_warn_coverage_ratio:
        pct = Decimal(coverage * 100).quantize(Decimal("0.1"), rounding=ROUND_HALF_UP)
        warnings.append("清理后语音覆盖率偏低 (%s%%)，建议人工复核" % pct)
    if suspicious:
        spans = ", ".join("%s-%s" % (format_timestamp(r["start"]), format_timestamp(r["end"])) for r in suspicious[:4])
        warnings.append("检测到可疑丢失时间段：" + spans)
    return {
        "input_speech_sec": _js_round(t_in, 100),
        "output_speech_sec": _js_round(t_out, 100),
        "removed_speech_sec": _js_round(max(0.0, t_in - t_out), 100),
        "coverage_ratio": _js_round(coverage, 10000),
        "suspicious_ranges": [{"start": r["start"], "end": r["end"], "duration": _js_round(r["duration"], 100),
                               "reason": "+".join(r["reasons"])} for r in suspicious],
        "warnings": warnings,
    }


def post_process_segments(segments: List[dict], opts: Optional[FilterOptions] = None) -> dict:
    """Clean every segment's text, drop prompt leaks / hallucinations / filler noise (remembering the time
    ranges and why), merge neighbours, and report.  Returns ``{"segments", "stats", "qualityReport",
    "removedRanges"}`` with the reference's key names (processor.js:52-113); the input is not modified."""
    opts = FilterOptions.from_env() if opts is None else opts
    stats = dict(input_segments=len(segments), output_segments=0, removed_prompt_only_segments=0,
                 removed_hallucination_segments=0, removed_noise_segments=0, cleaned_prompt_fragments=0,
                 cleaned_hallucination_fragments=0, merged_segments=0)
    removed: List[dict] = []
    kept: List[dict] = []

    def drop(start, end, reason, counter):
        stats[counter] += 1
        e = max(start, end)
        removed.append({"start": start, "end": e, "duration": max(0.0, e - start), "reason": reason})

    for seg in segments:
        fields = seg if isinstance(seg, Mapping) else {}
        text, n_leak, n_hall = clean_text(fields.get("text"), opts)
        stats["cleaned_prompt_fragments"] += n_leak
        stats["cleaned_hallucination_fragments"] += n_hall
        start, end = _bounds(fields)
        if not text or is_prompt_leak(text):
            if not text and n_hall > 0:
                drop(start, end, "hallucination", "removed_hallucination_segments")
            else:
                drop(start, end, "prompt", "removed_prompt_only_segments")
        elif is_hallucination(text, opts):
            drop(start, end, "hallucination", "removed_hallucination_segments")
        elif is_noise(text, max(0.0, end - start), opts):
            drop(start, end, "noise", "removed_noise_segments")
        else:
            kept.append({**fields, "text": text, "start": start, "end": end})

    final = kept
    if opts.merge_adjacent_segments:
        final, stats["merged_segments"] = merge_adjacent(kept, opts.max_merge_gap_sec)
    stats["output_segments"] = len(final)
    return {"segments": final, "stats": stats,
            "qualityReport": quality_report(segments, final, removed, stats, opts), "removedRanges": removed}


def format_to_markdown(segments: Iterable[dict]) -> str:
    """The web service's transcript: a title, ``[start - end] text`` paragraphs, a heading whenever a
    (truthy) speaker changes (processor.js:115-134)."""
    md = "# 录音转写\n\n"
    speaker = None
    for seg in segments:
        text = seg["text"].strip()
        if not text:
            continue
        who = seg.get("speaker")
        if who and who != speaker:
            md += "\n### %s\n\n" % who
            speaker = who
        md += "[%s - %s] %s\n\n" % (format_timestamp(seg["start"]), format_timestamp(seg["end"]), text)
    return md
