"""Golden vectors for segment_filter.py from the reference's own JavaScript, run here.

    python tools/make_filter_golden.py        # needs /root/reference and node; writes tests/golden/segment_filter_cases.json

The reference's web/api/lib/processor.js is an ES module that this image's node 12 refuses for one piece of
syntax: five uses of optional chaining on the loop variable (`seg?.text`, `seg?.start`, `seg?.end`).  The file is
read as text, those five tokens are rewritten in memory to `(seg||{}).` (the same value for every input: a
missing or non-object segment yields undefined either way), and the result is written with its one import
(web/api/_localEnv.js, verbatim) to a temporary directory OUTSIDE the repository, imported by node, and deleted.
Nothing of the reference's source enters the repository: the fixture holds inputs, the environment each set ran
under, and what `postProcessSegments` / `formatToMarkdown` returned.

The module reads its switches from the environment once, at import: every environment in ENVS below is its own
node process.
"""
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

REF = "/root/reference/web/api"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "segment_filter_cases.json")

RUNNER = """
import { postProcessSegments, formatToMarkdown } from "./api/lib/processor.js";
import fs from "fs";
const cases = JSON.parse(fs.readFileSync(0, "utf8"));
const out = cases.map((segments) => {
    const result = postProcessSegments(segments);
    return { result, markdown: formatToMarkdown(result.segments) };
});
process.stdout.write(JSON.stringify(out));
"""

ENVS = [
    {},
    {"MERGE_ADJACENT_SEGMENTS": "false"},
    {"DROP_SHORT_NOISE": "0", "MAX_MERGE_GAP_SEC": "2.5"},
    {"STRIP_PROMPT_LEAK": "no", "STRIP_HALLUCINATION": "off"},
    {"DROP_NOISE_MAX_SEC": "3", "DROP_NOISE_MAX_CHARS": "4", "MIN_WARN_REMOVED_SPAN_SEC": "1", "MIN_WARN_COVERAGE_RATIO": "0.99"},
    {"TERM_REPLACEMENTS_JSON": json.dumps({"阿里爸爸": "阿里巴巴", "AAP": "应用"}, ensure_ascii=False)},
    {"TERM_REPLACEMENTS": "微心=微信, 之付宝=支付宝;AAP=APP2"},
    {"MERGE_ADJACENT_SEGMENTS": "maybe", "MAX_MERGE_GAP_SEC": "abc"},
]

FRAGMENTS = [
    "请使用简体中文。", "请用简体中文！", "术语参考：微信、支付宝、二维码", "中文会议讨论记录：", "语文会议讨论记录",
    "请不吝点赞 订阅 转发 打赏支持明镜与点点栏目", "点赞订阅转发打赏支持", "明镜与点点栏目", "字幕由Amara.org社区提供", "谢谢观看",
    "嗯", "哦。", "啊啊", "好", "对", "是的",
    "我们 我们 我们 我们 开会", "今天今天今天今天今天讨论预算", "打开AAP里面的二维码，，对吧？？", "唯一表识和文艺标识", "微心和之付宝都可以",
    "好的，收到。", "下周一上线。", "这个需求  需要\n\n再确认一下", "预算大概三十万左右", "阿里爸爸的接口文档", "OK, let's move on.", "  ", "",
    "第一点，第一点，第一点，第一点，性能", "哈哈哈哈哈哈哈哈", "。。。", "Thank you.", "thanks for watching!",
]
SPEAKERS = [None, "SPEAKER_00", "SPEAKER_01", ""]


def random_case(rng: random.Random):
    t = 0.0
    segs = []
    for _ in range(rng.randint(0, 9)):
        t += rng.choice([0.0, 0.1, 0.3, 0.59, 0.6, 0.61, 1.0, 2.4, 7.0])
        dur = rng.choice([0.2, 0.8, 1.2, 1.21, 2.0, 4.5, 11.0, 14.0])
        text = "".join(rng.sample(FRAGMENTS, rng.randint(1, 3)))
        seg = {"text": text, "start": round(t, 2), "end": round(t + dur, 2)}
        who = rng.choice(SPEAKERS)
        if who is not None:
            seg["speaker"] = who
        kind = rng.random()
        if kind < 0.04:
            seg["text"] = None
        elif kind < 0.08:
            seg["text"] = 12
        elif kind < 0.12:
            seg["start"] = "abc"
        elif kind < 0.16:
            seg["end"] = None
        elif kind < 0.20:
            seg["end"] = seg["start"] - 1.0 if isinstance(seg["start"], float) else 0
        elif kind < 0.24:
            seg["start"], seg["end"] = str(seg["start"]), str(seg["end"])
        elif kind < 0.27:
            del seg["end"]
        elif kind < 0.29:
            seg["words"] = [{"word": "x", "start": 0}]
        segs.append(seg)
        t += dur
    if rng.random() < 0.05:
        segs.insert(rng.randint(0, len(segs)), None)
    if rng.random() < 0.05:
        segs.insert(rng.randint(0, len(segs)), "oops")
    return segs


HAND_CASES = [
    [],
    [{"text": "  请使用简体中文。 ", "start": 0, "end": 2.5}, {"text": "嗯。", "start": 2.5, "end": 3.2},
     {"text": "我们 我们 我们 我们 开会", "start": 3.2, "end": 6.0, "speaker": "SPEAKER_00"},
     {"text": "打开AAP里面的二维码，，对吧？？", "start": 6.3, "end": 9.0, "speaker": "SPEAKER_00"},
     {"text": "请不吝点赞 订阅 转发 打赏支持明镜与点点栏目", "start": 9.5, "end": 22.0, "speaker": "SPEAKER_01"},
     {"text": "好的，收到。", "start": 22.4, "end": 24.0, "speaker": "SPEAKER_01"}, {"text": "哦", "start": 24.1, "end": 24.3}],
    [{"text": "请使用简体中文。请使用简体中文！会议开始，术语参考：微信、支付宝。好", "start": 1, "end": 5}],
    [{"text": "本期明镜与点点栏目 结束", "start": 0, "end": 3}, {"text": "a\n\n b\t c", "start": 3, "end": 6}],
    [{"text": "唯一表识和文艺标识", "start": 0, "end": 2}, {"text": "唯一表识和文艺标识", "start": 2.6, "end": 4, "speaker": "A"},
     {"text": "继续", "start": 4.5, "end": 6, "speaker": "B"}],
    [{"text": "第一段", "start": 0, "end": 10}, {"text": "请不吝点赞 订阅 转发 打赏支持", "start": 10, "end": 25},
     {"text": "谢谢观看", "start": 25.2, "end": 40}, {"text": "第二段", "start": 41, "end": 50}],
]


def main():
    if not os.path.isdir(REF):
        sys.exit("needs the reference tree at " + REF)
    rng = random.Random(20240607)
    cases = HAND_CASES + [random_case(rng) for _ in range(54)]
    tmp = tempfile.mkdtemp(prefix="mw_filter_golden_")
    try:
        os.makedirs(os.path.join(tmp, "api", "lib"))
        src = open(os.path.join(REF, "lib", "processor.js"), encoding="utf-8").read()
        assert src.count("seg?.") == 5 and "?." not in src.replace("seg?.", "") and "??" not in src, "the reference changed: re-read it"
        open(os.path.join(tmp, "api", "lib", "processor.js"), "w", encoding="utf-8").write(src.replace("seg?.", "(seg||{})."))
        env_src = open(os.path.join(REF, "_localEnv.js"), encoding="utf-8").read()
        open(os.path.join(tmp, "api", "_localEnv.js"), "w", encoding="utf-8").write(env_src)
        open(os.path.join(tmp, "package.json"), "w").write('{"type": "module"}')
        open(os.path.join(tmp, "runner.mjs"), "w").write(RUNNER)
        sets = []
        for env in ENVS:
            full = {k: v for k, v in os.environ.items() if k in ("PATH", "HOME")}
            full.update(env)
            r = subprocess.run(["node", "runner.mjs"], cwd=tmp, env=full, input=json.dumps(cases, ensure_ascii=False), capture_output=True,
                               text=True, timeout=120)
            if r.returncode != 0:
                sys.exit("node failed: " + r.stderr[-2000:])
            outs = json.loads(r.stdout)
            sets.append({"env": env, "outputs": outs})
        with open(OUT, "w", encoding="utf-8") as f:
            json.dump({"generator": "tools/make_filter_golden.py", "node": subprocess.run(["node", "--version"], capture_output=True, text=True).stdout.strip(),
                       "cases": cases, "sets": sets}, f, ensure_ascii=False, separators=(",", ":"))
        print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(cases), "cases x", len(ENVS), "environments")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
