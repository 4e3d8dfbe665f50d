"""Command line front end: same arguments, files and prints as reference transcribe.py:229-251.

    python transcribe.py <audio>        ->  <stem>_transcript.md + <stem>_transcript.json in the CWD

Environment (reference transcribe.py:30-36 plus this engine's own knobs):
  BATCH_SIZE     chunks per engine call (default 16 here; the reference's CPU default is 4)
  WHISPER_MODEL  model size / local checkpoint directory / synthetic:<size> (default large-v3)
  MW_MODEL_DIR   where model-size names are looked up
  MW_ARRANGE     grouped (default): encoder halves of BATCH_SIZE chunks on MW_ENCODERS engines (default 2), greedy decoding
                 of up to MW_GROUP batches at once (default 7) on MW_DECODERS engines (default 3), sized to the recording;
                 coalesced: MW_ENGINES engines (default 3) that each run whole passes of MW_COALESCE batches (default 8)
  MW_DTYPE       float16 (default) or float8 (e4m3 Linear weights quantised at load)
  MW_SEGMENT_FILTER  1: clean up with the web service's rules instead (reference web/api/lib/processor.js:52-113:
                 filler noise dropped, neighbours merged; the JSON gains cleanup_stats and quality_report as in
                 web/api/transcribe.js:239)
  MW_ALIGN_MODEL local directory of a wav2vec2 CTC checkpoint (Hugging Face layout): when set, segment times are
                 refined by forced alignment as in reference transcribe.py:127-135 (whisperx would download one)
  HF_TOKEN       accepted for compatibility; diarization is not part of this engine
Launched under torch.distributed.run, the chunks are sharded over the ranks and rank 0 writes the files.
"""
from __future__ import annotations

import os
import sys
import time
from pathlib import Path

from . import emit, postprocess, segment_filter
from .audio import load_audio
from .dims import SAMPLE_RATE
from .model import load_model

DEVICE = os.environ.get("MW_DEVICE", "cuda")
BATCH_SIZE = int(os.environ.get("BATCH_SIZE", "16"))
COMPUTE_TYPE = os.environ.get("MW_DTYPE", "float16")
MODEL_SIZE = os.environ.get("WHISPER_MODEL", "large-v3")
COALESCE = max(1, int(os.environ.get("MW_COALESCE", "8")))
ARRANGE = os.environ.get("MW_ARRANGE", "grouped")
HF_TOKEN = os.environ.get("HF_TOKEN")

# reference transcribe.py:39-46
DOMAIN_TERMS = ["微信", "支付宝", "二维码", "收款码", "小程序", "公众号", "NFC", "Node ID", "UID", "UIA", "ADNA", "APP", "H5"]
INITIAL_PROMPT = f"中文会议讨论记录，保持原意与术语准确。术语参考：{'、'.join(DOMAIN_TERMS)}。"
VAD_OPTIONS = {"vad_onset": 0.5, "vad_offset": 0.363}


def _dist_env():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), world


def transcribe_audio(audio_file, hf_token=None):
    """Load, transcribe, clean.  Returns (result dict, audio duration in seconds)."""
    rank, local_rank, world = _dist_env()
    say = print if rank == 0 else (lambda *a, **k: None)
    say(f"🎙️ 正在转写: {audio_file}")
    say(f"📊 配置: 模型={MODEL_SIZE} | 设备={DEVICE} | 精度={COMPUTE_TYPE} | 批大小={BATCH_SIZE}")
    t = time.time()
    say(f"📝 加载 Whisper {MODEL_SIZE} 模型...")
    model = load_model(MODEL_SIZE, DEVICE, device_index=local_rank, compute_type=COMPUTE_TYPE, language="zh",
                       asr_options={"initial_prompt": INITIAL_PROMPT}, vad_options=VAD_OPTIONS,
                       max_batch=max(1, BATCH_SIZE) if ARRANGE == "grouped" else min(128, max(1, BATCH_SIZE) * COALESCE),
                       n_engines=int(os.environ.get("MW_ENGINES", "3")), coalesce=COALESCE, arrangement=ARRANGE,
                       group=int(os.environ.get("MW_GROUP", "7")), n_encoders=int(os.environ.get("MW_ENCODERS", "2")),
                       n_decoders=int(os.environ.get("MW_DECODERS", "3")))
    say(f"   模型加载耗时: {time.time() - t:.1f}s")
    say("🔊 加载音频...")
    audio = load_audio(audio_file)
    audio_duration = len(audio) / SAMPLE_RATE
    say(f"   音频时长: {audio_duration:.0f}s ({audio_duration/60:.1f}min)")
    t = time.time()
    say("✍️ 转写中 (可能需要几分钟)...")
    if world > 1:
        import torch
        import torch.distributed as dist
        from .parallel import broadcast_from_rank0, shard_range, transcribe_sharded
        if not dist.is_initialized():
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        # rank 0 alone searches the recording for speech windows; every rank then cuts and transcribes its own block
        windows = broadcast_from_rank0(model.window_list(audio) if rank == 0 else None)
        spans = [(round(float(w["start"]), 3), round(float(w["end"]), 3)) for w in windows]

        def mine(a, b):
            return model.cut(audio, windows[a:b])[0]

        a0, b0 = shard_range(len(windows), rank, world)
        durations = [w["end"] - w["start"] for w in windows[a0:b0]]
        ids = transcribe_sharded(lambda c: model.transcribe_ids_chunks(c, BATCH_SIZE, "zh", durations=durations), mine,
                                 model.max_new, device=torch.device("cuda", local_rank), n_items=len(windows))
        result = {"segments": [{"text": model.detokenize(i), "start": a, "end": b} for (a, b), i in zip(spans, ids)],
                  "language": "zh"}
    else:
        result = model.transcribe(audio, batch_size=BATCH_SIZE, language="zh")
    dt = time.time() - t
    say(f"   转写耗时: {dt:.1f}s (实时比: {dt/max(audio_duration, 1e-9):.3f}x, {audio_duration/max(dt, 1e-9):.0f} audio-s/s)")
    model.close()
    if os.environ.get("MW_ALIGN_MODEL"):    # reference transcribe.py:127-135
        # every rank holds the audio and the whole segment list (the ids were gathered on all of them): each aligns its
        # contiguous block of segments on its own GPU, one gather of the blocks puts the result together
        from .alignment import align, load_align_model
        from .parallel import align_sharded
        t = time.time()
        say("🎯 对齐时间戳...")
        model_a, metadata = load_align_model(language_code="zh", device=f"cuda:{local_rank}")
        aligned = align_sharded(lambda segs: align(segs, model_a, metadata, audio, DEVICE, return_char_alignments=False),
                                result["segments"])
        model_a.close()
        result = dict(aligned, language=result.get("language", "zh"))
        say(f"   对齐耗时: {time.time() - t:.1f}s")
    else:
        say("⚠️ 未设置 MW_ALIGN_MODEL，跳过时间戳对齐：片段时间为语音活动窗口（≤30 秒）的边界。说话人分离不在本引擎范围内。")
    say("🧹 清理转写文本...")
    if os.environ.get("MW_SEGMENT_FILTER", "0").strip().lower() in ("1", "true", "yes", "on"):
        done = segment_filter.post_process_segments(result.get("segments", []))
        result.update(segments=done["segments"], cleanup_stats=done["stats"], quality_report=done["qualityReport"])
        for line in done["qualityReport"]["warnings"]:
            say(f"   ⚠️ {line}")
        return result, audio_duration
    kept, changed, dropped = postprocess.clean_segments(result.get("segments", []))
    result["segments"] = kept
    if changed:
        say(f"   修复了 {changed} 处文本问题")
    if dropped:
        say(f"   删除了 {dropped} 条提示词污染片段")
    return result, audio_duration


def main(argv=None):
    argv = sys.argv if argv is None else argv
    t0 = time.time()
    audio_file = argv[1] if len(argv) > 1 else "New Recording 46.m4a"
    if not os.path.exists(audio_file):
        print(f"❌ 文件不存在: {audio_file}")
        sys.exit(1)
    output_file = Path(audio_file).stem + "_transcript.md"
    result, audio_duration = transcribe_audio(audio_file, HF_TOKEN)
    total = time.time() - t0
    if _dist_env()[0] == 0:
        emit.format_transcript(result, audio_file, output_file, audio_duration, total, model_size=MODEL_SIZE,
                               compute_type=COMPUTE_TYPE, device=DEVICE)
        print("\n🎉 转写完成!")
        print(f"⏱️ 总耗时: {total:.1f}s | 音频时长: {audio_duration:.0f}s | 实时比: {total/max(audio_duration, 1e-9):.3f}x")


if __name__ == "__main__":
    main()
