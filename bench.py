#!/usr/bin/env python3
"""Throughput of the transcription hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path (log-mel -> encoder -> cross K/V -> greedy decode) over one
batch of 16 synthetic 30 s chunks per GPU, PCM already resident in HBM, token ids back on the
host.  Whisper large-v3 architecture, fp16 weights/activations with fp32 accumulation, random-init
synthetic weights (no checkpoint exists offline), 4 forced prompt ids + 128 sampled ids per chunk
with end-of-text suppressed (random weights never emit it on cue).  Chunks are independent, so
ranks share nothing on the data path; at the end the ranks' id rows are gathered once (RCCL
all_gather), as a multi-GPU transcription of one file would.  Steps are issued to --engines
engines per GPU (an engine plus forks that share its weights; own stream, caches and captured
graph each), one host thread per engine, so up to that many steps are in flight at once: a batch-16
decode step is a chain of short latency-bound launches and leaves most of the chip idle.

Rank 0 prints ONE JSON line.  `roofline` is the decode-time cross-attention kernel (the largest
HBM stream of a step), `roofline_gemv` the six weight-streaming projection launches of a decoder layer, both
timed with HIP events on the engine's stream; `fp8_b32` (N=1) is BASELINE configs[4]'s per-GPU shape; `cpu_baseline` is the numpy
oracle on this node's host cores on a bounded sample (N=1 only).
"""
import argparse
import json
import os
import sys
import time

# One hardware queue per engine stream (plus torch's own): with HIP's default of 4 (which the environment may
# also spell out) the fourth engine shares a queue and its launches serialise behind another engine's:
# 1842 audio-s/s against 2126 with 8 queues.  Must be set before the HIP runtime starts.  MW_HW_QUEUES overrides.
# The package does the same when it is imported (manual_whisper_amd._lib.configure_hw_queues: the command line front
# end and load_model get the same queues); bench.py imports torch first, so it says it here as well.
os.environ["GPU_MAX_HW_QUEUES"] = str(max(1, min(32, int(os.environ.get("MW_HW_QUEUES", "8")))))

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E, spec
MFMA_F16_PEAK_TFLOPS = 2500.0    # MI355X dense fp16 MFMA, spec (no sparsity)
MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X fp32 matrix (v_mfma_f32_32x32x2_f32: 256 FLOP/cycle/CU x 256 CUs x 2.4 GHz)
# HBM bytes per launch of the cross-attention decode kernel at batch 16 from the PMC counters
# (2 x FETCH_SIZE 60336.0 KB + WRITE_SIZE 530 KB, profiles/round3_pmc_counters_f16.csv); algorithmic bytes are 122.88 MB.
XATTN_TRAFFIC_BYTES = 2 * 60336.0 * 1024 + 530.0 * 1024
# the same at 112 sequences, the shape of a decode group (algorithmic 860.16 MB), profiles/round3_pmc_counters_f16_b112.csv
XATTN_TRAFFIC_BYTES_112 = 2 * 420413.9 * 1024 + 280.0 * 1024        # cross_attn_stream_kernel, grid 573440
# the same for cross_attn8_kernel (e4m3 cache; algorithmic 65.28 MB), profiles/round1d_pmc_counters_fp8_kv8.csv
XATTN8_TRAFFIC_BYTES = 2 * 32439.7 * 1024 + 530.0 * 1024
BATCH = 16
NEW_TOKENS = 128
MODEL = "large-v3"


def decode_bytes(dims, batch, n_prompt, n_new, layer_weight_bytes=2, kv8=False):
    """Algorithmic HBM bytes of one greedy loop over `batch` sequences: every step reads the decoder weights once,
    every sequence's cross K/V once and the self-attention cache up to the position."""
    D, F, T = dims.d_model, dims.ffn_dim, dims.n_audio_ctx
    steps = n_prompt + n_new - 1
    w_bytes = dims.dec_layers * (6 * D * D + 2 * D * F) * layer_weight_bytes + dims.vocab * D * 2
    ckv_bytes = batch * dims.dec_layers * 2 * T * (D + 4 * dims.n_heads if kv8 else D * 2)
    self_bytes = sum(batch * dims.dec_layers * 2 * (t + 1) * D * 2 for t in range(steps))
    return steps * (w_bytes + ckv_bytes) + self_bytes


def stage_rooflines(dims, batch, n_prompt, n_new, ms, layer_weight_bytes=2, kv8=False):
    """Algorithmic bytes / flops of each stage of one step over the stage's measured time (one
    engine alone on the GPU).  2*M*N*K per contraction, softmax/LayerNorm/GELU not counted."""
    D, F, T = dims.d_model, dims.ffn_dim, dims.n_audio_ctx
    mel_bytes = batch * (480000 * 4 + dims.n_mels * 3000 * 2)
    enc_flop = batch * (2 * 3000 * D * 3 * dims.n_mels + 2 * T * D * 3 * D
                        + dims.enc_layers * (2 * T * (4 * D * D + 2 * D * F) + 4 * T * T * D))
    ckv_flop = batch * 2 * T * D * 2 * D * dims.dec_layers
    dec_bytes = decode_bytes(dims, batch, n_prompt, n_new, layer_weight_bytes, kv8)

    def hbm(nbytes, t_ms):
        a = nbytes / (t_ms * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a / HBM_PEAK_GBS, 4)}

    def mfma(flop, t_ms):
        a = flop / (t_ms * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": round(a, 1), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(a / MFMA_F16_PEAK_TFLOPS, 4)}

    # log-mel: its bytes are trivial; the kernel is a dense 400 x 402 DFT + mel filter bank on the fp32 MFMA (32x32x2)
    mel_flop = batch * 3000 * (2.0 * 400 * 402 + 2.0 * 201 * dims.n_mels)
    mel_tf = mel_flop / (ms["logmel"] * 1e-3) / 1e12
    logmel = {"bound": "mfma (fp32 matrix)", "achieved": round(mel_tf, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
              "frac": round(mel_tf / MFMA_F32_PEAK_TFLOPS, 4), "hbm_GBps": hbm(mel_bytes, ms["logmel"])["achieved"]}
    out = {"logmel": logmel, "encoder": mfma(enc_flop, ms["encoder"]),
           "cross_kv": mfma(ckv_flop, ms["cross_kv"]), "decode": hbm(dec_bytes, ms["decode"])}
    floor_ms = (mel_bytes + dec_bytes) / (HBM_PEAK_GBS * 1e6) + (enc_flop + ckv_flop) / (MFMA_F16_PEAK_TFLOPS * 1e9)
    out["sum_of_stage_floors_ms"] = round(floor_ms, 2)
    out["solo_step_ms"] = round(sum(ms[k] for k in ("logmel", "encoder", "cross_kv", "decode")), 2)
    out["algorithmic_hbm_bytes_per_step"] = mel_bytes + dec_bytes
    out["algorithmic_flop_per_step"] = enc_flop + ckv_flop
    return out


def cpu_baseline(dims, weights_np, opts, gpu_ids_fn, weight_format="f16", kv_format="f16", gpu_ids_grouped_fn=None):
    """Oracle on the host: one 30 s chunk, 8 sampled tokens; decode time extrapolated to 128."""
    from threadpoolctl import threadpool_info
    from manual_whisper_amd import synth
    from oracle.whisper_ref import WhisperOracle, log_mel          # checker / baseline leg only
    sample_new = 8
    pcm = synth.make_pcm_np(1, seed=100)
    orc = WhisperOracle(dims, weights_np, weight_format, kv_format)
    t0 = time.perf_counter()
    mel = log_mel(pcm, dims.n_mels)
    t1 = time.perf_counter()
    enc = orc.encode(mel)
    t2 = time.perf_counter()
    ids, lens, _ = orc.greedy(enc, list(opts.prompt), sample_new, opts.eot, opts.effective_suppress(), opts.begin_suppress)
    t3 = time.perf_counter()
    n_prompt = len(opts.prompt)
    # greedy() = cross K/V once + (n_prompt + sample_new - 1) decoder steps; split by a second timing of cross K/V
    t4 = time.perf_counter()
    orc.cross_kv(enc)
    t_ckv = time.perf_counter() - t4
    steps_sample = n_prompt + sample_new - 1
    steps_full = n_prompt + NEW_TOKENS - 1
    t_steps = max(t3 - t2 - t_ckv, 1e-9)
    t_full = (t1 - t0) + (t2 - t1) + t_ckv + t_steps * steps_full / steps_sample
    gpu_ids = gpu_ids_fn(pcm, sample_new)
    grouped_ids = gpu_ids_grouped_fn(pcm, sample_new) if gpu_ids_grouped_fn is not None else None
    threads = max([i.get("num_threads", 1) for i in threadpool_info() if i.get("user_api") == "blas"] or [1])
    return {
        "value": round(30.0 / t_full, 4), "unit": "audio-seconds/second", "cores": int(threads), "host_cpu_count": os.cpu_count(), "kind": "port",
        "sample": (f"numpy fp32 oracle, {MODEL}, 1 chunk of 30 s: log-mel {t1 - t0:.2f}s + encoder {t2 - t1:.2f}s + "
                   f"cross-KV {t_ckv:.2f}s + {steps_sample} decode steps {t_steps:.2f}s measured; decode extrapolated "
                   f"linearly to {steps_full} steps ({NEW_TOKENS} sampled tokens)"),
        "measured_seconds": round(t3 - t0, 2),
        "ids_match_gpu": bool((gpu_ids == ids).all()),
        # the same chunk as sequence 0 of a decode group of 112 (the path the headline times: encode_into, gemm_dec_kernel,
        # cross_attn_stream_kernel), the other 111 sequences being other audio
        "ids_match_gpu_grouped": None if grouped_ids is None else bool((grouped_ids == ids).all()),
    }


def side_run(dims, opts, dev, local_rank, batch, engines, weights="f16", kv="f16", rounds=2):
    """A second measurement on the same GPU after the main one: `rounds * engines` passes of `batch` chunks over
    `engines` engines in flight; audio-seconds per second and ms per 16 chunks."""
    import torch
    from manual_whisper_amd import synth
    from manual_whisper_amd.engine import Engine
    from manual_whisper_amd.scheduler import EnginePool
    eng = Engine(dims, device=local_rank, max_batch=batch, weight_format=weights, kv_format=kv)
    eng.load_weights(synth.make_weights_torch(dims, seed=1234, device=dev))
    eng.set_decode_options(opts)
    pool = EnginePool(eng, engines)
    pcms = [synth.make_pcm_torch(batch, seed=5000 + i, device=dev) for i in range(2)]
    torch.cuda.synchronize()
    for e in pool.engines:
        e.transcribe_batch(pcms[0])
    pool.map([pcms[i % 2] for i in range(engines)])
    torch.cuda.synchronize()
    steps = rounds * engines
    t0 = time.perf_counter()
    pool.map([pcms[i % 2] for i in range(steps)])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pool.close()
    torch.cuda.empty_cache()
    return {"value": round(steps * batch * 30.0 / dt, 2), "unit": "audio-seconds/second", "passes_timed": steps,
            "ms_per_16_chunks": round(dt / (steps * batch / 16) * 1e3, 3)}


def grouped_side_run(dims, opts, dev, local_rank, batch, group, n_enc, n_dec, weights="f16", kv="f16", steps=14):
    """A second measurement in the headline's arrangement (scheduler.GroupedPool): `steps` batches of `batch` chunks,
    encoder halves on `n_enc` engines, greedy decoding of up to `group` batches at once on `n_dec` engines."""
    import torch
    from manual_whisper_amd import synth
    from manual_whisper_amd.engine import Engine
    from manual_whisper_amd.scheduler import GroupedPool
    eng = Engine(dims, device=local_rank, max_batch=batch, weight_format=weights, kv_format=kv)
    eng.load_weights(synth.make_weights_torch(dims, seed=1234, device=dev))
    eng.set_decode_options(opts)
    pool = GroupedPool(eng, batch, group, n_enc, n_dec)
    pool.set_decode_options(opts)
    pcms = [synth.make_pcm_torch(batch, seed=5000 + i, device=dev) for i in range(2)]
    torch.cuda.synchronize()
    pool.map([pcms[0]] * (group * n_dec))               # every decode engine captures its graphs once
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pool.map([pcms[i % 2] for i in range(steps)])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pool.close()
    torch.cuda.empty_cache()
    return {"value": round(steps * batch * 30.0 / dt, 2), "unit": "audio-seconds/second", "steps_timed": steps,
            "ms_per_step": round(dt / steps * 1e3, 3), "ms_per_16_chunks": round(dt / (steps * batch / 16) * 1e3, 3)}


def coalesced_run(args, dims, opts, dev, local_rank):
    """The same workload with the scheduler coalescing several batches of 16 into one engine pass: the decode
    GEMVs then stream every weight once for all of them (groups of 16 sequences inside each launch)."""
    nb, ne = args.also_batch, args.also_engines
    out = side_run(dims, opts, dev, local_rank, nb, ne, args.weights, args.kv)
    out["workload"] = (f"as config.workload, but {nb} chunks ({nb // 16} batches of 16) per engine pass, {ne} engines in flight, "
                       f"{out['passes_timed']} passes timed")
    return out


def fp8_b32_run(dims, opts, dev, local_rank):
    """BASELINE configs[4]'s per-GPU shape beside the fp16 headline: e4m3 Linear weights, 32 chunks per pass.
    The e4m3 cross-attention K/V cache is this build's own extension, not part of configs[4]: reported apart."""
    out = {"workload": "whisper-large-v3, fp8 (e4m3) Linear weights with fp8 MFMA in the encoder, batch=32x30 s chunks per step, greedy, "
                       f"{len(opts.prompt)} prompt + {opts.max_new} sampled tokens per chunk (EOT suppressed), 1 GPU; the headline's arrangement: "
                       "every step's encoder half on its own 32 chunks (2 encoder engines), greedy loop over up to 7 steps' 224 sequences "
                       "at once (2 decode engines), 14 steps timed",
           "fp8_weights": grouped_side_run(dims, opts, dev, local_rank, 32, 7, 2, 2, "fp8", "f16"),
           "fp8_weights_and_fp8_cross_kv_cache": grouped_side_run(dims, opts, dev, local_rank, 32, 7, 2, 2, "fp8", "fp8")}
    out["fp8_weights_and_fp8_cross_kv_cache"]["note"] = "the e4m3 cross K/V cache is an extension of this build, outside configs[4]"
    out["fp8_weights"]["whole_passes"] = side_run(dims, opts, dev, local_rank, 32, 3, "fp8", "f16")
    out["fp8_weights"]["whole_passes"]["workload"] = "3 engines in flight that each run whole passes of 32 chunks (no decode groups)"
    return out


def align_side_run(local_rank, n_seg=48):
    """The step after the hot path in the reference (transcribe.py:127-135): forced alignment of every segment with a
    wav2vec2-large-xlsr-53-shaped CTC model (synthetic weights, vocabulary 3503), `n_seg` windows of 30 s with 150
    characters each through alignment.align()."""
    from manual_whisper_amd import alignment, localckpt, synth
    cfg = localckpt.wav2vec2_config("w2v-large")
    model = alignment.Aligner(cfg, device=local_rank, max_seconds=30)
    model.load_weights(localckpt.make_wav2vec2_weights_np(cfg, 77))
    vocab = localckpt.synthetic_ctc_vocab(cfg["vocab_size"])
    chars = [c for c in vocab if len(c) == 1 and ord(c) >= 0x4E00]
    meta = {"language": "zh", "dictionary": vocab, "type": "huggingface"}
    audio = np.tile(synth.hash_uniform_np(30 * 16000, "wave.bench", 5) * np.float32(0.6), n_seg)
    rng = np.random.default_rng(1)
    segs = [{"text": "".join(rng.choice(chars, 150)), "start": 30.0 * i, "end": 30.0 * (i + 1)} for i in range(n_seg)]
    alignment.align(segs[:8 * model.n_streams], model, meta, audio, "cuda")
    t = time.perf_counter()
    res = alignment.align(segs, model, meta, audio, "cuda")
    dt = time.perf_counter() - t
    t = time.perf_counter()
    model.emissions(audio[:480000])
    one = time.perf_counter() - t
    model.close()
    return {"workload": f"alignment.align() over {n_seg} windows of 30 s x 150 characters, wav2vec2-large-xlsr-53 geometry (24 layers of 1024, "
                        f"vocabulary {cfg['vocab_size']}), synthetic weights, packs of up to {model.desc.max_samples // 480000} windows per pass of the model, "
                        f"{model.n_streams} passes in flight, host buffers in and out",
            "value": round(n_seg * 30 / dt, 1), "unit": "audio-seconds/second", "ms_per_window": round(dt / n_seg * 1e3, 2),
            "one_window_alone_ms": round(one * 1e3, 2), "words": len(res["word_segments"]),
            "note": "outside the headline metric: the reference runs this after model.transcribe"}


def make_model(eng, dims, opts, **kw):
    """The call surface's Model over an engine that already holds the weights, with the benchmark's decode rules."""
    from manual_whisper_amd.model import Model
    m = Model(eng, dims, "zh", arrangement="grouped", synthetic=True, max_new=opts.max_new, **kw)
    m.fixed_options = opts
    return m


def rank_share_run(eng, dims, opts, dev, batch):
    """BASELINE configs[3] as written is ONE hour of audio (120 chunks of 30 s) split over the ranks: a rank's share is
    15 / 30 / 60 / 120 chunks at N = 8 / 4 / 2 / 1.  Each share through Model.transcribe_ids_chunks (the arrangement
    Model._pool_for picks for a job of that size), engines built and graphs captured by a first call, then idle GPU ->
    ids on the host, best of two."""
    import torch
    from manual_whisper_amd import synth
    model = make_model(eng, dims, opts)
    pcm = synth.make_pcm_torch(120, seed=4242, device=dev)
    torch.cuda.synchronize()
    out = {"workload": "one rank's share of a 1 h recording (120 chunks of 30 s) at N = 8, 4, 2, 1 ranks, whole job from an idle GPU to "
                       f"ids on the host, batch_size {batch}, {len(opts.prompt)} prompt + {opts.max_new} sampled tokens, PCM resident in HBM"}
    for n in (15, 30, 60, 120):
        model.transcribe_ids_chunks(pcm[:n], batch)
        torch.cuda.synchronize()
        best = None
        for _ in range(2):
            t = time.perf_counter()
            model.transcribe_ids_chunks(pcm[:n], batch)
            dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
        key = model._pool_key
        out[f"{n}_chunks"] = {"ranks": 120 // n, "ms": round(best * 1e3, 1), "value": round(n * 30.0 / best, 1), "unit": "audio-seconds/second per GPU",
                              "whole_recording_audio_s_per_s_at_that_many_ranks": round(120 * 30.0 / best, 1),
                              "arrangement": {"batch": key[1], "batches_per_decode_group": key[2], "encoder_engines": key[3], "decode_engines": key[4]}}
    model._close_pool()
    return out


def prompt75_run(eng, dims, opts, dev, batch, args):
    """The reference always passes initial_prompt (transcribe.py:40,111), about 70 tokens, which whisperx feeds as
    <|startofprev|> prompt <|startoftranscript|> <|zh|> <|transcribe|> <|notimestamps|>: 75 forced positions before the
    first sampled token.  The headline's arrangement with such a prompt (prefilled in one pass per group)."""
    import torch
    from manual_whisper_amd import synth
    from manual_whisper_amd.decode_options import DecodeOptions
    from manual_whisper_amd.scheduler import GroupedPool
    rng = np.random.default_rng(0)
    prompt = tuple(int(t) for t in rng.integers(0, 50000, size=75 - len(opts.prompt))) + tuple(opts.prompt)
    o75 = DecodeOptions(prompt, opts.eot, opts.suppress, opts.begin_suppress, opts.max_new)
    pool = GroupedPool(eng, batch, args.group, args.encoders, args.decoders)
    pool.set_decode_options(o75)
    pcms = [synth.make_pcm_torch(batch, seed=7000 + i, device=dev) for i in range(2)]
    torch.cuda.synchronize()
    pool.map([pcms[0]] * (args.group * args.decoders))
    torch.cuda.synchronize()
    steps = 14
    t0 = time.perf_counter()
    pool.map([pcms[i % 2] for i in range(steps)])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for e in reversed(pool.engines):
        if e is not eng:
            e.close()
    eng.set_decode_options(opts)
    torch.cuda.empty_cache()
    return {"workload": f"as config.workload with a forced prompt of 75 ids per chunk (the reference's initial_prompt + task tokens), {steps} steps timed",
            "value": round(steps * batch * 30.0 / dt, 2), "unit": "audio-seconds/second", "ms_per_step": round(dt / steps * 1e3, 3)}


def beam5_run(dims, opts, dev, local_rank, weights="f16", kv="f16", n_rec=48, engines=3, passes=3):
    """whisperx decodes with a beam of 5 unless told otherwise, and the reference overrides only initial_prompt
    (transcribe.py:111): the same chunks through mw_decode_beam, `n_rec` recordings (x 5 hypotheses) per pass, on the
    engines load_model(asr_options={"beam_size": 5}) builds: forks for n_rec * 5 sequences whose cross-attention K/V
    cache holds the n_rec recordings once."""
    import torch
    from manual_whisper_amd import synth
    from manual_whisper_amd.engine import Engine
    from manual_whisper_amd.scheduler import EnginePool
    beam = 5
    eng = Engine(dims, device=local_rank, max_batch=16, weight_format=weights, kv_format=kv)
    eng.load_weights(synth.make_weights_torch(dims, seed=1234, device=dev))
    eng.set_decode_options(opts)
    pool = EnginePool(eng, engines, beam_size=beam, fork_batch=n_rec * beam, fork_cache=n_rec)
    pool.set_decode_options(opts)
    pcms = [synth.make_pcm_torch(n_rec, seed=6000 + i, device=dev) for i in range(2)]
    torch.cuda.synchronize()
    pool.map([pcms[i % 2] for i in range(engines)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pool.map([pcms[i % 2] for i in range(engines * passes)])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pool.close()
    torch.cuda.empty_cache()
    return {"workload": f"beam search, beam 5, patience 1, {n_rec} chunks x 5 hypotheses per pass, {engines} engines in flight, {engines * passes} passes timed, "
                        f"{len(opts.prompt)} prompt + {opts.max_new} sampled tokens (EOT suppressed)",
            "value": round(engines * passes * n_rec * 30.0 / dt, 2), "unit": "audio-seconds/second", "ms_per_pass": round(dt / (engines * passes) * 1e3, 2)}


def e2e_in_child(local_rank, tokens):
    """e2e_run in a process of its own, as `python transcribe.py recording` is.  In THIS process the engines of the
    measurements above have come and gone around a root engine that is still alive, and a second model built in that
    state runs its decode loops 20 % slower on this runtime (tools/e2e_probe.py with E2E_CHURN=1..7: 3.1 s against
    2.4 s for the hour; every kernel alone as fast, encoder passes as fast)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--e2e-child", str(local_rank), str(tokens)]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    for line in reversed(res.stdout.splitlines()):
        if line.startswith("{"):
            out = json.loads(line)
            out["process"] = "a fresh process (as the reference's one run per recording)"
            return out
    return {"error": "the child process printed no result", "returncode": res.returncode, "stderr_tail": res.stderr[-400:]}


def e2e_run(local_rank, minutes=60, tokens=128):
    """The reference's call shape end to end (transcribe.py:107-135): load_model with its VAD_OPTIONS, model.transcribe on
    a host buffer holding a 60-minute recording (speech windows -> chunks -> passes -> segments), then forced alignment of
    the segments.  Synthetic weights, white noise, no vocabulary: the aligned text is random characters of the alignment
    model's dictionary, the times are the segments'."""
    import manual_whisper_amd as mw
    from manual_whisper_amd import alignment, localckpt
    rng = np.random.default_rng(0)
    audio = (0.1 * rng.standard_normal(minutes * 60 * 16000)).astype(np.float32)
    t = time.perf_counter()
    model = mw.load_model("synthetic:large-v3", "cuda", device_index=local_rank, compute_type="float16", language="zh",
                          asr_options={"max_new_tokens": tokens}, vad_options={"vad_onset": 0.5, "vad_offset": 0.363})
    t_load = time.perf_counter() - t
    # first call: engines sized to this recording and their captured graphs (the reference's run pays its counterpart
    # inside load_model; reported apart).  The same recording, so that the timed call finds the arrangement it needs.
    t = time.perf_counter()
    model.transcribe(audio, batch_size=BATCH, language="zh")
    t_first = time.perf_counter() - t
    t = time.perf_counter()
    result = model.transcribe(audio, batch_size=BATCH, language="zh")
    t_tr = time.perf_counter() - t
    model.close()
    segs = result["segments"]
    cfg = localckpt.wav2vec2_config("w2v-large")
    al = alignment.Aligner(cfg, device=local_rank, max_seconds=30)
    al.load_weights(localckpt.make_wav2vec2_weights_np(cfg, 77))
    vocab = localckpt.synthetic_ctc_vocab(cfg["vocab_size"])
    chars = [c for c in vocab if len(c) == 1 and ord(c) >= 0x4E00]
    meta = {"language": "zh", "dictionary": vocab, "type": "huggingface"}
    for s_ in segs:
        s_["text"] = "".join(rng.choice(chars, max(1, int(5 * (s_["end"] - s_["start"])))))     # ~5 characters per second
    alignment.align(segs[:8 * al.n_streams], al, meta, audio, "cuda")
    t = time.perf_counter()
    res = alignment.align(segs, al, meta, audio, "cuda")
    t_al = time.perf_counter() - t
    al.close()
    dur = minutes * 60.0
    return {"workload": f"load_model('synthetic:large-v3', vad_options of the reference) -> model.transcribe({minutes} min of white noise in host memory, "
                        f"batch_size={BATCH}) -> align(segments), {tokens} sampled tokens per window (no end-of-text in synthetic weights)",
            "windows": len(segs), "load_model_s": round(t_load, 2), "first_transcribe_s": round(t_first, 3), "transcribe_s": round(t_tr, 3), "align_s": round(t_al, 3),
            "transcribe_audio_s_per_s": round(dur / t_tr, 1), "align_audio_s_per_s": round(dur / t_al, 1),
            "value": round(dur / (t_tr + t_al), 1), "unit": "audio-seconds of recording/second (transcribe + align)", "words": len(res["word_segments"]),
            "note": "white noise is all speech to the energy detector, whose cut rule halves it into ~15 s windows, each padded to a 30 s chunk: "
                    "per second of recording this does twice the chunks of the headline"}


def strong_scaling(args, eng, dims, opts, dev, cdev, dist, rank, world, rehearse):
    """--total-chunks T: BASELINE configs[3] as written.  ONE recording of T chunks (1 h = 120), rank r transcribes the
    contiguous block parallel.shard_range gives it through Model.transcribe_ids_chunks, one all_gather of (length, ids)
    rows, the wall time is the slowest rank's.  A step is one transcription of the whole recording."""
    import torch
    from manual_whisper_amd import parallel, synth
    T = args.total_chunks
    a, b = parallel.shard_range(T, rank, world)
    # chunk i of the recording is the same whatever N is (row i of one seeded construction); a rank keeps only its block
    pcm = synth.make_pcm_torch(T, seed=9000, device=dev)[a:b].contiguous()
    torch.cuda.synchronize()
    model = make_model(eng, dims, opts, group=args.group, n_encoders=args.encoders, n_decoders=args.decoders)
    per = (T + world - 1) // world

    def one_pass():
        ids = model.transcribe_ids_chunks(pcm, args.batch) if b > a else []
        rows = torch.from_numpy(parallel.pack_rows(ids, opts.max_new, per)).to(cdev)
        if world > 1:
            gathered = torch.empty((world * per, rows.shape[1]), dtype=torch.int32, device=cdev)
            dist.all_gather_into_tensor(gathered, rows)
            rows = gathered
        return rows

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):            # the first pass builds the engines and captures the graphs
        one_pass()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rows = one_pass()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    n_rows = int((rows[:, 0] >= 0).sum().item())
    key = model._pool_key
    model._close_pool()
    if rank != 0:
        return
    assert n_rows == T, (n_rows, T)
    audio_s = T * 30.0 * args.steps
    print(json.dumps({
        "metric": "audio-seconds transcribed/sec (RTF) large-v3 fp16, 30 s chunks, 1/2/4/8 GPU",
        "value": round(audio_s / elapsed, 2), "unit": "audio-seconds/second", "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup),
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f16" if args.weights == "f16" else "fp8 weights (e4m3) / f16 activations",
        "data": "synthetic" if not rehearse else "synthetic; REHEARSAL: all ranks on one GPU, gloo exchange, not a measurement",
        "rtf": round(elapsed / audio_s, 6),
        "config": {"workload": f"whisper-{args.model}, ONE recording of {T} chunks of 30 s ({T * 30 / 3600:.2f} h) per step, contiguous blocks of "
                               f"{per} chunks per rank, batch_size {args.batch}, greedy, {len(opts.prompt)} prompt + {opts.max_new} sampled tokens per chunk "
                               "(EOT suppressed), one all_gather of ids per step; a step runs from an idle GPU to the gathered ids",
                   "chunks_per_rank": per, "hw_queues": int(os.environ["GPU_MAX_HW_QUEUES"]),
                   "arrangement_per_rank": {"batch": key[1], "batches_per_decode_group": key[2], "encoder_engines": key[3], "decode_engines": key[4]},
                   "parallelism": f"dp{world} over the chunks of one recording, one all_gather of ids at the end"},
        "roofline": None, "cpu_baseline": None,
        "note": "strong-scaling side mode (--total-chunks); the default command is the weak-scaling headline and carries roofline and cpu_baseline"}), flush=True)


def free_port() -> int:
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n_gpus: int, argv) -> int:
    """`python bench.py --gpus N` outside a launcher: start N fresh rank processes (one per GPU) with
    torch.distributed.run and pass on their exit code; rank 0's JSON line reaches stdout through the inherited
    descriptors.  Called before torch or the HIP runtime are imported: this process never touches a GPU."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))


def main(argv=None):
    if argv is None and len(sys.argv) >= 4 and sys.argv[1] == "--e2e-child":
        print(json.dumps(e2e_run(int(sys.argv[2]), tokens=int(sys.argv[3]))), flush=True)
        return
    global NEW_TOKENS
    argv = sys.argv[1:] if argv is None else list(argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: whole rounds of the default arrangement (three decode groups of seven steps each = 21 steps), ~6 s timed
    ap.add_argument("--steps", type=int, default=42)
    ap.add_argument("--warmup", type=int, default=7)
    ap.add_argument("--model", default=MODEL)
    ap.add_argument("--engines", type=int, default=4, help="with --group 1: engines in flight per GPU, each running whole passes (1 = strictly sequential steps)")
    ap.add_argument("--group", type=int, default=7,
                    help="steps (batches) decoded together: the encoder half of every step runs on its own 16 chunks, the greedy loop over "
                         "up to this many steps' sequences at once (scheduler.GroupedPool); 1 = every engine runs whole passes")
    ap.add_argument("--encoders", type=int, default=2, help="with --group > 1: engines running encoder halves")
    ap.add_argument("--decoders", type=int, default=3, help="with --group > 1: engines running decode groups")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--weights", choices=("f16", "fp8"), default="f16",
                    help="fp8: BASELINE configs[4] (e4m3 Linear weights, fp8 MFMA in the encoder); the default is the metric's configs[2]")
    ap.add_argument("--kv", choices=("f16", "fp8"), default="f16", help="fp8: e4m3 cross-attention K/V cache (needs --weights fp8)")
    ap.add_argument("--also-batch", type=int, default=0,
                    help="N=1 only: after the main measurement, the same chunks with this many per engine pass (batches of 16 coalesced "
                         "by the scheduler into one lock-step decode; 0 = skip), reported under 'coalesced'")
    ap.add_argument("--also-engines", type=int, default=3)
    ap.add_argument("--no-ungrouped", action="store_true", help="N=1 only: skip the side measurement with --group 1 --engines 4")
    ap.add_argument("--no-fp8-b32", action="store_true", help="N=1 only: skip the configs[4] side measurement (fp8 weights, batch 32)")
    ap.add_argument("--no-align", action="store_true", help="N=1 only: skip the forced-alignment side measurement")
    ap.add_argument("--batch", type=int, default=16, help="30 s chunks per step per GPU (configs[4] uses 32)")
    ap.add_argument("--total-chunks", type=int, default=0,
                    help="strong scaling (BASELINE configs[3] as written): ONE recording of this many 30 s chunks (1 h = 120) per step, split "
                         "over the ranks in contiguous blocks; 0 = the weak-scaling headline (--batch chunks per step per GPU)")
    ap.add_argument("--no-rank-share", action="store_true", help="N=1 only: skip the per-rank shares of a 1 h recording (15 / 30 / 60 / 120 chunks)")
    ap.add_argument("--no-prompt75", action="store_true", help="N=1 only: skip the side measurement with a 75-id forced prompt")
    ap.add_argument("--no-beam5", action="store_true", help="N=1 only: skip the beam-5 side measurement")
    ap.add_argument("--no-e2e", action="store_true", help="N=1 only: skip the end-to-end call-surface measurement (VAD windows, transcribe, align)")
    ap.add_argument("--only-main", action="store_true", help="skip every side measurement and the CPU baseline")
    ap.add_argument("--tokens", type=int, default=NEW_TOKENS, help="sampled tokens per chunk (the metric is quoted at 128; 32 and 224 are the side cases)")
    args = ap.parse_args(argv)
    NEW_TOKENS = args.tokens
    if args.only_main:
        args.no_cpu_baseline = args.no_ungrouped = args.no_fp8_b32 = args.no_align = True
        args.no_rank_share = args.no_prompt75 = args.no_beam5 = args.no_e2e = True
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, argv))

    import torch
    import manual_whisper_amd as mw
    from manual_whisper_amd import synth
    from manual_whisper_amd.decode_options import synthetic_options
    from manual_whisper_amd.engine import Engine
    from manual_whisper_amd.scheduler import EnginePool, GroupedPool

    rank = int(os.environ.get("RANK", "0"))
    # MW_BENCH_REHEARSE=1: a rehearsal of the N > 1 path on a box with one GPU (every rank on GPU 0, the exchange over
    # gloo from host memory); the line it prints is marked and is no measurement
    rehearse = os.environ.get("MW_BENCH_REHEARSE") == "1"
    local_rank = 0 if rehearse else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    cdev = torch.device("cpu") if rehearse else dev        # where the tensors of the collectives live

    dims = mw.get_dims(args.model)
    opts = synthetic_options(dims, max_new=NEW_TOKENS, suppress_eot=True)
    weights = synth.make_weights_torch(dims, seed=1234, device=dev)
    batch = args.batch
    eng = Engine(dims, device=local_rank, max_batch=batch, weight_format=args.weights, kv_format=args.kv)
    eng.load_weights(weights)
    eng.set_decode_options(opts)
    if args.total_chunks > 0:
        del weights
        torch.cuda.empty_cache()
        strong_scaling(args, eng, dims, opts, dev, cdev, dist, rank, world, rehearse)
        eng.close()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    grouped = args.group > 1
    pool = GroupedPool(eng, batch, args.group, args.encoders, args.decoders) if grouped else EnginePool(eng, args.engines)
    if grouped:
        pool.set_decode_options(opts)
    weights_np = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        weights_np = {k: v.float().cpu().numpy() for k, v in weights.items()}
    del weights
    torch.cuda.empty_cache()

    n_iter = args.warmup + args.steps
    pcms = [synth.make_pcm_torch(batch, seed=1000 * rank + i, device=dev) for i in range(min(n_iter, 4))]
    torch.cuda.synchronize()                    # the engines read pcm on their own streams: it must be complete
    if grouped:                                 # set-up: every decode engine captures its graphs once, every encoder runs once
        pool.map([pcms[0]] * (args.group * args.decoders))
    else:
        for e in pool.engines:                  # set-up: every engine captures its decode graph once
            e.transcribe_batch(pcms[0])

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    pool.map([pcms[i % len(pcms)] for i in range(args.warmup)])
    fence()
    t0 = time.perf_counter()
    results = pool.map([pcms[(args.warmup + i) % len(pcms)] for i in range(args.steps)])
    if world > 1:       # one exchange of (length, ids) rows for everything this rank transcribed
        rows = np.concatenate([np.concatenate([lens[:, None], ids], axis=1) for ids, lens in results], axis=0)
        rows_dev = torch.from_numpy(rows).to(cdev)
        gathered = torch.empty((world * rows_dev.shape[0], rows_dev.shape[1]), dtype=torch.int32, device=cdev)     # rank-major
        dist.all_gather_into_tensor(gathered, rows_dev)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    stage = {}
    if grouped:                                 # last call of each engine (encoders: one step's encoder half; decoders: one group's loop)
        for k in ("logmel", "encoder", "cross_kv"):
            stage[k] = sum(e.stage_ms()[k] for e in pool.encoders) / len(pool.encoders)
        stage["decode"] = sum(e.stage_ms()["decode"] for e in pool.decoders) / len(pool.decoders)
        stage["decode_steps"] = pool.decoders[0].stage_ms()["decode_steps"]
    else:
        for e in pool.engines:                  # last step of each engine; engines overlap, so these do not add up to the wall time
            for k, v in e.stage_ms().items():
                stage[k] = stage.get(k, 0.0) + v / len(pool.engines)

    eng.transcribe_batch(pcms[0])               # one step with the GPU to itself: per-stage rooflines
    solo_ms = eng.stage_ms()
    group_solo = None
    if grouped:                                 # one decode group with the GPU to itself
        dec, n_seq = pool.decoders[0], batch * args.group
        for i in range(args.group):
            pool.encoders[0].encode_into(pcms[i % len(pcms)], dec, i * batch)
        dec.set_encoded(n_seq)
        # with the forms an engine takes when it has the GPU to itself (the pool selected those for engines that
        # share it: one cross-attention workgroup per CU, fewer and larger projection workgroups)
        shared = {"xattn_stream_per_cu": int(os.environ.get("MW_XSTREAM_PER_CU", "1" if args.decoders > 1 else "0")),
                  "dec_gemm_fat": int(os.environ.get("MW_DEC_GEMM_FAT", "1" if args.decoders > 1 else "0"))}
        for k in shared:
            dec.set_option(k, 0)
        dec.decode_greedy(n_seq)                # (graph capture)
        dec.decode_greedy(n_seq)
        g_ms = dec.stage_ms()["decode"]
        for k, v in shared.items():
            dec.set_option(k, v)
        g_bytes = decode_bytes(dims, n_seq, len(opts.prompt), NEW_TOKENS, 1 if args.weights == "fp8" else 2, args.kv == "fp8")
        g_rate = g_bytes / (g_ms * 1e-3) / 1e9
        group_solo = {"bound": "hbm", "sequences": n_seq, "loop_ms": round(g_ms, 2), "ms_per_16_sequences": round(g_ms * 16 / n_seq, 2),
                      "achieved": round(g_rate, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(g_rate / HBM_PEAK_GBS, 4),
                      "note": "the greedy loop of one decode group alone on the GPU, in the kernel forms of an engine that is alone (options xattn_stream_per_cu = 0, dec_gemm_fat = 0); in the timed run several groups and the encoders overlap (whole_run)"}

    # dominant kernel: decode-time cross-attention, HIP events on the engine's own stream; at the number of
    # sequences it runs on in the timed region (a decode group), and at one step's 16 beside it
    ca16_ms, ca16_bytes = eng.profile_cross_attention(batch, iters=256)
    ca_ms, ca_bytes, ca_seqs = ca16_ms, ca16_bytes, batch
    if grouped:
        ca_seqs = batch * args.group
        ca_ms, ca_bytes = pool.decoders[0].profile_cross_attention(ca_seqs, iters=128)
        # the pool holds the kernel to one resident workgroup per CU (what the timed region runs: the figure above);
        # beside it, the same launches with as many as the registers allow (the fastest form for an engine alone)
        per_cu = int(os.environ.get("MW_XSTREAM_PER_CU", "1" if args.decoders > 1 else "0"))
        pool.decoders[0].set_option("xattn_stream_per_cu", 0)
        ca_free_ms, _ = pool.decoders[0].profile_cross_attention(ca_seqs, iters=128)
        pool.decoders[0].set_option("xattn_stream_per_cu", per_cu)
    achieved = ca_bytes / (ca_ms * 1e-3) / 1e9
    # the six weight-streaming projections of a decoder layer, back to back as in a step (launch boundaries included)
    gv_ms, gv_bytes = eng.profile_decode_gemvs(batch, iters=256)
    gv_achieved = gv_bytes / (gv_ms * 1e-3) / 1e9

    if rank == 0:
        audio_s = world * batch * 30.0 * args.steps
        out = {
            "metric": "audio-seconds transcribed/sec (RTF) large-v3 fp16, 30 s chunks, 1/2/4/8 GPU",
            "value": round(audio_s / elapsed, 2), "unit": "audio-seconds/second",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if args.weights == "f16" else "fp8 weights (e4m3) / f16 activations, fp8 MFMA for the encoder's LayerNorm-fed products",
            "data": "synthetic" if not rehearse else "synthetic; REHEARSAL: all ranks on one GPU, gloo exchange, not a measurement",
            "rtf": round(elapsed / audio_s, 6),
            "config": {"workload": f"whisper-{args.model} {'fp16' if args.weights == 'f16' else 'fp8 weights'}{' + fp8 cross K/V cache' if args.kv == 'fp8' else ''}, batch={batch}x30 s chunks per GPU, greedy, "
                                   f"{len(opts.prompt)} prompt + {NEW_TOKENS} sampled tokens per chunk (EOT suppressed), "
                                   f"hipGraph decode step, " + (f"every step's log-mel / encoder / cross-K/V on its own 16 chunks ({args.encoders} encoder engines), "
                                                                f"greedy loop over the sequences of up to {args.group} steps at once ({args.decoders} decode engines), "
                                                                if grouped else f"{args.engines} engines (steps) in flight per GPU, ") +
                                   f"GPU_MAX_HW_QUEUES={os.environ['GPU_MAX_HW_QUEUES']} (what the package sets for every caller when it is imported)",
                       "hw_queues": int(os.environ["GPU_MAX_HW_QUEUES"]),
                       "parallelism": f"dp{world} over chunks, one all_gather of ids at the end"},
            "stage_ms_per_step": {k: round(v, 3) for k, v in stage.items() if k != "decode_steps"},
            "decode_steps_per_step": int(stage.get("decode_steps", 0)),
            "stage_roofline": stage_rooflines(dims, batch, len(opts.prompt), NEW_TOKENS, solo_ms, 1 if args.weights == "fp8" else 2, args.kv == "fp8"),
            "whole_run": None,
            "roofline": {"kernel": (("cross_attn_stream_kernel" if args.kv == "f16" else "cross_attn8_stream_kernel")
                                    if ca_seqs * dims.n_heads >= 1024 else
                                    ("dec_attn_kernel<256,12,4>" if args.kv == "f16" else "cross_attn8_kernel<256,12,4>"))
                                   + " (decode cross-attention, one launch per layer per token)",
                         "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "sequences": ca_seqs,
                         "traffic": ((XATTN_TRAFFIC_BYTES if args.kv == "f16" else XATTN8_TRAFFIC_BYTES) if ca_seqs == 16 else
                                     XATTN_TRAFFIC_BYTES_112 if (ca_seqs == 112 and args.kv == "f16") else None),
                         "traffic_source": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE, separate passes, profiles/"
                                           + (("round3_pmc_counters_f16_b112.csv" if ca_seqs == 112 else "round3_pmc_counters_f16.csv") if args.kv == "f16" else "round1d_pmc_counters_fp8_kv8.csv"),
                         "bytes_per_launch": ca_bytes, "avg_launch_us": round(ca_ms * 1e3, 2)},
        }
        if grouped:
            a_free = ca_bytes / (ca_free_ms * 1e-3) / 1e9
            out["roofline"]["workgroups_per_cu"] = per_cu if per_cu > 0 else "as many as fit (3)"
            out["roofline"]["unrestricted_occupancy"] = {"achieved": round(a_free, 1), "frac": round(a_free / HBM_PEAK_GBS, 4),
                                                         "avg_launch_us": round(ca_free_ms * 1e3, 2),
                                                         "note": "the same launches at three resident workgroups per CU: faster alone, slower for the run as a whole"}
            a16 = ca16_bytes / (ca16_ms * 1e-3) / 1e9
            out["roofline"]["at_16_sequences"] = {"achieved": round(a16, 1), "frac": round(a16 / HBM_PEAK_GBS, 4), "bytes_per_launch": ca16_bytes,
                                                  "avg_launch_us": round(ca16_ms * 1e3, 2)}
        out["roofline_gemv"] = {
            "kernels": "the six projection launches of a decoder layer (gemv_kernel x5, gemv_splitk_kernel), one token, back to back",
            "bound": "hbm", "achieved": round(gv_achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gv_achieved / HBM_PEAK_GBS, 4),
            "bytes_per_layer": gv_bytes, "avg_layer_us": round(gv_ms * 1e3, 2), "traffic": None}
        # the timed region as a whole (engines overlapped): algorithmic HBM bytes and MFMA flops of all steps over the wall time
        sr = out["stage_roofline"]
        step_bytes = sr["algorithmic_hbm_bytes_per_step"]
        if grouped:     # a decode group streams the decoder weights once per token for all its steps' sequences
            n_seq = batch * args.group
            step_bytes = (sr["algorithmic_hbm_bytes_per_step"] - decode_bytes(dims, batch, len(opts.prompt), NEW_TOKENS, 1 if args.weights == "fp8" else 2, args.kv == "fp8")
                          + decode_bytes(dims, n_seq, len(opts.prompt), NEW_TOKENS, 1 if args.weights == "fp8" else 2, args.kv == "fp8") / args.group)
        step_floor_ms = step_bytes / (HBM_PEAK_GBS * 1e6) + sr["algorithmic_flop_per_step"] / (MFMA_F16_PEAK_TFLOPS * 1e9)
        hb = step_bytes * world * args.steps / elapsed / 1e9
        fl = sr["algorithmic_flop_per_step"] * world * args.steps / elapsed / 1e12
        out["whole_run"] = {"hbm_GBps_per_gpu": round(hb / world, 1), "hbm_frac": round(hb / world / HBM_PEAK_GBS, 4),
                            "mfma_TFLOPs_per_gpu": round(fl / world, 1), "mfma_frac": round(fl / world / MFMA_F16_PEAK_TFLOPS, 4),
                            "algorithmic_hbm_bytes_per_step": int(step_bytes), "step_floor_ms": round(step_floor_ms, 2),
                            "step_floor_over_step": round(step_floor_ms / (elapsed / args.steps * 1e3), 4)}
        if grouped:
            out["whole_run"]["note"] = (f"bytes of a step as this arrangement moves them: decoder weights once per token per group of {args.group} steps "
                                        "(stage_roofline prices one step decoding alone, weights streamed for its 16 sequences)")
        if group_solo is not None:
            out["stage_roofline"]["decode_group"] = group_solo
        if grouped:
            out["stage_ms_note"] = "decode: one group's loop (all its steps' sequences); the other stages: one step's 16 chunks"
        # the side measurements build engines of their own: give the memory of this run's forks back first
        # (the root engine stays: the CPU baseline compares its ids)
        for e in reversed(pool.engines):
            if e is not eng:
                e.close()
        torch.cuda.empty_cache()
        # Order: first what runs on this run's root engine (pools rebuilt on the same root, the CPU baseline's id checks), then
        # the root is closed, then the measurements that load models of their own: a model whose root engine is created
        # beside an older root that has had forks decodes 20 % slower on this runtime (tools/e2e_probe.py, E2E_CHURN).
        if world == 1 and grouped and not args.no_rank_share and args.model == MODEL and args.weights == "f16":
            out["rank_share"] = rank_share_run(eng, dims, opts, dev, batch)
        if world == 1 and grouped and not args.no_prompt75 and args.model == MODEL:
            out["prompt75"] = prompt75_run(eng, dims, opts, dev, batch, args)
        if weights_np is not None:
            def gpu_ids(pcm, n_new):
                o = synthetic_options(dims, max_new=n_new, suppress_eot=True)
                eng.set_decode_options(o)
                ids, _ = eng.transcribe_batch(pcm)
                return ids
            def gpu_ids_grouped(pcm, n_new):
                o = synthetic_options(dims, max_new=n_new, suppress_eot=True)
                gp = GroupedPool(eng, batch, args.group, 1, 1)
                gp.set_decode_options(o)
                first = torch.cat([torch.from_numpy(pcm).to(dev), synth.make_pcm_torch(batch - 1, seed=8100, device=dev)])
                rest = synth.make_pcm_torch(batch, seed=8101, device=dev)
                torch.cuda.synchronize()
                res = gp.map([first] + [rest] * (args.group - 1))
                for e in reversed(gp.engines):
                    if e is not eng:
                        e.close()
                return res[0][0][:1]
            out["cpu_baseline"] = cpu_baseline(dims, weights_np, opts, gpu_ids, args.weights, args.kv, gpu_ids_grouped if grouped else None)
        eng.close()
        torch.cuda.empty_cache()
        if world == 1 and args.also_batch > batch:
            out["coalesced"] = coalesced_run(args, dims, opts, dev, local_rank)
        if world == 1 and grouped and not args.no_ungrouped:
            out["ungrouped"] = side_run(dims, opts, dev, local_rank, batch, 4, args.weights, args.kv)
            out["ungrouped"]["workload"] = "as config.workload, but 4 engines in flight that each run whole passes of 16 chunks (decode per step)"
        if world == 1 and not args.no_fp8_b32 and args.model == MODEL and args.weights == "f16":
            out["fp8_b32"] = fp8_b32_run(dims, opts, dev, local_rank)
        if world == 1 and not args.no_align and args.model == MODEL:
            out["align"] = align_side_run(local_rank)
        if world == 1 and not args.no_beam5 and args.model == MODEL and args.weights == "f16":
            out["beam5"] = beam5_run(dims, opts, dev, local_rank)
        if world == 1 and not args.no_e2e and args.model == MODEL and args.weights == "f16":
            out["e2e"] = e2e_in_child(local_rank, NEW_TOKENS)
        print(json.dumps(out), flush=True)
    pool.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
