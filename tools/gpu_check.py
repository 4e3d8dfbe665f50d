#!/usr/bin/env python3
"""Development aid: run every stage of the HIP path on one GPU and print its error against the
CPU oracle (no assertions).  Usage: python tools/gpu_check.py micro [tiny ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import manual_whisper_amd as mw                                  # noqa: E402
from manual_whisper_amd import synth                             # noqa: E402
from manual_whisper_amd.decode_options import synthetic_options  # noqa: E402
from manual_whisper_amd.engine import Engine                     # noqa: E402
from oracle.whisper_ref import WhisperOracle, log_mel            # noqa: E402


def main():
    for name in sys.argv[1:] or ["micro"]:
        dims = mw.get_dims(name)
        B = 2
        w = synth.make_weights_np(dims, 1234)
        pcm = synth.make_pcm_np(B, 0)
        opts = synthetic_options(dims, max_new=16)
        t = time.time()
        eng = Engine(dims, 0, max_batch=4)
        eng.load_weights(w)
        eng.set_decode_options(opts)
        print(f"[{name}] engine up in {time.time() - t:.2f}s", flush=True)
        orc = WhisperOracle(dims, w)

        mel_ref = log_mel(pcm, dims.n_mels)
        mel = eng.logmel(pcm)
        print(f"[{name}] mel max|err| {np.abs(mel - mel_ref).max():.3e}  (ref range {mel_ref.min():.3f}..{mel_ref.max():.3f})", flush=True)

        enc_ref = orc.encode(mel_ref)
        enc = eng.encode(B, mel=mel_ref)
        err = np.abs(enc - enc_ref)
        print(f"[{name}] enc max|err| {err.max():.3e} mean|err| {err.mean():.3e} (ref |mean| {np.abs(enc_ref).mean():.3f})", flush=True)
        enc2 = eng.encode(B)       # from the engine's own features
        print(f"[{name}] enc(from own mel) max|err| {np.abs(enc2 - enc_ref).max():.3e}", flush=True)

        toks = np.array([list(opts.prompt) + [5, 17, 33, 100]] * B, np.int32)
        toks[1, -3:] = [40, 41, 42]
        lg_ref = orc.forced_logits(enc_ref, toks)
        lg = eng.decode_forced(toks)
        err = np.abs(lg - lg_ref)
        print(f"[{name}] forced logits max|err| {err.max():.3e} mean {err.mean():.3e} (std {lg_ref.std():.3f}); per pos {err.max(axis=(0, 2))}", flush=True)

        ids_ref, lens_ref, lref = orc.greedy(enc_ref, list(opts.prompt), opts.max_new, opts.eot, opts.suppress, opts.begin_suppress, return_logits=True)
        for graph in (0, 1):
            eng.set_option("graph", graph)
            ids, lens, lgs = eng.decode_greedy(B, want_logits=True)
            same = (ids == ids_ref).all()
            print(f"[{name}] greedy graph={graph} ids equal {same} lens {lens} ref {lens_ref}", flush=True)
            if not same:
                print(ids, ids_ref, sep="\n")
            n = lref.shape[0]
            print(f"[{name}]   sampled-logits max|err| {np.abs(lgs[:n] - lref).max():.3e}", flush=True)
        ids3, lens3 = eng.transcribe_batch(pcm)
        print(f"[{name}] transcribe_batch ids equal {(ids3 == ids_ref).all()}  stages {eng.stage_ms()}", flush=True)
        eng.close()


if __name__ == "__main__":
    main()
