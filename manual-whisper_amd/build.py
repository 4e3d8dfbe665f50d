"""Build libmwhip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m manual_whisper_amd.build [--force]

hipcc cross-compiles without a GPU.  The library is never built at import or
inside a test: it is built ahead of time and travels with the tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmwhip.so")
SOURCES = ["engine.hip", "kernels_mel.hip", "kernels_gemm.hip", "kernels_enc.hip", "kernels_dec.hip", "kernels_vad.hip", "kernels_w2v.hip", "aligner.hip"]
HEADERS = [os.path.join(CSRC, "mw_kernels.h"), os.path.join(os.path.dirname(HERE), "include", "mwhip.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
# kernels_enc.hip: the encoder attention takes maxima of MFMA results; told that no NaN needs quieting, the compiler
# drops a `v_max x, x` per operand (44 of the 58 max instructions of a key tile)
EXTRA_FLAGS = {"kernels_enc.hip": ["-fno-honor-nans"]}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; cannot build libmwhip.so")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out.strip() and verbose:
            print(out)
        if p.returncode != 0:
            failed = True
            print(f"hipcc failed on {src}", file=sys.stderr)
            if not verbose:
                print(out, file=sys.stderr)
    if failed:
        raise RuntimeError("building libmwhip.so failed")
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


def build_gemm_check(force: bool = False, verbose: bool = True) -> str:
    """tools/gemm_bench: a stand-alone program that runs launch_gemm (the encoder's matrix products, every fused
    epilogue) against a naive GPU product; the GPU tests run it (tests/test_gpu_gemm.py).  Links the library's
    own kernels_gemm.o."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tools", "gemm_bench.hip")
    exe = os.path.join(root, "tools", "gemm_bench")
    obj = os.path.join(CSRC, "kernels_gemm.o")
    if force or _stale(exe, [src, obj] + HEADERS):
        hipcc = _hipcc()
        tmp = os.path.join(root, "tools", "gemm_bench.o")
        for cmd in ([hipcc, "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-I", CSRC, "-c", src, "-o", tmp],
                    [hipcc, f"--offload-arch={ARCH}", tmp, obj, "-o", exe]):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
    return exe


def build_attention_check(force: bool = False, verbose: bool = True) -> str:
    """tools/attn_bench: the encoder attention (launch_enc_attention) on its own against a naive softmax(QK^T)V, with
    operands that make the running reference exponent rise at every key tile and with per-element key counts; the GPU
    tests run it (tests/test_gpu_attention.py).  Links the library's own kernels_enc.o."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tools", "attn_bench.hip")
    exe = os.path.join(root, "tools", "attn_bench")
    obj = os.path.join(CSRC, "kernels_enc.o")
    if force or _stale(exe, [src, obj] + HEADERS):
        hipcc = _hipcc()
        tmp = os.path.join(root, "tools", "attn_bench.o")
        for cmd in ([hipcc, "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-I", CSRC, "-c", src, "-o", tmp],
                    [hipcc, f"--offload-arch={ARCH}", tmp, obj, "-o", exe]):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
    return exe


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
    print(build_gemm_check(force="--force" in sys.argv))
    print(build_attention_check(force="--force" in sys.argv))
