#!/usr/bin/env python3
"""Development aid: what the vendor BLAS behind torch.matmul reaches on the encoder's product shapes (fp16, fp32
accumulation, no epilogue), beside this build's kernels (tools/gemm_bench).  Usage: python tools/blas_probe.py"""
import time

import torch

dev = torch.device("cuda", 0)
M = 24000
for name, N, K in (("qkv", 3840, 1280), ("fc1", 5120, 1280), ("fc2", 1280, 5120), ("out", 1280, 1280), ("cross k/v", 2560, 1280)):
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16)
    for _ in range(5):
        c = a @ w.t()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(50):
        c = a @ w.t()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 50
    print(f"{name}: M={M} N={N} K={K}: {dt * 1e6:.1f} us, {2.0 * M * N * K / dt / 1e12:.0f} TFLOP/s", flush=True)
