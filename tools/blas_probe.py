#!/usr/bin/env python3
"""Development aid: what the vendor BLAS behind torch.matmul reaches on the encoder's product shapes (fp16, fp32
accumulation, no epilogue), beside this build's kernels (tools/gemm_bench) ON THE SAME BOX (devices differ by 10-20 %
in the clock they hold under this load).  Operands as in tools/gemm_bench (uniform in [-1, 1) and [-0.05, 0.05)): the
clock the chip holds depends on the data.  Usage: python tools/blas_probe.py [iterations, default 400: sustained]"""
import sys
import time

import torch

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda", 0)
M = 24000
for name, N, K in (("qkv", 3840, 1280), ("fc1", 5120, 1280), ("fc2", 1280, 5120), ("out", 1280, 1280), ("cross k/v", 2560, 1280)):
    a = (torch.rand(M, K, device=dev) * 2 - 1).half()
    w = ((torch.rand(N, K, device=dev) * 2 - 1) * 0.05).half()
    for _ in range(100):
        c = a @ w.t()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        c = a @ w.t()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / iters
    print(f"{name}: M={M} N={N} K={K}: {dt * 1e6:.1f} us, {2.0 * M * N * K / dt / 1e12:.0f} TFLOP/s", flush=True)
