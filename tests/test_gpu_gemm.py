"""The encoder's matrix products on their own: tools/gemm_bench (built by manual_whisper_amd.build) runs launch_gemm
with each fused epilogue on random fp16 / e4m3 data and compares a spread of rows (tile edges, the last rows, random
ones) with a naive product on the GPU.  The shapes are those of a large-v3 encoder layer at 16 chunks (M = 24000)
plus ragged and small cases that take the other tile forms."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "gemm_bench")

CASES = [
    # M, N, K, mode                      what it exercises
    (24000, 3840, 1280, "qkv"),        # 256x256 form: head split, V tiles computed transposed
    (24000, 1280, 1280, "resid"),      # fp32 read-modify-write
    (24000, 5120, 1280, "gelu"),       # GELU, 16-byte fp16 stores
    (24000, 1280, 5120, "resid"),      # long K
    (23999, 2560, 1280, ""),           # ragged M: clamped rows in the last tile
    (24000, 3840, 1280, "f8"),         # e4m3 operands, fp8 MFMA
    (3000, 5120, 1280, "gelu"),        # too few tiles for the 256x256 form: 128x128 form
    (1499, 1024, 4096, "resid"),       # one alignment segment's MLP-out product
    (1500, 1152, 384, "qkv"),          # tiny: d_model 384 (V tiles not transposed: 384 is no multiple of 256)
    (777, 512, 200, ""),               # K no multiple of 64: register-staged form
]


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,mode", CASES)
def test_gemm_epilogues_against_a_naive_product(M, N, K, mode):
    assert os.path.exists(EXE), f"{EXE} is missing: build it with `python -m manual_whisper_amd.build`"
    cmd = [EXE, str(M), str(N), str(K), "3"] + ([mode] if mode else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"check: (\d+) rows x (\d+) cols, max \|err\| ([0-9.]+) .* mismatches (\d+)", r.stdout)
    assert m, r.stdout
    assert int(m.group(2)) == N and int(m.group(4)) == 0 and int(m.group(1)) >= 100
