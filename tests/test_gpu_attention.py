"""The encoder attention on its own: tools/attn_bench (built by manual_whisper_amd.build) runs launch_enc_attention
on random fp16 operands and compares a spread of query rows (the first and last rows of a batch element, random ones)
with a naive fp32 softmax(QK^T)V on the GPU.  "rising" operands make every query's scores grow by about 12 powers of
two per tile of 64 keys, so the kernel's running reference exponent is raised, and the accumulators rescaled, at
almost every tile instead of about once; "varlen" gives every batch element its own key count (the forced aligner's
packed form).  Shapes: large-v3 (1500 keys, 20 heads), ragged key counts around tile edges, a single short tile."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "attn_bench")

CASES = [
    # B, H, T, mode
    (4, 20, 1500, "plain"),
    (4, 20, 1500, "rising"),
    (5, 16, 1499, "varlen"),
    (3, 7, 333, "rising"),       # 6 tiles, the last with 13 keys
    (2, 6, 64, "plain"),         # exactly one tile
    (2, 6, 65, "rising"),        # one key in the second tile
    (3, 4, 40, "varlen"),        # less than a tile, key counts 21..40
]


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,T,mode", CASES)
def test_encoder_attention_against_a_naive_softmax(B, H, T, mode):
    assert os.path.exists(EXE), f"{EXE} is missing: build it with `python -m manual_whisper_amd.build`"
    r = subprocess.run([EXE, str(B), str(H), str(T), mode, "0"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"check: (\d+) rows, max \|err\| ([0-9.]+), mismatches (\d+)", r.stdout)
    assert m, r.stdout
    assert int(m.group(1)) == 96 and int(m.group(3)) == 0 and float(m.group(2)) < 4e-3
