"""mw_logmel_noalloc (the entry point that allocates nothing; scratch in a file-scope __device__ array) against
mw_logmel, through the C ABI: bit-identical features on two calls in a row, the second with a smaller input, and
the encoder fed by either gives the same output."""
import numpy as np
import pytest

import manual_whisper_amd as mw
from manual_whisper_amd import synth
from manual_whisper_amd.engine import Engine, EngineError

pytestmark = pytest.mark.gpu


def test_logmel_noalloc_matches_logmel_on_two_calls_in_a_row():
    import torch
    dev = torch.device("cuda", 0)
    dims = mw.get_dims("micro")
    eng = Engine(dims, device=0, max_batch=6)
    eng.load_weights(synth.make_weights_np(dims, 1234))
    try:
        for batch, seed in ((5, 21), (2, 22)):           # the second call is smaller: stale maxima of the first must not leak
            pcm_np = synth.make_pcm_np(batch, seed=seed)
            pcm_np[0, 100000:] *= 1e-3                   # chunks with very different maxima
            if batch > 1:
                pcm_np[1] = 0.0                          # silence: the floor is the clamp value itself
            want = eng.logmel(pcm_np)
            enc_want = eng.encode(batch)
            pcm = torch.from_numpy(pcm_np).to(dev)
            out = torch.full((batch, dims.n_mels, 3000), float("nan"), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            free_before = torch.cuda.mem_get_info(dev)[0]
            eng.logmel_noalloc(pcm, out)
            assert torch.cuda.mem_get_info(dev)[0] == free_before          # nothing was allocated on the device
            got = out.cpu().numpy()
            assert np.array_equal(got, want), float(np.abs(got - want).max())
            enc_got = eng.encode(batch)                  # the features left in the context feed the encoder as well
            assert np.array_equal(enc_got, enc_want)
        with pytest.raises(EngineError, match="device memory"):
            eng.logmel_noalloc(synth.make_pcm_np(1, seed=1), out[:1])
    finally:
        eng.close()
