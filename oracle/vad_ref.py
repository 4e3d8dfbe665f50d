"""CPU restatement of the frame-energy front end (TEST INFRASTRUCTURE ONLY: imported by tests/,
never by the product).  Parity with the reference is unpinned: the reference delegates speech
detection to whisperx's neural VAD (transcribe.py:43-46,112), which does not exist offline; this
file states the arithmetic the HIP kernel `frame_energy_kernel` must reproduce."""
import numpy as np


def frame_energy_db(pcm: np.ndarray, frame: int = 400, hop: int = 160) -> np.ndarray:
    """out[f] = 10*log10(mean(x[f*hop : f*hop+frame]**2 over the samples that exist) + 1e-10), f < ceil(n/hop)."""
    x = np.asarray(pcm, np.float64).reshape(-1)
    n = x.shape[0]
    n_frames = (n + hop - 1) // hop
    sq = np.concatenate([[0.0], np.cumsum(x * x)])
    starts = np.arange(n_frames) * hop
    ends = np.minimum(starts + frame, n)
    mean = (sq[ends] - sq[starts]) / (ends - starts)
    return (10.0 * np.log10(mean.astype(np.float32).astype(np.float64) + 1e-10)).astype(np.float32)
