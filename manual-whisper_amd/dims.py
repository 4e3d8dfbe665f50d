"""Whisper architecture constants for the transcription hot path.

The reference never states these: it only passes a model *name* to
``whisperx.load_model`` (reference transcribe.py:33,107).  The numbers are the
published Whisper architecture constants; ``micro`` is a test-only shape that
keeps every code path (two heads, two layers each side, an odd vocabulary size)
while staying cheap for the CPU oracle.
"""
from dataclasses import dataclass, asdict

SAMPLE_RATE = 16000          # reference transcribe.py:118 divides by 16000
CHUNK_SECONDS = 30
CHUNK_SAMPLES = SAMPLE_RATE * CHUNK_SECONDS   # 480000
N_FFT = 400
HOP = 160
N_FRAMES = CHUNK_SAMPLES // HOP               # 3000
N_FREQ = N_FFT // 2 + 1                       # 201


@dataclass(frozen=True)
class WhisperDims:
    name: str
    n_mels: int
    d_model: int
    n_heads: int
    enc_layers: int
    dec_layers: int
    ffn_dim: int
    vocab: int
    n_audio_ctx: int = 1500
    n_text_ctx: int = 448

    @property
    def head_dim(self) -> int:
        return self.d_model // self.n_heads

    def as_dict(self):
        return asdict(self)


MODELS = {
    "micro": WhisperDims("micro", 80, 128, 2, 2, 2, 512, 1031),
    "tiny": WhisperDims("tiny", 80, 384, 6, 4, 4, 1536, 51865),
    "base": WhisperDims("base", 80, 512, 8, 6, 6, 2048, 51865),
    "small": WhisperDims("small", 80, 768, 12, 12, 12, 3072, 51865),
    "medium": WhisperDims("medium", 80, 1024, 16, 24, 24, 4096, 51865),
    "large-v2": WhisperDims("large-v2", 80, 1280, 20, 32, 32, 5120, 51865),
    "large-v3": WhisperDims("large-v3", 128, 1280, 20, 32, 32, 5120, 51866),
}


def get_dims(name: str) -> WhisperDims:
    try:
        return MODELS[name]
    except KeyError:
        raise ValueError(f"unknown Whisper model size {name!r}; known: {sorted(MODELS)}")
