"""MI355X-native Whisper transcription path behind the reference's call surface.

Public surface mirrors the three calls of reference transcribe.py:107-123:
``load_model``, ``load_audio`` and ``Model.transcribe``.  Importing this package does not load the
HIP library; building an engine does, and fails loudly when the library is missing.
"""
from .dims import WhisperDims, MODELS, get_dims  # noqa: F401
from .audio import load_audio  # noqa: F401
from .postprocess import clean_segment_text, remove_hallucination_loops  # noqa: F401
from .segment_filter import FilterOptions, post_process_segments  # noqa: F401
from .emit import format_timestamp, format_transcript  # noqa: F401


def load_model(*args, **kwargs):
    from .model import load_model as _load
    return _load(*args, **kwargs)
