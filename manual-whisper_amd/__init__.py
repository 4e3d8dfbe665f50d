"""MI355X-native Whisper transcription path behind the reference's call surface.

Public surface mirrors the calls of reference transcribe.py:107-131:
``load_model``, ``load_audio``, ``Model.transcribe``, ``load_align_model`` and ``align``.  Importing this package does not load the
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


def load_align_model(*args, **kwargs):
    from .alignment import load_align_model as _load
    return _load(*args, **kwargs)


def align(*args, **kwargs):
    from .alignment import align as _align
    return _align(*args, **kwargs)


def __getattr__(name):
    if name == "EngineError":
        from .engine import EngineError
        return EngineError
    raise AttributeError(name)
