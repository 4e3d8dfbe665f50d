"""MI355X-native Whisper transcription path behind the reference's call surface.

Public surface mirrors the calls of reference transcribe.py:107-131:
``load_model``, ``load_audio``, ``Model.transcribe``, ``load_align_model`` and ``align``.  Importing this package does not load the
HIP library; building an engine does, and fails loudly when the library is missing.
"""
from ._lib import configure_hw_queues as _configure_hw_queues

_configure_hw_queues()      # GPU_MAX_HW_QUEUES must be in the environment before the HIP runtime starts (see _lib.py)

from .dims import WhisperDims, MODELS, get_dims  # noqa: F401,E402
from .audio import load_audio  # noqa: F401,E402
from .postprocess import clean_segment_text, remove_hallucination_loops  # noqa: F401,E402
from .segment_filter import FilterOptions, post_process_segments  # noqa: F401,E402
from .emit import format_timestamp, format_transcript  # noqa: F401,E402


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
