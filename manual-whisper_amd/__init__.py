"""MI355X-native Whisper transcription path behind the reference's call surface.

Public surface mirrors the three calls of reference transcribe.py:107-123:
``load_model``, ``load_audio`` and ``Model.transcribe``.
"""
from .dims import WhisperDims, MODELS, get_dims  # noqa: F401
