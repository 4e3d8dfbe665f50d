"""CPU oracle (test infrastructure only; see whisper_ref.py)."""
