#!/usr/bin/env python3
"""Drop-in for the reference's `python transcribe.py <audio>` on an MI355X (see manual-whisper_amd/cli.py)."""
import manual_whisper_amd  # noqa: F401  (registers the package under its importable name)
from manual_whisper_amd.cli import main

if __name__ == "__main__":
    main()
