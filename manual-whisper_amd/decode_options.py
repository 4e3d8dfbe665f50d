"""Decoding rules as data.

The reference overrides only ``initial_prompt`` (transcribe.py:111) and leaves
every other decoding rule to whisperx's defaults; none of that code is in the
reference tree.  The rules are therefore kept as plain data handed to the
engine (prompt ids, suppressed ids, end-of-text id), never as constants inside
kernels, so a caller with a real tokenizer can pass the exact lists.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Sequence, Tuple

from .dims import WhisperDims

# Non-speech token ids of the multilingual vocabulary without the special ids
# at the tail (transformers configuration_whisper.py:34-44).
_NON_SPEECH_MULTI = (
    1, 2, 7, 8, 9, 10, 14, 25, 26, 27, 28, 29, 31, 58, 59, 60, 61, 62, 63, 90, 91, 92, 93, 359, 503, 522,
    542, 873, 893, 902, 918, 922, 931, 1350, 1853, 1982, 2460, 2627, 3246, 3253, 3268, 3536, 3846, 3961,
    4183, 4667, 6585, 6647, 7273, 9061, 9383, 10428, 10929, 11938, 12033, 12331, 12562, 13793, 14157,
    14635, 15265, 15618, 16553, 16604, 18362, 18956, 20075, 21675, 22520, 26130, 26161, 26435, 28279,
    29464, 31650, 32302, 32470, 36865, 42863, 47425, 49870, 50254,
)

# language token offsets from <|startoftranscript|> + 1 in the multilingual vocabulary
_LANG_INDEX = {"en": 0, "zh": 1, "de": 2, "es": 3, "ru": 4, "ko": 5, "fr": 6, "ja": 7}


@dataclass(frozen=True)
class DecodeOptions:
    prompt: Tuple[int, ...]                 # forced ids, fed before sampling starts
    eot: int
    suppress: Tuple[int, ...] = ()          # masked at every sampled position
    begin_suppress: Tuple[int, ...] = ()    # additionally masked at the first sampled position
    max_new: int = 224                      # sampled tokens per chunk (half of 448, as Whisper)
    suppress_eot: bool = False              # throughput runs: random weights never emit EOT on cue

    def effective_suppress(self) -> Tuple[int, ...]:
        return tuple(sorted(set(self.suppress) | ({self.eot} if self.suppress_eot else set())))


def whisper_options(dims: WhisperDims, language: str = "zh", prompt_ids: Sequence[int] = (),
                    max_new: int = 224, suppress_eot: bool = False) -> DecodeOptions:
    """Rules for the multilingual vocabulary (51865 ids; 51866 for large-v3).

    Prompt = [<|startofprev|>, prompt_ids..., <|startoftranscript|>, <|lang|>,
    <|transcribe|>, <|notimestamps|>]; the previous-text part is dropped when
    ``prompt_ids`` is empty.  large-v3 has one more language token, which moves
    every id after the language block up by one.
    """
    eot, sot = 50257, 50258
    shift = 1 if dims.vocab >= 51866 else 0
    translate, transcribe = 50358 + shift, 50359 + shift
    startoflm, startofprev = 50360 + shift, 50361 + shift
    nospeech, notimestamps = 50362 + shift, 50363 + shift
    if language not in _LANG_INDEX:
        raise ValueError(f"language {language!r} not in the built-in table {sorted(_LANG_INDEX)}")
    lang = sot + 1 + _LANG_INDEX[language]
    prompt = []
    if prompt_ids:
        keep = dims.n_text_ctx // 2 - 1
        prompt = [startofprev] + list(prompt_ids)[-keep:]
    prompt += [sot, lang, transcribe, notimestamps]
    suppress = set(_NON_SPEECH_MULTI) | {sot, translate, transcribe, startoflm, startofprev, nospeech}
    return DecodeOptions(tuple(prompt), eot, tuple(sorted(suppress)), (220, eot), max_new, suppress_eot)


def synthetic_options(dims: WhisperDims, max_new: int = 24, suppress_eot: bool = False,
                      extra_prompt: int = 0) -> DecodeOptions:
    """Rules for synthetic checkpoints of any vocabulary size.

    With the real vocabulary size the real ids are used; for the small test
    vocabulary the ids are scaled into range so that every rule (forced prompt,
    begin-suppress, suppress list, EOT) is still exercised.
    """
    if dims.vocab >= 51865:
        o = whisper_options(dims, "zh", tuple(range(1000, 1000 + extra_prompt)), max_new, suppress_eot)
        return o
    v = dims.vocab
    eot = v - 9
    prompt = tuple(range(100, 100 + extra_prompt)) + (v - 8, v - 6, v - 3, v - 1)
    suppress = tuple(sorted({1, 2, 7, 8, 9, 10, 14, 25, v - 8, v - 7, v - 4, v - 3, v - 2}))
    return DecodeOptions(prompt, eot, suppress, (220, eot), max_new, suppress_eot)
