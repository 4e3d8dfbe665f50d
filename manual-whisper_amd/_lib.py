"""ctypes binding of libmwhip.so (declared in include/mwhip.h).

There is no CPU fallback: if the library is missing or cannot be loaded the
import error says how to build it.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MW_LIB_PATH") or os.path.join(_HERE, "libmwhip.so")      # MW_LIB_PATH: A/B runs of two builds

# every symbol include/mwhip.h declares
SYMBOLS = (
    "mw_create", "mw_fork", "mw_fork_sized", "mw_fork_cached", "mw_set_weight", "mw_finalize", "mw_set_decode_opts", "mw_destroy", "mw_last_error",
    "mw_logmel", "mw_logmel_noalloc", "mw_encode", "mw_encode_into", "mw_set_encoded", "mw_read_cross_kv", "mw_decode_greedy", "mw_decode_beam", "mw_decode_sample", "mw_sample_uniform", "mw_decode_forced", "mw_transcribe_batch",
    "mw_frame_energy_db", "mw_stage_ms", "mw_profile_cross_attention", "mw_profile_decode_gemvs", "mw_set_option", "mw_synchronize",
    "mw_align_frames", "mw_align_create", "mw_align_destroy", "mw_align_last_error", "mw_align_set_weight", "mw_align_finalize",
    "mw_align_emissions", "mw_align_fork", "mw_align_path", "mw_align_segment", "mw_align_segments",
)


class ModelDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_mels", "d_model", "n_heads", "enc_layers", "dec_layers",
                                          "ffn_dim", "vocab", "n_audio_ctx", "n_text_ctx", "weight_format", "kv_format")]


WEIGHT_FORMATS = {"f16": 0, "fp8": 1}        # mw_weight_format
KV_FORMATS = {"f16": 0, "fp8": 1}            # mw_kv_format


class DecodeOpts(C.Structure):
    _fields_ = [("prompt", C.POINTER(C.c_int32)), ("n_prompt", C.c_int32),
                ("suppress", C.POINTER(C.c_int32)), ("n_suppress", C.c_int32),
                ("begin_suppress", C.POINTER(C.c_int32)), ("n_begin_suppress", C.c_int32),
                ("eot", C.c_int32), ("max_new", C.c_int32)]


class AlignDesc(C.Structure):
    _fields_ = [("n_conv", C.c_int32), ("conv_dim", C.c_int32), ("conv_kernel", C.c_int32 * 8), ("conv_stride", C.c_int32 * 8),
                ("hidden", C.c_int32), ("n_heads", C.c_int32), ("n_layers", C.c_int32), ("ffn_dim", C.c_int32),
                ("pos_kernel", C.c_int32), ("pos_groups", C.c_int32), ("vocab", C.c_int32), ("max_samples", C.c_int32)]


_lib = None

# Hardware queues.  Every engine stream wants a hardware queue of its own: the default arrangement opens 2 encoder
# and 3 decode engine streams beside the root engine's and torch's, and with HIP's default of 4 queues per process
# (GPU_MAX_HW_QUEUES, which the environment may also spell out) streams share queues and their launches serialise
# behind each other's (2057 -> 1842 audio-s/s measured with a fourth engine on a shared queue).  The HIP runtime
# reads the variable once, when it starts, so it has to be in the environment before anything makes a HIP call.
DEFAULT_HW_QUEUES = 8
MAX_HW_QUEUES = 32
_hw_queues = {"effective": None, "late": False}


def _hip_runtime_mapped() -> bool:
    try:
        with open("/proc/self/maps", encoding="ascii", errors="replace") as f:
            return any("libamdhip64" in line for line in f)
    except OSError:
        return False


def configure_hw_queues() -> int:
    """Put GPU_MAX_HW_QUEUES into the environment (MW_HW_QUEUES, default 8, at most 32) unless a HIP runtime is
    already mapped into the process, in which case what it started with stands.  Called when the package is
    imported and again before the library is loaded; returns the number of queues the process will have."""
    if _hw_queues["effective"] is not None:
        return _hw_queues["effective"]
    try:
        want = int(os.environ.get("MW_HW_QUEUES", DEFAULT_HW_QUEUES))
    except ValueError:
        want = DEFAULT_HW_QUEUES
    want = max(1, min(MAX_HW_QUEUES, want))
    if _hip_runtime_mapped():
        # someone (torch, most likely) loaded the runtime first.  If it has not made its first HIP call yet the
        # variable still takes effect, so it is raised all the same; what is counted on is the value it had
        try:
            have = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
        except ValueError:
            have = 4
        if have < want:
            os.environ["GPU_MAX_HW_QUEUES"] = str(want)
        _hw_queues.update(effective=have, late=True)
        return have
    os.environ["GPU_MAX_HW_QUEUES"] = str(want)
    _hw_queues.update(effective=want, late=False)
    return want


def check_hw_queues(n_streams: int) -> None:
    """Warn when an arrangement opens more engine streams (+ torch's own) than the process has hardware queues."""
    have = configure_hw_queues()
    if n_streams + 1 > have:
        import warnings
        why = ("the HIP runtime was loaded before manual_whisper_amd was imported, so GPU_MAX_HW_QUEUES may not have been raised "
               "in time: import manual_whisper_amd (or set GPU_MAX_HW_QUEUES) before torch") if _hw_queues["late"] else \
              "raise MW_HW_QUEUES (at most 32)"
        warnings.warn(f"{n_streams} engine streams on {have} hardware queues: streams share queues and their launches "
                      f"serialise; {why}", RuntimeWarning, stacklevel=3)


def _share_torch_hip_runtime():
    """One HIP runtime per process.

    PyTorch-ROCm wheels bundle their own libamdhip64.so (soname libamdhip64.so.7, the same soname
    libmwhip.so asks for).  Two HIP/HSA runtimes in one process cannot both own the GPU: whichever
    initialises second sees no devices.  So when torch is installed its bundled runtime is loaded
    first, by path; libmwhip.so then binds to it through the soname, and a later ``import torch``
    finds the very same file already mapped.  Without torch the system ROCm runtime is used.
    """
    import importlib.util
    if os.environ.get("MW_SYSTEM_HIP") == "1":     # torch-free process (e.g. under a counter profiler): system runtime
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m manual_whisper_amd.build` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    configure_hw_queues()           # before any HIP runtime is mapped by this package
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    vp, i32p, f32p = C.c_void_p, C.POINTER(C.c_int32), C.c_void_p
    lib.mw_create.argtypes = [C.POINTER(ModelDesc), C.c_int, C.c_int, C.POINTER(vp)]
    lib.mw_fork.argtypes = [vp, C.POINTER(vp)]
    lib.mw_fork_sized.argtypes = [vp, C.c_int, C.POINTER(vp)]
    lib.mw_fork_cached.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
    lib.mw_encode_into.argtypes = [vp, f32p, C.c_int, vp, C.c_int]
    lib.mw_set_encoded.argtypes = [vp, C.c_int]
    lib.mw_read_cross_kv.argtypes = [vp, C.c_int, C.c_int, f32p, f32p]
    lib.mw_set_weight.argtypes = [vp, C.c_char_p, vp, C.POINTER(C.c_int64), C.c_int, C.c_int]
    lib.mw_finalize.argtypes = [vp]
    lib.mw_set_decode_opts.argtypes = [vp, C.POINTER(DecodeOpts)]
    lib.mw_destroy.argtypes = [vp]
    lib.mw_destroy.restype = None
    lib.mw_last_error.argtypes = [vp]
    lib.mw_last_error.restype = C.c_char_p
    lib.mw_logmel.argtypes = [vp, f32p, C.c_int, f32p]
    lib.mw_logmel_noalloc.argtypes = [vp, f32p, C.c_int, f32p]
    lib.mw_encode.argtypes = [vp, f32p, C.c_int, f32p]
    lib.mw_decode_greedy.argtypes = [vp, C.c_int, vp, vp, f32p]
    lib.mw_decode_beam.argtypes = [vp, C.c_int, C.c_int, C.c_float, vp, vp, f32p, vp, vp, f32p]
    lib.mw_decode_sample.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.c_uint32, vp, vp, vp, f32p, vp, vp, f32p, f32p]
    lib.mw_sample_uniform.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.mw_sample_uniform.restype = C.c_float
    lib.mw_decode_forced.argtypes = [vp, C.c_int, vp, C.c_int, f32p]
    lib.mw_transcribe_batch.argtypes = [vp, f32p, C.c_int, vp, vp]
    lib.mw_frame_energy_db.argtypes = [vp, f32p, C.c_int64, C.c_int, C.c_int, f32p, C.c_int64]
    lib.mw_stage_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.mw_profile_cross_attention.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double)]
    lib.mw_profile_decode_gemvs.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double)]
    lib.mw_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    lib.mw_synchronize.argtypes = [vp]
    lib.mw_align_frames.argtypes = [C.POINTER(AlignDesc), C.c_int64]
    lib.mw_align_frames.restype = C.c_int64
    lib.mw_align_create.argtypes = [C.POINTER(AlignDesc), C.c_int, C.POINTER(vp)]
    lib.mw_align_destroy.argtypes = [vp]
    lib.mw_align_destroy.restype = None
    lib.mw_align_last_error.argtypes = [vp]
    lib.mw_align_last_error.restype = C.c_char_p
    lib.mw_align_set_weight.argtypes = [vp, C.c_char_p, vp, C.POINTER(C.c_int64), C.c_int, C.c_int]
    lib.mw_align_finalize.argtypes = [vp]
    lib.mw_align_emissions.argtypes = [vp, f32p, C.c_int64, f32p, C.POINTER(C.c_int32)]
    lib.mw_align_fork.argtypes = [vp, C.POINTER(vp)]
    lib.mw_align_segment.argtypes = [vp, f32p, C.c_int64, vp, C.c_int32, C.c_int32, vp, vp, C.POINTER(C.c_int32)]
    lib.mw_align_segments.argtypes = [vp, C.c_int32, vp, vp, vp, vp, C.c_int32, vp, vp, vp, vp]
    lib.mw_align_path.argtypes = [vp, C.c_int32, C.c_int32, vp, C.c_int32, C.c_int32, vp, vp]
    for name in SYMBOLS:
        if name not in ("mw_destroy", "mw_last_error", "mw_align_destroy", "mw_align_last_error", "mw_align_frames", "mw_sample_uniform"):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib
