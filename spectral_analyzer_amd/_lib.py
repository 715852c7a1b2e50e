"""ctypes binding of the C ABI in include/specgpu.h (libspecgpu.so).

The library is the product; this module only loads it.  There is no Python or
CPU fallback: if the shared object is missing or fails to load, importing
callers get an ImportError telling them to run ``python -m
spectral_analyzer_amd.build``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libspecgpu.so")
# experiments only (tools/ablate.sh): SPEC_LIB_VARIANT=<name> loads lib/libspecgpu_<name>.so, a library built
# with ablation macros whose results are wrong by construction; the product path never sets it
if os.environ.get("SPEC_LIB_VARIANT"):
    LIB_PATH = os.path.join(_HERE, "lib", "libspecgpu_%s.so" % os.environ["SPEC_LIB_VARIANT"])

# spec_status
SPEC_OK, SPEC_EINVAL, SPEC_ERANGE, SPEC_EDEVICE, SPEC_ENOMEM, SPEC_EUNSUPPORTED = range(6)
# spec_dtype
(DT_UNKNOWN, DT_CU8, DT_CI8, DT_CI16_LE, DT_CI16_BE, DT_CF32_LE, DT_CF32_BE, DT_CF64_LE,
 DT_CF64_BE) = range(9)
WIN_RECT, WIN_HANN = 0, 1
OUT_DB20_F32, OUT_POW_F32, OUT_DB20_F64, OUT_POW_F64 = range(4)
PSD_DENSITY, PSD_SPECTRUM = 0, 1
CMAP_GRAYSCALE, CMAP_HEATMAP = 0, 1
FLAG_REF_CF64_ZERO = 0x1
FLAG_NULL_STREAM = 0x2
FLAG_REF_EDC_CF64_STRIDE8 = 0x4
DC_FAST, DC_LPF = 0, 1

# every symbol include/specgpu.h declares, with its ctypes signature
_u64, _u32, _i32, _vp, _cp, _dbl = C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_char_p, C.c_double
SIGNATURES = {
    "spec_create": (_i32, [_i32, _vp, _u32, C.POINTER(_vp)]),
    "spec_destroy": (None, [_vp]),
    "spec_last_error": (_cp, [_vp]),
    "spec_status_string": (_cp, [_i32]),
    "spec_sync": (_i32, [_vp]),
    "spec_stream": (_vp, [_vp]),
    "spec_set_option": (_i32, [_vp, _cp, C.c_int64]),
    "spec_get_option": (_i32, [_vp, _cp, C.POINTER(C.c_int64)]),
    "spec_dtype_from_sigmf": (_i32, [_cp]),
    "spec_bytes_per_sample": (_u32, [_i32]),
    "spec_count_lines": (_u64, [_u64, _u64, _i32, _u32, _u32]),
    "spec_compute_magnitudes": (_i32, [_vp, _vp, _u64, C.c_int64, _u32, _cp, _i32, _vp]),
    "spec_waterfall": (_i32, [_vp, _vp, _i32, _u64, _u64, _i32, _u32, _u32, _u64, _i32, _i32, _dbl, _vp, _i32]),
    "spec_shard_lines": (None, [_u64, _u32, _u32, C.POINTER(_u64), C.POINTER(_u64)]),
    "spec_shard_span": (None, [_u64, _u64, _i32, _u32, _u32, C.POINTER(_u64), C.POINTER(_u64)]),
    "spec_waterfall_multi": (_i32, [C.POINTER(_vp), _u32, C.POINTER(_vp), _i32, _u64, _u64, _i32, _u32, _u32, _u64, _i32,
                                    _i32, _dbl, _vp, _i32, _u32]),
    "spec_open_recording": (_i32, [_vp, _cp, _u64, C.POINTER(_vp)]),
    "spec_recording_bytes": (_u64, [_vp]),
    "spec_close_recording": (None, [_vp]),
    "spec_waterfall_recording": (_i32, [_vp, _vp, _u64, _i32, _u32, _u32, _u64, _i32, _i32, _dbl, _vp, _i32]),
    "spec_compute_magnitudes_recording": (_i32, [_vp, _vp, _u64, _u32, _cp, _i32, _vp]),
    "spec_welch_psd": (_i32, [_vp, _vp, _i32, _u64, _u64, _u64, _u32, _i32, _u32, _u32, _u32, _i32, _i32,
                              _dbl, _i32, _vp, _vp, _i32]),
    "spec_welch_psd_multi": (_i32, [C.POINTER(_vp), _u32, C.POINTER(_vp), _i32, C.POINTER(_u64), _u64, _u64, _u32, _i32, _u32, _u32,
                                    _u32, _i32, _i32, _dbl, _i32, _vp, _vp, _i32]),
    "spec_render_spectrogram": (_i32, [_vp, _vp, _i32, _u32, _u32, _u32, _dbl, _dbl, _dbl, _i32, _vp, _i32]),
    "spec_waterfall_render": (_i32, [_vp, _vp, _i32, _u64, _u64, _i32, _u32, _u32, _u32, _i32, _u32, _dbl, _dbl,
                                     _dbl, _i32, _vp, _i32]),
    "spec_welch_psd_planar_f64": (_i32, [_vp, _vp, _vp, _i32, _u64, _u32, _u32, _i32, _i32, _dbl, _i32, _vp, _vp]),
    "spec_extract_iq": (_i32, [_vp, _vp, _i32, _u64, _u64, _u64, _i32, _vp, _vp, _i32]),
    "spec_down_convert": (_i32, [_vp, _vp, _i32, _u64, _u64, _u64, _i32, _dbl, _u32, _i32, _vp, _vp, _i32]),
    "spec_magnitude_trace": (_i32, [_vp, _vp, _vp, _i32, _u64, _dbl, _vp, _i32]),
    "spec_inst_freq_trace": (_i32, [_vp, _vp, _vp, _i32, _u64, _dbl, _dbl, _dbl, _vp, _i32]),
    "spec_synth_iq": (_i32, [_vp, _vp, _i32, _u64, _u64, _u64]),
}

_lib = None


def load() -> C.CDLL:
    """Load libspecgpu.so; raises ImportError (never falls back) when absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libspecgpu.so is not built (%s). Run `python -m spectral_analyzer_amd.build`; "
                "there is no CPU fallback." % LIB_PATH)
        # PyTorch's wheel carries its own ROCm runtime libraries.  If libspecgpu.so (linked against the system
        # ROCm's libamdhip64) is loaded first, torch later brings in a second HIP runtime and finds no GPU
        # ("No HIP GPUs are available").  Loading torch first makes both use the one runtime; hosts without
        # torch (C, C++, Java) are not affected.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:  # missing ROCm runtime etc.
            raise ImportError("cannot load %s: %s" % (LIB_PATH, e)) from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib
