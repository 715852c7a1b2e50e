"""spectral_analyzer_amd -- MI355X-native spectrogram / PSD path.

Only the hot path of GassiusODude/spectral_analyzer lives here: sample reader ->
window -> FFT -> |X| -> 20 log10 / Welch PSD, as hand-written HIP kernels for
gfx950 behind the C ABI of ``include/specgpu.h``.  ``SpectralService`` mirrors
the reference's service class on top of that ABI.
"""
from ._lib import (CMAP_GRAYSCALE, CMAP_HEATMAP, DT_CF32_BE, DT_CF32_LE, DT_CF64_BE, DT_CF64_LE, DT_CI8, DT_CI16_BE, DT_CI16_LE,  # noqa: F401
                   DT_CU8, DT_UNKNOWN, OUT_DB20_F32, OUT_DB20_F64, OUT_POW_F32, OUT_POW_F64,
                   PSD_DENSITY, PSD_SPECTRUM, WIN_HANN, WIN_RECT)
from .spectral_service import (SpectralService, bytes_per_sample, compute_waterfall_multi, dtype_from_sigmf,  # noqa: F401
                               shard_lines, shard_span, welch_psd_multi)

from .extract_down_convert_service import ExtractDownConvertService  # noqa: F401,E402
from . import sigmf  # noqa: F401,E402

__all__ = ["SpectralService", "ExtractDownConvertService", "bytes_per_sample", "dtype_from_sigmf", "sigmf",
           "compute_waterfall_multi", "welch_psd_multi", "shard_lines", "shard_span"]
