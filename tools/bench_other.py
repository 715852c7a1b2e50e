#!/usr/bin/env python3
"""Timings of the other BASELINE configurations (development tool):
cfg1 1024-pt, cfg4 Welch 16384 batch, cfg5 65536-pt cf64, plus generic sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectral_analyzer_amd as sa

st = torch.cuda.Stream(); torch.cuda.set_stream(st)
svc = sa.SpectralService(0, stream=st.cuda_stream)

def timeit(fn, reps=8, warm=6):
    for _ in range(warm): fn()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); fn(); b.record(st); ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))

def spectro(dt, nfft, hop, log2s, fmt=sa.OUT_DB20_F32, window=0, label=""):
    bps = sa.bytes_per_sample(dt); S = 1 << log2s
    n = (S - nfft) // hop + 1
    iq = svc.synth_iq(dt, 7, 0, S)
    out = torch.empty((n, nfft), dtype=torch.float32 if fmt < 2 else torch.float64, device="cuda")
    ms = timeit(lambda: svc.compute_waterfall(iq, 0, nfft, dt, n, hop=hop, window=window, out_fmt=fmt, out=out))
    osz = 4 if fmt < 2 else 8
    b = n * (hop * bps + nfft * osz)
    print("%-34s %9d lines  %8.3f ms  %8.2f Mlines/s  %7.0f GB/s algorithmic (%.1f%% of 8 TB/s)" % (
        label or "%s n=%d hop=%d" % (dt, nfft, hop), n, ms, n / ms / 1e3, b / ms / 1e6, b / ms / 1e6 / 80), flush=True)
    del iq, out; torch.cuda.empty_cache()

which = sys.argv[1:] or ["cfg1", "sizes", "welch", "cfg5"]
if "cfg1" in which:
    spectro("cf32_le", 1024, 512, 20, label="cfg1 1024/512 cf32 2^20")
    spectro("cf32_le", 1024, 512, 28, label="1024/512 cf32 2^28")
if "sizes" in which:
    for n in (64, 128, 256, 512, 1024, 2048):
        spectro("cf32_le", n, n // 2, 28)
    for n in (8192, 16384):  # long lines: 2^28 samples are only a few workgroup rounds
        spectro("cf32_le", n, n // 2, 30)
    spectro("cf32_le", 4096, 4096, 28, label="4096 hop=nfft (reference) cf32")
    spectro("cf32_le", 4096, 2048, 28, window=1, label="4096/2048 cf32 hann")
    spectro("cu8", 4096, 2048, 28, label="4096/2048 cu8")
    spectro("cf32_be", 4096, 2048, 28, label="4096/2048 cf32_be")
if "refhop" in which:  # the reference's own call shape: hop = nfft (MC:984-985), every tick of its NFFT slider (main-scene.fxml:129-132)
    for dt in ("cf32_le", "ci16_le"):
        for lg in range(6, 17):
            spectro(dt, 1 << lg, 1 << lg, 28 if lg < 13 else 29, label="%s n=%d hop=nfft" % (dt, 1 << lg))
if "bytefmt" in which:  # the 2-byte formats against ci16 at the same sizes: lines/s should match where the kernel is not load-granularity bound
    for lg in (9, 10, 11, 12):
        for hop_div in (1, 2):
            for dt in ("ci16_le", "cu8", "ci8"):
                spectro(dt, 1 << lg, (1 << lg) // hop_div, 28)
if "cfg5only" in which:
    spectro("cf64_le", 65536, 32768, 28, fmt=sa.OUT_DB20_F64, label="cfg5 65536/32768 cf64->f64 2^28")
if "cfg5" in which:
    spectro("cf64_le", 65536, 32768, 28, fmt=sa.OUT_DB20_F64, label="cfg5 65536/32768 cf64->f64 2^28")
    spectro("cf32_le", 65536, 32768, 28, label="65536/32768 cf32->f32 2^28")
    spectro("cf32_le", 32768, 16384, 28, label="32768/16384 cf32->f32 2^28")
if "welch" in which:
    nfft, hop, nseg = 16384, 4096, 256
    per = (nseg - 1) * hop + nfft
    for n_psd in (1, 64, 1024):
        iq = svc.synth_iq("cf32_le", 3, 0, per * n_psd)
        ms = timeit(lambda: svc.welch_psd(iq, 0, "cf32_le", 1e6, nfft=nfft, hop=hop, n_seg=nseg, n_psd=n_psd,
                                          psd_stride_bytes=per * 8))
        b = n_psd * (per * 8 + nfft * 4)
        print("cfg4 welch 16384/4096 x256 seg, %4d PSDs: %8.3f ms  %8.1f PSD/s  %7.0f GB/s algorithmic (%.1f%%)" % (
            n_psd, ms, n_psd / ms * 1e3, b / ms / 1e6, b / ms / 1e6 / 80), flush=True)
        del iq
