import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import spectral_analyzer_amd as sa
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
svc = sa.SpectralService(0, stream=st.cuda_stream)
def timeit(fn, reps=8, warm=5):
    for _ in range(warm): fn()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); fn(); b.record(st); ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
for dt in ("cf64_le", "cf32_le", "ci16_le"):
    for nfft in (1024, 4096, 8192, 16384):
        S = 1 << 27; hop = nfft // 2; n = (S - nfft) // hop + 1
        iq = svc.synth_iq(dt, 7, 0, S)
        out = torch.empty((n, nfft), dtype=torch.float64, device="cuda")
        r = [timeit(lambda: svc.compute_waterfall(iq, 0, nfft, dt, n, hop=hop, window=w, out_fmt=sa.OUT_DB20_F64, out=out)) for w in (0, 1)]
        print("%-8s -> f64 n=%-5d rect %.3f ms  hann %.3f ms  (+%.0f %%)" % (dt, nfft, r[0], r[1], 100 * (r[1] / r[0] - 1)), flush=True)
        del iq, out; torch.cuda.empty_cache()
