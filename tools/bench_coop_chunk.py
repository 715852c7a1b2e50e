import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import spectral_analyzer_amd as sa
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
svc = sa.SpectralService(0, stream=st.cuda_stream)
def timeit(fn, reps=12, warm=8):
    for _ in range(warm): fn()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); fn(); b.record(st); ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
CH = (0, 1, 2, 4, 8, 16)
print("%-34s" % "blocks per chunk" + "".join("%8d" % c for c in CH))
for dt, nfft in (("cf32_le", 64), ("cf32_le", 128), ("ci16_le", 64), ("ci16_le", 128), ("cu8", 256)):
    for hop in (nfft, nfft // 2):
        for lg in (28, 30):
            S = 1 << lg; n = (S - nfft) // hop + 1; bps = sa.bytes_per_sample(dt)
            if n * nfft * 4 > (24 << 30): continue
            iq = svc.synth_iq(dt, 7, 0, S); out = torch.empty((n, nfft), dtype=torch.float32, device="cuda")
            row = []
            for c in CH:
                svc.set_option("lines_per_wg", c)
                ms = timeit(lambda: svc.compute_waterfall(iq, 0, nfft, dt, n, hop=hop, out=out))
                row.append(n * (hop * bps + nfft * 4) / ms / 1e6 / 8000)
            svc.set_option("lines_per_wg", 0)
            print("%-8s n=%-4d hop=%-4d 2^%d     " % (dt, nfft, hop, lg) + "".join("%8.3f" % x for x in row), flush=True)
            del iq, out; torch.cuda.empty_cache()
