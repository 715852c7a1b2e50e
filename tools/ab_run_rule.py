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
for dt in ("cf32_le", "ci16_le"):
  for nfft in (1024, 2048, 4096):
    for lg in (26, 28, 30):
        S = 1 << lg; hop = nfft; n = S // hop; bps = sa.bytes_per_sample(dt)
        iq = svc.synth_iq(dt, 7, 0, S); out = torch.empty((n, nfft), dtype=torch.float32, device="cuda")
        res = []
        for rep in range(2):
            for k in (32, 0):
                svc.set_option("lines_per_wg", k)
                ms = timeit(lambda: svc.compute_waterfall(iq, 0, nfft, dt, n, hop=hop, out=out))
                res.append(n * (hop * bps + nfft * 4) / ms / 1e6 / 8000)
        svc.set_option("lines_per_wg", 0)
        print("%-8s n=%d 2^%d samples: runs of 32: %.3f %.3f   rule: %.3f %.3f" % (dt, nfft, lg, res[0], res[2], res[1], res[3]), flush=True)
        del iq, out; torch.cuda.empty_cache()
