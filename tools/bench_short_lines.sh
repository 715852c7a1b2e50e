#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "short_lines or waterfall_matches" 2>&1 | tail -12
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
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
for dt in ("cf32_le", "ci16_le", "cu8"):
    for n, hop in ((64, 32), (64, 64), (128, 64), (128, 128)):
        bps = sa.bytes_per_sample(dt); S = 1 << 28
        nl = (S - n) // hop + 1
        iq = svc.synth_iq(dt, 7, 0, S)
        out = torch.empty((nl, n), dtype=torch.float32, device="cuda")
        for gen in (0, 1):
            svc.set_option("force_generic", gen)
            ms = timeit(lambda: svc.compute_waterfall(iq, 0, n, dt, nl, hop=hop, out=out))
            b = nl * (hop * bps + n * 4)
            print("%-8s n=%-4d hop=%-4d %s  %8.3f ms  %8.1f Mlines/s  %6.0f GB/s (%.1f%% of 8 TB/s)" % (dt, n, hop, "generic" if gen else "v2n    ", ms, nl / ms / 1e3, b / ms / 1e6, b / ms / 1e6 / 80), flush=True)
        svc.set_option("force_generic", 0)
        del iq, out; torch.cuda.empty_cache()
PY
