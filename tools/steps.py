"""Per-step kernel times of the bench workload, back-to-back and with idle gaps
(development tool: shows what sustained load does to the clock)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import spectral_analyzer_amd as sa

st = torch.cuda.Stream()
torch.cuda.set_stream(st)
svc = sa.SpectralService(0, stream=st.cuda_stream)
dt, nfft, hop = "cf32_le", 4096, 2048
S = 1 << 30
n_lines = (S - nfft) // hop + 1
iq = svc.synth_iq(dt, 1, 0, S)
out = torch.empty((n_lines, nfft), dtype=torch.float32, device="cuda")
for gap in (0.0, 0.05, 0.0):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for a, b in ev:
        a.record(st)
        svc.compute_waterfall(iq, 0, nfft, dt, n_lines, hop=hop, out=out)
        b.record(st)
        if gap:
            torch.cuda.synchronize()
            time.sleep(gap)
    torch.cuda.synchronize()
    print("gap", gap, " ".join("%.2f" % a.elapsed_time(b) for a, b in ev))
