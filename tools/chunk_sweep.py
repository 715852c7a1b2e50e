import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectral_analyzer_amd as sa
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
svc = sa.SpectralService(0, stream=st.cuda_stream)
dt, nfft, hop, S = "cf64_le", 65536, 32768, 1 << 28
n = (S - nfft) // hop + 1
iq = svc.synth_iq(dt, 7, 0, S)
out = torch.empty((n, nfft), dtype=torch.float64, device="cuda")
for mb in (16, 32, 64, 96, 128, 192, 256, 512, 1024, 4096):
    svc.set_option("large_chunk_mb", mb)
    ts = []
    for r in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); svc.compute_waterfall(iq, 0, nfft, dt, n, hop=hop, out_fmt=sa.OUT_DB20_F64, out=out); b.record(st)
        torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ms = float(np.median(ts[2:]))
    print("chunk %4d MiB: %.3f ms  %.2f Mlines/s  %.1f%% of 8 TB/s" % (mb, ms, n / ms / 1e3, n * 1048576 / ms / 1e6 / 80))
