import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch, spectral_analyzer_amd as sa
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
svc = sa.SpectralService(0, stream=st.cuda_stream)
nfft, hop, nseg = 16384, 4096, 256
per = (nseg - 1) * hop + nfft
iq = svc.synth_iq("cf32_le", 3, 0, per)
for run in (0, 1, 2, 4, 8, 16):
    svc.set_option("lines_per_wg", run)
    for _ in range(10): svc.welch_psd(iq, 0, "cf32_le", 1e6, nfft=nfft, hop=hop, n_seg=nseg)
    ev = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); svc.welch_psd(iq, 0, "cf32_le", 1e6, nfft=nfft, hop=hop, n_seg=nseg); b.record(st); ev.append((a, b))
    torch.cuda.synchronize()
    print("run %2d: %.1f us" % (run, 1e3 * float(np.median([a.elapsed_time(b) for a, b in ev]))))
