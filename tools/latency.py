"""The reference's own call shape, timed end to end from host memory (PCIe and stream syncs included): a 1000-column
redraw at hop = nfft (MainController.updateDisplay, MC:962-1049) for several NFFT slider positions --
  * the UNMODIFIED loop: 1000 computeMagnitudes calls, one per slice (MC:982-993), without and with the library's read-ahead
  * ONE spec_waterfall call; ONE spec_waterfall_render call (only the image comes back)
  * the same redraw from a recording opened BY PATH (spec_open_recording / spec_waterfall_recording)
  * the CPU oracle on one thread (the reference runs on the single JavaFX thread)
Development tool; python tools/latency.py > profiles/rNN_latency.txt"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spectral_analyzer_amd as sa
from spectral_analyzer_amd import sigmf
from oracle import spec_oracle as so
svc = sa.SpectralService(0)
tmp = tempfile.mkdtemp(prefix="spec_lat_")
W = 1000
for dt, nfft in (("ci16_le", 1024), ("cf32_le", 4096), ("cf32_le", 32768), ("cf32_le", 65536)):
    bps = so.bytes_per_sample(dt)
    iq = so.synth_iq(dt, 1, 0, W * nfft)
    for _ in range(20): svc.compute_magnitudes(iq, 0, nfft, dt)
    t0 = time.perf_counter(); n = 200
    for i in range(n): svc.compute_magnitudes(iq, 0, nfft, dt)
    one = (time.perf_counter() - t0) / n
    for _ in range(3): svc.compute_waterfall(iq, 0, nfft, dt, W)
    t0 = time.perf_counter()
    for i in range(10): svc.compute_waterfall(iq, 0, nfft, dt, W)
    wf = (time.perf_counter() - t0) / 10
    rh = min(600, nfft)
    for _ in range(3): svc.waterfall_render(iq, 0, nfft, dt, W, rh, 1e6)
    t0 = time.perf_counter()
    for i in range(10): svc.waterfall_render(iq, 0, nfft, dt, W, rh, 1e6)
    wr = (time.perf_counter() - t0) / 10
    reps = 3 if nfft <= 4096 else 1
    t0 = time.perf_counter()
    for i in range(reps): so.waterfall(iq, 0, dt, nfft, nfft, W)
    cpu = (time.perf_counter() - t0) / reps
    # the UNMODIFIED reference loop (MC:982-993): one computeMagnitudes call per slice, slice after slice
    walk = {}
    for ra in (0, 256):
        svc.set_option("readahead_lines", ra)
        for _ in range(2): [svc.compute_magnitudes(iq, t * nfft * bps, nfft, dt) for t in range(W)]
        t0 = time.perf_counter()
        for _ in range(3): [svc.compute_magnitudes(iq, t * nfft * bps, nfft, dt) for t in range(W)]
        walk[ra] = (time.perf_counter() - t0) / 3
    # the same redraw from the data file, handed over by path
    iq.tofile(os.path.join(tmp, "r.sigmf-data"))
    open(os.path.join(tmp, "r.sigmf-meta"), "w").write('{"global": {"core:datatype": "%s"}, "captures": [{}]}' % dt)
    with sigmf.load(os.path.join(tmp, "r.sigmf-meta")).open_native(svc) as nat:
        for _ in range(3): nat.waterfall(0, nfft, W)
        t0 = time.perf_counter()
        for i in range(10): got = nat.waterfall(0, nfft, W)
        byp = (time.perf_counter() - t0) / 10
        assert np.array_equal(got, svc.compute_waterfall(iq, 0, nfft, dt, W))
    os.unlink(os.path.join(tmp, "r.sigmf-data"))
    print("%s nfft %5d: computeMagnitudes %.1f us/call | %d-line redraw at hop = nfft: batched %.2f ms, batched+render (%d rows) %.2f ms, "
          "recording by path %.2f ms, %d per-slice calls %.1f ms without / %.2f ms with read-ahead, CPU oracle 1 thread %.1f ms"
          % (dt, nfft, one * 1e6, W, wf * 1e3, rh, wr * 1e3, byp * 1e3, W, walk[0] * 1e3, walk[256] * 1e3, cpu * 1e3), flush=True)
