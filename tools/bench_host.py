#!/usr/bin/env python3
"""Host-side (PCIe-inclusive) rates of the spectrogram path, as a Java host drives it:
  mapped   spec_waterfall with the input in pageable host memory (the MappedByteBuffer) and the output in a
           host array -- the two-deep staged pipeline of DESIGN.md 6
  file     spec_waterfall_recording: the data file (in the page cache) opened by path -- staged from the library's
           own mapping of it (default), or pread into the pinned two-slot ring ("rec_pread"); output to the same
           host array
python tools/bench_host.py [log2_samples=27]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (before the library: one HIP runtime per process, see _lib.load)
import spectral_analyzer_amd as sa
from spectral_analyzer_amd import sigmf

log2s = int(sys.argv[1]) if len(sys.argv) > 1 else 27
svc = sa.SpectralService(0)
tmp = tempfile.mkdtemp(prefix="spec_host_")
for dt, nfft, hop in (("cf32_le", 4096, 2048), ("ci16_le", 4096, 2048), ("cf32_le", 1024, 1024)):
    S = 1 << log2s
    host = svc.synth_iq(dt, 3, 0, S).cpu().numpy()
    n = (S - nfft) // hop + 1
    out = np.empty((n, nfft), dtype=np.float32)
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        svc.compute_waterfall(host, 0, nfft, dt, n, hop=hop, out=out)
        best = min(best, time.perf_counter() - t0)
    moved = host.nbytes + out.nbytes
    ref = out.copy()
    line = "%-8s nfft %5d hop %5d  %8d lines  mapped %7.1f ms %6.2f Mlines/s %5.1f GB/s" % (
        dt, nfft, hop, n, best * 1e3, n / best / 1e6, moved / best / 1e9)
    # the same recording as a file, read by the library (pread -> pinned ring -> device)
    meta = os.path.join(tmp, "r.sigmf-meta")
    host.tofile(os.path.join(tmp, "r.sigmf-data"))
    open(meta, "w").write('{"global": {"core:datatype": "%s"}, "captures": [{}]}' % dt)
    with sigmf.load(meta).open_native(svc) as nat:
        lib, ctx = svc._lib, svc._ctx
        res = {}
        for mode in (0, 1):                                  # 0: the library's mapping of the file, 1: pread + pinned ring
            svc.set_option("rec_pread", mode)
            bestf = 1e9
            for _ in range(4):
                t0 = time.perf_counter()
                st = lib.spec_waterfall_recording(ctx, nat._h, 0, sa.dtype_from_sigmf(dt), nfft, hop, n, 0, 0, -150.0,
                                                  out.ctypes.data, 0)
                bestf = min(bestf, time.perf_counter() - t0)
                assert st == 0
            assert np.array_equal(out, ref)
            res[mode] = bestf
        svc.set_option("rec_pread", 0)
    print(line + "   file, mapped by the library %7.1f ms %5.1f GB/s   file, pread + pinned ring %7.1f ms %5.1f GB/s   "
          "(in %.2f GB + out %.2f GB over PCIe)" % (res[0] * 1e3, moved / res[0] / 1e9, res[1] * 1e3, moved / res[1] / 1e9,
                                                    host.nbytes / 1e9, out.nbytes / 1e9), flush=True)
    os.unlink(os.path.join(tmp, "r.sigmf-data"))
