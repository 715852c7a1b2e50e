#!/usr/bin/env python3
"""Development aid: time line of the eight waves of one workgroup of the 16384-point Welch kernel (spec_v2.h, MODE 1) over
three consecutive segments -- lane 0 of every wave notes the shader clock (s_memtime) at each phase boundary.
Needs the stamp variant of the library:
    python -m spectral_analyzer_amd.build --variant v2stamp
    SPEC_LIB_VARIANT=v2stamp python tools/v2_timeline.py [rows=0|1] [first_segment] [workgroup]
Prints, per segment and phase, when each wave reached the boundary (cycles after the workgroup's first stamp of the first
stamped segment) and how long the phase took for the fastest / slowest wave.
"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
import spectral_analyzer_amd as sa  # noqa: E402
from spectral_analyzer_amd import _lib as L  # noqa: E402

PHASES = {
    0: {0: "line top", 1: "window + decode + shift + requests", 2: "pass 0 (radix 32) [+ its stores: fused]", 3: "(late WAR barrier)",
        4: "exchange-0 stores left", 5: "barrier (RAW)", 6: "exchange-0 load + early WAR barrier", 7: "pass 1 (radix 32) [+ stores]",
        8: "(late WAR barrier)", 9: "exchange-1 stores left", 10: "barrier (RAW)", 11: "exchange-1 load + early WAR barrier",
        12: "pass 2 (2 x radix 16)", 15: "|X|^2 sums"},
    1: {0: "line top", 1: "window + decode + shift + requests", 2: "pass 0 (2 x radix 16 + twiddles)", 3: "(late WAR barrier)",
        4: "exchange-0 store", 5: "barrier (RAW)", 6: "row load + pass A (radix 32)", 7: "row exchange store (no barrier)",
        8: "row load + twiddles + early WAR barrier", 9: "pass B (radix 32)", 15: "|X|^2 sums"},
}


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    wg = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    lib = L.load()
    lib.spec_debug_v2_stamps.restype, lib.spec_debug_v2_stamps.argtypes = None, [C.c_void_p, C.c_uint32]
    svc = sa.SpectralService(0)
    svc.set_option("welch_rows", rows)
    nfft, hop, n_seg, n_psd = 16384, 4096, 256, 1024
    per_psd = (n_seg - 1) * hop + nfft
    iq = svc.synth_iq("cf32_le", 7, 0, per_psd * n_psd)
    out = torch.empty((n_psd, nfft), dtype=torch.float32, device=iq.device)
    buf = torch.zeros(8 * 8 * 48, dtype=torch.int32, device=iq.device)
    lib.spec_debug_v2_stamps(buf.data_ptr(), first)
    for _ in range(3):
        svc.welch_psd(iq, 0, "cf32_le", 1.0e6, nfft=nfft, hop=hop, n_seg=n_seg, window=1, n_psd=n_psd, psd_stride_bytes=per_psd * 8, out=out)
    torch.cuda.synchronize()
    st = buf.cpu().numpy().astype(np.uint32).reshape(8, 8, 3, 16)[wg]  # [wave][segment][phase]
    names = PHASES[rows]
    ids = sorted(names)
    t0 = st[:, 0, 0].min()
    rel = (st - t0).astype(np.int64)  # uint32 wrap-around handled by the unsigned subtraction
    print("welch_rows = %d, workgroup %d, segments %d..%d; cycles after the first wave's first stamp" % (rows, wg, first, first + 2))
    print("%-46s %s   | phase length min / max over the waves" % ("boundary reached (end of phase)", " ".join("wave%d" % w for w in range(8))))
    for seg in range(3):
        print("segment %d" % (first + seg))
        prev = None
        for k in ids:
            cur = rel[:, seg, k]
            line = "  %-44s %s" % (names[k], " ".join("%5d" % c for c in cur))
            if prev is not None:
                d = cur - prev
                line += "   | %5d / %5d" % (d.min(), d.max())
            print(line)
            prev = cur
    seg_len = rel[:, 1:, 0] - rel[:, :-1, 0]
    print("segment length per wave (top to top):", seg_len.mean(axis=1).round().astype(int).tolist(), "cycles")
    svc.close()


if __name__ == "__main__":
    main()
