#!/usr/bin/env python3
"""What this box's memory system does on plain streams, beside the roofline fractions of bench.py (which price
against the nominal 8 TB/s): device-to-device copy (read + write), fill (write only), reduction (read only) on
4 GiB buffers, medians of HIP-event times.     python tools/mem_ceiling.py"""
import numpy as np, torch

n = 1 << 30                     # 2^30 floats = 4 GiB
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)

def timeit(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    ev = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); ev.append((s, e))
    torch.cuda.synchronize()
    return float(np.median([s.elapsed_time(e) for s, e in ev]))

gb = a.numel() * 4 / 1e9
for name, fn, moved in (("copy (read + write)", lambda: b.copy_(a), 2 * gb), ("fill (write only)", lambda: b.zero_(), gb),
                        ("fill 1.0 (write only)", lambda: b.fill_(1.0), gb), ("sum (read only)", lambda: a.sum(), gb),
                        ("add in place (read + write)", lambda: b.add_(1.0), 2 * gb),
                        ("a + b -> b (2 reads + write)", lambda: torch.add(a, b, out=b), 3 * gb)):
    ms = timeit(fn)
    print("%-32s %8.3f ms  %7.0f GB/s" % (name, ms, moved / ms * 1e3), flush=True)
