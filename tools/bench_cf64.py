import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch, spectral_analyzer_amd as sa
exec(open('/root/repo/tools/bench_other.py').read().split('which = sys.argv')[0])
for n in (1024, 4096, 8192):
    spectro("cf64_le", n, n // 2, 27, fmt=sa.OUT_DB20_F64, label="cf64->f64 n=%d" % n)
    spectro("cf64_le", n, n // 2, 27, fmt=sa.OUT_DB20_F32, label="cf64->f32 n=%d" % n)
    spectro("cf32_le", n, n // 2, 27, fmt=sa.OUT_DB20_F64, label="cf32->f64 n=%d" % n)
