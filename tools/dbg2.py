import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectral_analyzer_amd as sa
from oracle import spec_oracle as so
svc = sa.SpectralService(0)
dt = "ci16_be"
n, first = 5000, 123456789
dev = svc.synth_iq(dt, 0x5EC7A11A, first, n).cpu().numpy()
host = so.synth_iq(dt, 0x5EC7A11A, first, n)
a = so.np_decode(dev, 0, n, dt); b = so.np_decode(host, 0, n, dt)
i = int(np.argmax(np.abs(a - b)))
print(i, a[i], b[i], dev[4*i:4*i+4], host[4*i:4*i+4], "n bad", int((np.abs(a-b) > 1e-3).sum()))
