#!/usr/bin/env python3
"""Names the Welch convention (and prices the down-converter) that reproduces what JDSP returned for the probe signals.

    python tools/fit_jdsp.py <observed directory> [--probe integration/java-test/jdsp-probe]

<observed directory> = what integration/java-test/JdspSemanticsProbe.java wrote on a machine that has JDSP (the
reference's Gradle build does): <name>.psd.f64 (freq row, psd row), <name>.dc_fast.f64 / <name>.dc_conv.f64, observed.json.
For every signal the build's Welch estimate is evaluated -- in plain numpy, independent of the library and of the oracle --
under every combination of
    window   rect | hann | hamming | blackman         overlap  0 | 50 | 75 %         detrend  none | mean
    scaling  density (/ (fs sum w^2)) | spectrum (/ (sum w)^2) | raw (/ nfft) | raw2 (/ nfft^2)
    output   linear | 10 log10                         order    fftshifted (-fs/2 first) | natural (0 first)
and the combinations are ranked by the largest relative error against the recording; a convention that reproduces JDSP on
ALL signals is the one the library's defaults (include/specgpu.h: Hann, 50 %, density, dB) must be changed to.  Windows the
library does not have yet (hamming, blackman) are in the list so that the answer is a name, not "none fits".
"""
import itertools
import json
import os
import sys

import numpy as np

WINDOWS = {
    "rect": lambda n: np.ones(n),
    "hann": lambda n: 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n),
    "hann_symmetric": lambda n: np.hanning(n),
    "hamming": lambda n: 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(n) / n),
    "blackman": lambda n: 0.42 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n) + 0.08 * np.cos(4 * np.pi * np.arange(n) / n),
}
OVERLAPS = {"0 %": 1.0, "50 %": 0.5, "75 %": 0.25}
SCALINGS = ("density", "spectrum", "raw", "raw2")


def welch(x, fs, nfft, window, hop_frac, detrend, scaling, db, shifted):
    w = WINDOWS[window](nfft)
    hop = max(1, int(round(nfft * hop_frac)))
    n_seg = (len(x) - nfft) // hop + 1
    acc = np.zeros(nfft)
    for s in range(n_seg):
        seg = x[s * hop:s * hop + nfft]
        if detrend:
            seg = seg - seg.mean()
        acc += np.abs(np.fft.fft(seg * w)) ** 2
    acc /= n_seg
    norm = {"density": fs * (w ** 2).sum(), "spectrum": w.sum() ** 2, "raw": float(nfft), "raw2": float(nfft) ** 2}[scaling]
    p = acc / norm
    if shifted:
        p = np.fft.fftshift(p)
    return 10 * np.log10(p + 1e-300) if db else p


def hypotheses():
    for window, ov, detrend, scaling, db, shifted in itertools.product(WINDOWS, OVERLAPS, (False, True), SCALINGS, (False, True), (True, False)):
        yield {"window": window, "overlap": ov, "detrend": "mean" if detrend else "none", "scaling": scaling,
               "output": "10 log10" if db else "linear", "order": "fftshifted" if shifted else "natural"}


def evaluate(h, x, fs, nfft):
    return welch(x, fs, nfft, h["window"], OVERLAPS[h["overlap"]], h["detrend"] == "mean", h["scaling"], h["output"] != "linear",
                 h["order"] == "fftshifted")


def error(got, want, db):
    if got.shape != want.shape:
        return float("inf")
    if db:
        return float(np.abs(got - want).max())                       # dB
    return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-300))


def load_signal(probe_dir, name):
    re = np.fromfile(os.path.join(probe_dir, name + ".re.f64"), dtype="<f8")
    im = np.fromfile(os.path.join(probe_dir, name + ".im.f64"), dtype="<f8")
    return re + 1j * im


def fit(observed_dir, probe_dir, verbose=True):
    meta = json.load(open(os.path.join(probe_dir, "probe.json")))
    fs = meta["fs"]
    total = {}
    per_signal = {}
    for s in meta["signals"]:
        path = os.path.join(observed_dir, s["name"] + ".psd.f64")
        if not os.path.exists(path):
            continue
        obs = np.fromfile(path, dtype="<f8")
        freq, psd = obs[:len(obs) // 2], obs[len(obs) // 2:]
        x = load_signal(probe_dir, s["name"])
        nfft = s["nfft"]
        ranked = []
        for h in hypotheses():
            e = error(evaluate(h, x, fs, nfft), psd, h["output"] != "linear")
            key = json.dumps(h, sort_keys=True)
            ranked.append((e, key))
            total[key] = max(total.get(key, 0.0), e)
        ranked.sort()
        per_signal[s["name"]] = ranked[:3]
        if verbose:
            print("%s (nfft %d, %d bins recorded, freq axis %.6g .. %.6g):" % (s["name"], nfft, len(psd), freq[0], freq[-1]))
            for e, key in ranked[:3]:
                print("   %.3g   %s" % (e, key))
    best = sorted((e, k) for k, e in total.items())[:5]
    if verbose:
        print("over all signals (largest error of each convention):")
        for e, key in best:
            print("   %.3g   %s" % (e, key))
    return best, per_signal


def fit_down_converter(observed_dir, probe_dir):
    """The build's two filters (boxcar for the polyphase call, Hamming-sinc otherwise: oracle/spec_oracle.c so_down_convert)
    against JDSP's outputs: length, best gain and delay, residual."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import spec_oracle as so
    meta = json.load(open(os.path.join(probe_dir, "probe.json")))
    for s in meta["signals"]:
        x = load_signal(probe_dir, s["name"])
        for mode, tag in ((0, "dc_fast"), (1, "dc_conv")):
            path = os.path.join(observed_dir, "%s.%s.f64" % (s["name"], tag))
            if not os.path.exists(path):
                continue
            obs = np.fromfile(path, dtype="<f8")
            z = obs[:len(obs) // 2] + 1j * obs[len(obs) // 2:]
            r, i = so.down_convert(x.real, x.imag, meta["freq_off"], meta["down"], mode)
            mine = r + 1j * i
            n = min(len(z), len(mine))
            best = None
            for d in range(-8, 9):                                   # delay in output samples
                a, b = (z[d:n], mine[:n - d]) if d >= 0 else (z[:n + d], mine[-d:n])
                if len(a) < 8:
                    continue
                g = np.vdot(b, a) / max(np.vdot(b, b).real, 1e-300)  # least-squares complex gain
                res = np.abs(a - g * b).max() / max(np.abs(a).max(), 1e-300)
                if best is None or res < best[0]:
                    best = (res, d, g)
            print("%s %s: JDSP %d samples, build %d; best delay %d outputs, gain %.6g%+.6gj, residual %.3g of the peak"
                  % (s["name"], tag, len(z), len(mine), best[1], best[2].real, best[2].imag, best[0]))


if __name__ == "__main__":
    obs_dir = sys.argv[1]
    probe = sys.argv[sys.argv.index("--probe") + 1] if "--probe" in sys.argv else os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "integration", "java-test", "jdsp-probe")
    fit(obs_dir, probe)
    fit_down_converter(obs_dir, probe)
