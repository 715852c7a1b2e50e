#!/usr/bin/env python3
"""Writes the input signals of the JDSP probe (integration/java-test/JdspSemanticsProbe.java): planar little-endian
float64 files <name>.re.f64 / <name>.im.f64 + probe.json under integration/java-test/jdsp-probe/.

The Welch PSD and the down-converter of the reference live in JDSP v1.3.1 (build.gradle:142; call sites
AnalysisDialogController.java:308-312, ExtractDownConvertService.java:106-112), whose source is not in the reference tree:
window, overlap, scaling and filter taps of this build are its own stated specification (parity unpinned).  The probe runs
JDSP itself on these signals on a machine that has it (the reference's own Gradle build does) and records what comes
back; tools/fit_jdsp.py then names the window / overlap / scaling / dB convention that reproduces the recording.

Signals (fs = 1 MHz, seeded):
  tone8192   one segment: a complex exponential exactly on bin 1000 of 8192 -- peak height = scaling, skirt = window
  noise40000 white noise + two tones, 40 000 samples: nfft = 8192 with several segments -- overlap and averaging
  burst625   625 samples: shorter than 8192, so the dialog passes nfft = 625 (ADC:303-307), not a power of two
  impulse8192 a single 1 at sample 3000 of 8192: the flat level is the window's value there
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(os.path.dirname(HERE)), "integration", "java-test", "jdsp-probe")


def signals():
    rng = np.random.default_rng(0x5EC7A11A)
    n = np.arange(8192)
    out = {"tone8192": np.exp(2j * np.pi * 1000 * n / 8192)}
    m = np.arange(40000)
    out["noise40000"] = (0.05 * (rng.standard_normal(40000) + 1j * rng.standard_normal(40000))
                         + 0.5 * np.exp(2j * np.pi * 0.123 * m) + 0.1 * np.exp(-2j * np.pi * 0.31 * m))
    k = np.arange(625)
    out["burst625"] = 0.3 * np.exp(2j * np.pi * 0.07 * k) + 0.02 * (rng.standard_normal(625) + 1j * rng.standard_normal(625))
    imp = np.zeros(8192, complex)
    imp[3000] = 1.0
    out["impulse8192"] = imp
    return out


def export(out_dir: str = OUT):
    os.makedirs(out_dir, exist_ok=True)
    meta = {"fs": 1.0e6, "generated_by": "tests/golden/export_jdsp_probe.py", "down": 8, "freq_off": 0.0731, "signals": []}
    for name, x in signals().items():
        x.real.astype("<f8").tofile(os.path.join(out_dir, name + ".re.f64"))
        x.imag.astype("<f8").tofile(os.path.join(out_dir, name + ".im.f64"))
        meta["signals"].append({"name": name, "samples": int(len(x)), "nfft": int(min(8192, len(x)))})   # ADC:303-313
    with open(os.path.join(out_dir, "probe.json"), "w") as f:
        json.dump(meta, f, indent=1)
        f.write("\n")
    return meta


if __name__ == "__main__":
    print("wrote", len(export()["signals"]), "probe signals to", OUT)
