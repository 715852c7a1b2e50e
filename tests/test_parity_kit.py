"""The parity-pinning kit ("Pinning parity"): the SigMF pairs + expectations under
integration/java-test/fixtures/ that integration/java-test/SpectralServiceParityTest.java feeds to the UNMODIFIED
reference on a machine with a JDK.  Here, without a JVM: the committed files are what tests/golden/export_sigmf.py
writes from the .npz fixtures, they load through the host mirror of the reference's loader (spectral_analyzer_amd/
sigmf.py: header skip, core:dataset, byte order -- SigMfHelper.java:43-94) to the very bytes of the .npz, and the
oracle reproduces every expectation bit for bit from the loaded buffer.  The Java source is checked for the calls it
must make (it cannot be compiled here)."""
import glob
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "integration", "java-test", "fixtures")
GOLD = os.path.join(ROOT, "tests", "golden")


def manifest():
    with open(os.path.join(FIX, "manifest.json")) as f:
        return json.load(f)


def test_manifest_covers_every_spectrogram_fixture():
    m = manifest()
    names = sorted(e["name"] for e in m["fixtures"])
    assert names == sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLD, "wf_*.npz")))
    assert m["tolerance_ulp"] == 4
    kinds = {e["reference"] for e in m["fixtures"]}
    assert kinds == {"lines", "flat-200", "no-window"}
    assert any(e["header_bytes"] for e in m["fixtures"]) and any(not e["data_file"].endswith(".sigmf-data") for e in m["fixtures"])


def test_committed_files_are_what_the_exporter_writes(tmp_path):
    import sys
    sys.path.insert(0, GOLD)
    try:
        import export_sigmf
    finally:
        sys.path.remove(GOLD)
    export_sigmf.export(str(tmp_path))
    fresh = sorted(os.listdir(tmp_path))
    assert fresh == sorted(os.listdir(FIX))
    for name in fresh:
        with open(os.path.join(tmp_path, name), "rb") as a, open(os.path.join(FIX, name), "rb") as b:
            assert a.read() == b.read(), name


@pytest.mark.parametrize("entry", manifest()["fixtures"], ids=lambda e: e["name"])
def test_pairs_reproduce_the_npz_fixtures_bit_for_bit(entry, oracle):
    from spectral_analyzer_amd import sigmf
    z = np.load(os.path.join(GOLD, entry["name"] + ".npz"))
    rec = sigmf.load(os.path.join(FIX, entry["name"] + ".sigmf-meta"))
    assert rec.datatype == entry["datatype"] == str(z["datatype"])
    assert rec.header_bytes == entry["header_bytes"]
    assert os.path.basename(rec.data_path) == entry["data_file"]
    assert np.array_equal(np.asarray(rec.buffer), z["iq"])              # the loader skipped the header, if any
    nfft, hop, lines = entry["nfft"], entry["hop"], entry["lines"]
    exp = np.fromfile(os.path.join(FIX, entry["name"] + ".expected.f64"), dtype="<f8").reshape(lines, nfft)
    assert np.array_equal(exp, z["db"][:lines])
    # the per-slice call the Java test makes: line l at byte l * hop * bps (MC:984-985), the reference's own transform
    bps = entry["bytes_per_sample"]
    assert bps == oracle.bytes_per_sample(entry["datatype"])
    if entry["window"] == 0:
        for l in range(lines):
            got = oracle.compute_magnitudes(np.asarray(rec.buffer), l * hop * bps, nfft, entry["datatype"], cf64_decode=True)
            assert np.array_equal(got, exp[l]), (entry["name"], l)
    if entry["reference"] == "flat-200":                                # what the unmodified reference returns for cf64
        got = oracle.compute_magnitudes(np.asarray(rec.buffer), 0, nfft, entry["datatype"], cf64_decode=False)
        assert np.all(got == -200.0)


def test_java_harness_calls_the_reference_entry_points():
    src = open(os.path.join(ROOT, "integration", "java-test", "SpectralServiceParityTest.java")).read()
    for needle in ("package net.kcundercover.spectral_analyzer;", "new SigMfHelper()", "helper.load(", "getDataBuffer()",
                   "new SpectralService()", "service.computeMagnitudes(buffer, l * hop * bps, nfft, datatype)",
                   "global().getBytesPerSample()", "manifest.json", "Math.ulp(want)", "@TestFactory"):
        assert needle in src, needle
    assert src.count("{") == src.count("}") and src.count("(") == src.count(")")
    assert not re.search(r"\bnative\b", src.split("class SpectralServiceParityTest")[1])   # pure Java: runs against either class


# ---- the JDSP probe (Welch PSD / down-converter semantics: source absent from the reference tree) ---------------------
PROBE = os.path.join(ROOT, "integration", "java-test", "jdsp-probe")


def test_jdsp_probe_inputs_are_what_the_exporter_writes(tmp_path):
    import sys
    sys.path.insert(0, GOLD)
    try:
        import export_jdsp_probe
    finally:
        sys.path.remove(GOLD)
    export_jdsp_probe.export(str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == sorted(os.listdir(PROBE))
    for name in os.listdir(PROBE):
        with open(os.path.join(tmp_path, name), "rb") as a, open(os.path.join(PROBE, name), "rb") as b:
            assert a.read() == b.read(), name
    meta = json.load(open(os.path.join(PROBE, "probe.json")))
    assert {s["name"]: s["nfft"] for s in meta["signals"]} == {"tone8192": 8192, "noise40000": 8192, "burst625": 625, "impulse8192": 8192}


@pytest.mark.parametrize("truth", [
    dict(window="hann", overlap="50 %", detrend="none", scaling="density", output="10 log10", order="fftshifted"),   # this build's default
    dict(window="rect", overlap="0 %", detrend="none", scaling="density", output="linear", order="natural"),   # (rect + spectrum == rect + raw2)
    dict(window="hamming", overlap="75 %", detrend="mean", scaling="raw", output="linear", order="fftshifted")])
def test_fit_tool_recovers_a_known_convention(tmp_path, truth):
    """tools/fit_jdsp.py against recordings SYNTHESISED under a known convention (no JDSP here): the convention comes out
    first, with an error at rounding level, on every probe signal -- so a recording made by the real library will be named."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import fit_jdsp
    finally:
        sys.path.remove(os.path.join(ROOT, "tools"))
    meta = json.load(open(os.path.join(PROBE, "probe.json")))
    for s in meta["signals"]:
        x = fit_jdsp.load_signal(PROBE, s["name"])
        psd = fit_jdsp.evaluate(truth, x, meta["fs"], s["nfft"])
        freq = (np.arange(s["nfft"]) - s["nfft"] // 2) * meta["fs"] / s["nfft"]
        np.concatenate([freq, psd]).astype("<f8").tofile(os.path.join(tmp_path, s["name"] + ".psd.f64"))
    best, per_signal = fit_jdsp.fit(str(tmp_path), PROBE, verbose=False)
    assert json.loads(best[0][1]) == truth and best[0][0] <= 1e-9
    assert best[1][0] > 1e-6                                  # ... and nothing else fits all four signals
    assert set(per_signal) == {s["name"] for s in meta["signals"]}


def test_jdsp_probe_source_makes_the_reference_calls():
    src = open(os.path.join(ROOT, "integration", "java-test", "JdspSemanticsProbe.java")).read()
    for needle in ("import net.kcundercover.jdsp.signal.PowerSpectralDensity;", "import net.kcundercover.jdsp.signal.Resampler;",
                   "PowerSpectralDensity.calculatePsdWelch(new double[][] {re, im}, fs, nfft)",
                   "Resampler.downConvertPolyphase(re, im, freqOff, 1.0, down)", "new Resampler(1, down).downConvert(re, im, freqOff, 1.0)",
                   "observed.json"):
        assert needle in src, needle
    assert src.count("{") == src.count("}") and src.count("(") == src.count(")")
