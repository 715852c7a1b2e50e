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
