"""The SigMF side of the boundary: host logic on CPU, and the path on real .sigmf-meta /
.sigmf-data pairs ("identical SigMF inputs") on the GPU."""
import json
import os

import numpy as np
import pytest

from spectral_analyzer_amd import sigmf


def write_pair(tmp_path, oracle, datatype, n, header=0, dataset=None, seed=3, rate=2.5e6, name="rec"):
    raw = oracle.synth_iq(datatype, seed, 0, n)
    data_name = dataset or (name + ".sigmf-data")
    with open(tmp_path / data_name, "wb") as f:
        f.write(b"\xAB" * header)
        f.write(raw.tobytes())
    meta = {"global": {"core:datatype": datatype, "core:version": "1.0.0", "x:unknown": 1},
            "captures": [{"core:sample_start": 0}], "annotations": []}
    if rate is not None:
        meta["global"]["core:sample_rate"] = rate
    if header:
        meta["captures"][0]["core:header_bytes"] = header
    if dataset:
        meta["global"]["core:dataset"] = dataset
    p = tmp_path / (name + ".sigmf-meta")
    p.write_text(json.dumps(meta))
    return str(p), raw


def test_load_rules(tmp_path, oracle):
    p, raw = write_pair(tmp_path, oracle, "ci16_be", 1000)
    r = sigmf.load(p)
    assert r.datatype == "ci16_be" and r.big_endian and r.sample_rate == 2.5e6 and r.header_bytes == 0
    assert r.bytes_per_sample == 4 and r.total_samples == 1000 and np.array_equal(np.asarray(r.buffer), raw)
    # header skip (SMH:60-67) + non-conforming dataset named in the meta file (SMH:49-53)
    p, raw = write_pair(tmp_path, oracle, "cf32_le", 500, header=44, dataset="capture.wav", name="b")
    r = sigmf.load(p)
    assert r.data_path.endswith("capture.wav") and r.header_bytes == 44 and not r.big_endian
    assert np.array_equal(np.asarray(r.buffer), raw)
    # sample-rate default (Global.java:40-42)
    p, _ = write_pair(tmp_path, oracle, "cu8", 10, rate=None, name="c")
    assert sigmf.load(p).sample_rate == 1e6
    bad = tmp_path / "bad.sigmf-meta"
    bad.write_text(json.dumps({"global": {}}))
    with pytest.raises(ValueError):
        sigmf.load(str(bad))


@pytest.mark.gpu
@pytest.mark.parametrize("datatype", ["cf32_le", "ci16_be", "cu8", "cf64_le"])
def test_sigmf_file_end_to_end(tmp_path, oracle, svc, datatype):
    from test_gpu_parity import check_fp32, check_fp64
    nfft, canvas_w, header = 1024, 40, 128
    p, raw = write_pair(tmp_path, oracle, datatype, 30 * nfft + 17, header=header)
    rec = sigmf.load(p)
    off = 5 * nfft                                                 # currentSampleOffset (scroll bar)
    got = rec.waterfall(svc, off, nfft, canvas_w)                  # MC:980-999: hop = fftSize
    ref = oracle.waterfall(raw, off * rec.bytes_per_sample, datatype, nfft, nfft, canvas_w)
    valid = oracle.count_lines(raw.size, off * rec.bytes_per_sample, datatype, nfft, nfft)
    assert valid == 25 and np.all(got[valid:] == -150.0)
    check_fp32(got[:valid], ref[:valid], nfft)
    one = rec.compute_magnitudes(svc, off * rec.bytes_per_sample, nfft)
    check_fp64(one[None, :], oracle.compute_magnitudes(raw, off * rec.bytes_per_sample, nfft, datatype, cf64_decode=True)[None, :])
