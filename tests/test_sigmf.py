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


@pytest.mark.gpu
@pytest.mark.parametrize("dataset", [None, "big capture.raw"])
def test_recording_larger_than_2_gib(tmp_path, oracle, svc, dataset):
    """SigMfHelper.java:78-84 maps at most 2 GiB - 1 bytes and MainController.java:985 casts the byte offset to
    int.  Here: a sparse 5 GiB data file (header 1000 bytes; three written windows, the rest holes), read by
    the library itself through spec_open_recording -- lines at byte offsets beyond 2^31 and 2^32 match the
    oracle, holes are silence (-200 dB, SS:81), lines that run past the end are -150.0 (MC:994-998)."""
    from test_gpu_parity import check_fp32, check_fp64
    datatype, nfft, header = "ci16_le", 4096, 1000
    bps = 4
    total_samples = (5 << 30) // bps + 123                      # not a whole number of lines
    name = dataset or "big.sigmf-data"
    data_path = tmp_path / name
    windows = {}                                                # first sample -> raw bytes
    with open(data_path, "wb") as f:
        f.truncate(header + total_samples * bps)
        for first in (0, (1 << 31) // bps + 77, (1 << 32) // bps + 4096 * 3 + 5, total_samples - 3 * nfft - 10):
            n = 3 * nfft + 10
            raw = oracle.synth_iq(datatype, 100 + len(windows), first, n)
            f.seek(header + first * bps)
            f.write(raw.tobytes())
            windows[first] = raw
    meta = {"global": {"core:datatype": datatype, "core:sample_rate": 1e6}, "captures": [{"core:header_bytes": header}]}
    if dataset:
        meta["global"]["core:dataset"] = dataset
    mp = tmp_path / "big.sigmf-meta"
    mp.write_text(json.dumps(meta))
    rec = sigmf.load(str(mp))
    assert rec.total_samples == total_samples and rec.header_bytes == header
    with rec.open_native(svc) as nat:
        assert nat.n_bytes == total_samples * bps > (1 << 32)
        for first, raw in windows.items():
            got = nat.waterfall(first, nfft, 3)                  # hop = fftSize, as MC:984
            ref = oracle.waterfall(raw, 0, datatype, nfft, nfft, 3)
            check_fp32(got, ref, nfft)
            one = nat.compute_magnitudes(first * bps, nfft)      # a startByte no int can hold (MC:985)
            check_fp64(one[None, :], oracle.compute_magnitudes(raw, 0, nfft, datatype)[None, :])
        hole = nat.waterfall((3 << 30) // bps, nfft, 2)
        assert np.abs(hole + 200.0).max() < 1e-3                 # zeros -> 20 log10(1e-10), in fp32
        tail = nat.waterfall(total_samples - 3 * nfft - 10, nfft, 6)
        assert np.all(tail[3:] == -150.0) and not np.any(tail[:3] == -150.0)
        with pytest.raises(IndexError):
            nat.compute_magnitudes(total_samples * bps - 100, nfft)
        # the chunked pipeline (file -> pinned ring -> device -> host) across the 2^32 boundary: 40 MiB of
        # lines with a 4 MiB staging chunk, compared with the mapped-buffer path of spec_waterfall
        svc.set_option("stage_chunk_mb", 4)
        try:
            start = (1 << 32) // bps - 1200 * nfft
            lines = 2500
            a = nat.waterfall(start, nfft, lines)                # the library's own mapping of the file
            b = rec.waterfall(svc, start, nfft, lines)           # the caller's mapping (numpy.memmap)
            svc.set_option("rec_pread", 1)
            c = nat.waterfall(start, nfft, lines)                # pread into the pinned ring
            one = nat.compute_magnitudes(((1 << 31) // bps + 77) * bps, nfft)
            svc.set_option("rec_pread", 0)
            assert np.array_equal(a, b) and np.array_equal(a, c)
            assert np.array_equal(one, nat.compute_magnitudes(((1 << 31) // bps + 77) * bps, nfft))
            w0 = (1 << 32) // bps + 4096 * 3 + 5                 # a written window lies inside that span
            assert not np.all(a == -200.0) and (w0 - start) // nfft < lines
        finally:
            svc.set_option("stage_chunk_mb", 64)
            svc.set_option("rec_pread", 0)
    os.unlink(data_path)


@pytest.mark.gpu
def test_a_recording_that_shrinks_after_it_was_opened_is_read_with_pread(tmp_path, oracle, svc):
    """The library maps the whole data file.  If the file is truncated behind its back, touching the mapping past the
    new end would be SIGBUS for the host process (a JVM included); the library re-stats before every call and reads with
    pread for that call instead: bytes past the new end read as zero (flat -200 dB lines), nobody dies."""
    nfft, lines = 256, 40
    p, raw = write_pair(tmp_path, oracle, "ci16_le", lines * nfft, name="shrink")
    rec = sigmf.load(p)
    with sigmf.NativeRecording(svc, rec) as h:
        full = h.waterfall(0, nfft, lines)
        os.truncate(rec.data_path, 10 * nfft * 4)                  # 10 lines are left on disk
        cut = h.waterfall(0, nfft, lines)
        assert np.array_equal(cut[:10], full[:10])
        assert np.abs(cut[10:] + 200.0).max() < 1e-3               # zeros -> 20 log10(1e-10) (SS:81) in fp32; the span is still "in range"


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [14, 15])
def test_random_redraws_of_recordings_equal_the_buffer_path(tmp_path, oracle, svc, seed):
    """Random SigMF pairs (datatype, header bytes, length) and random redraws (scroll offset, size, hop, canvas width,
    window, output format): the library's own file reader -- whole-file mapping, and the pread ring of `rec_pread` -- gives
    the very tile of the buffer path on the same bytes (which the other tests pin to the oracle), lines past the end -150;
    the single-line call likewise."""
    rng = np.random.default_rng(seed)
    dtypes = ["cf32_le", "cf32_be", "ci16_le", "ci16_be", "cu8", "ci8", "cf64_le", "cf64_be"]
    for k in range(6):
        dt = str(rng.choice(dtypes))
        n = int(rng.integers(1, 200000))
        header = int(rng.choice([0, 1, 44, int(rng.integers(2, 5000))]))
        p, raw = write_pair(tmp_path, oracle, dt, n, header=header, seed=int(rng.integers(1, 1 << 30)), name="r%d_%d" % (seed, k))
        rec = sigmf.load(p)
        for pread in (0, 1):
            svc.set_option("rec_pread", pread)
            try:
                with rec.open_native(svc) as nat:
                    assert nat.n_bytes == raw.size
                    for _ in range(5):
                        nfft = 1 << int(rng.choice([6, 8, 10, 12, 13]))
                        hop = int(rng.choice([nfft, nfft // 2, int(rng.integers(1, 2 * nfft + 1))]))
                        off = int(rng.integers(0, n + 5))
                        width = int(rng.integers(1, 60))
                        window = int(rng.integers(0, 2))
                        fmt = int(rng.choice([0, 1, 2]))                     # dB f32, |X|^2 f32, dB f64
                        tag = (dt, n, header, pread, nfft, hop, off, width, window, fmt)
                        ref = svc.compute_waterfall(raw, off * rec.bytes_per_sample, nfft, dt, width, hop=hop, window=window, out_fmt=fmt)
                        got = nat.waterfall(off, nfft, width, hop=hop, window=window, out_fmt=fmt)
                        assert got.dtype == ref.dtype and np.array_equal(got, ref), tag
                        if off * rec.bytes_per_sample + nfft * rec.bytes_per_sample <= raw.size:
                            assert np.array_equal(nat.compute_magnitudes(off * rec.bytes_per_sample, nfft),
                                                  svc.compute_magnitudes(raw, off * rec.bytes_per_sample, nfft, dt, rec.big_endian)), tag
            finally:
                svc.set_option("rec_pread", 0)
