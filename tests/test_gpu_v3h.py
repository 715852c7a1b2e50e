"""16384-point fp64 lines in ONE workgroup (spec_k_v3h.hip, round 4): the strict-parity pipeline (cf64 recordings, double
outputs -- SpectralService.java:33-85 computes in double) one size above the fp64 family's largest plan.  A radix-2
decimation-in-frequency step in registers, then two 8192-point transforms of the fp64 family through the same LDS buffer;
before, these lines took the four-step team kernel.

Against the oracle on the same bytes (every format, either byte order, the register-reuse hop, the reference's own hop, odd
hops, window, every output format, lines past the end, several lines per workgroup) and against the four-step path it
replaces in the default dispatch."""
import numpy as np
import pytest

import spectral_analyzer_amd as sa
from test_gpu_parity import check_fp64, fp64_pow_tol

pytestmark = pytest.mark.gpu

NFFT = 16384


@pytest.mark.parametrize("datatype", ["cf64_le", "cf64_be", "cf32_le", "cf32_be", "ci16_le", "ci16_be", "cu8", "ci8"])
@pytest.mark.parametrize("hop,window", [(8192, sa.WIN_RECT), (16384, sa.WIN_RECT), (8192, sa.WIN_HANN), (5000, sa.WIN_HANN),
                                        (4096, sa.WIN_RECT), (20000, sa.WIN_RECT)])
def test_single_workgroup_fp64_lines_match_oracle(svc, oracle, datatype, hop, window):
    import torch
    assert svc.get_option("large_single") == 1 and svc.get_option("large_team") == 1    # the default dispatch
    n_lines = 9
    iq = oracle.synth_iq(datatype, seed=hop + window + 2, first_sample=1, n_samples=(n_lines - 1) * hop + NFFT)
    ref = oracle.waterfall(iq, 0, datatype, NFFT, hop, n_lines + 2, window=window)      # two lines past the end
    d = torch.from_numpy(iq).cuda()
    try:
        for lpw in (0, 4):                                 # one line per workgroup; runs of four (tail run of three)
            svc.set_option("lines_per_wg", lpw)
            got = svc.compute_waterfall(d, 0, NFFT, datatype, n_lines + 2, hop=hop, window=window, out_fmt=sa.OUT_DB20_F64)
            torch.cuda.synchronize()
            got = got.cpu().numpy()
            assert got.dtype == np.float64 and np.all(got[n_lines:] == -150.0)         # MC:994-998
            check_fp64(got[:n_lines], ref[:n_lines])
    finally:
        svc.set_option("lines_per_wg", 0)


@pytest.mark.parametrize("datatype", ["cf64_le", "cf32_le", "cu8"])
def test_every_output_format_start_byte_and_host_buffer(svc, oracle, datatype):
    bps = oracle.bytes_per_sample(datatype)
    hop, n_lines, start = 8192, 6, 3 * bps
    iq = oracle.synth_iq(datatype, 23, 0, 3 + (n_lines - 1) * hop + NFFT)
    ref = oracle.waterfall(iq, start, datatype, NFFT, hop, n_lines)
    p_ref = oracle.waterfall(iq, start, datatype, NFFT, hop, n_lines, power=True)
    p = svc.compute_waterfall(iq, start, NFFT, datatype, n_lines, hop=hop, out_fmt=sa.OUT_POW_F64)
    assert p.dtype == np.float64 and np.abs(p - p_ref).max() <= fp64_pow_tol(NFFT) * p_ref.max()
    if datatype.startswith("cf64"):                        # fp64 arithmetic, fp32 storage (cf64 recordings only)
        f32 = svc.compute_waterfall(iq, start, NFFT, datatype, n_lines, hop=hop, out_fmt=sa.OUT_DB20_F32)
        assert f32.dtype == np.float32 and np.abs(f32 - ref).max() <= 2e-5
        p32 = svc.compute_waterfall(iq, start, NFFT, datatype, n_lines, hop=hop, out_fmt=sa.OUT_POW_F32)
        assert p32.dtype == np.float32 and np.abs(p32 - p_ref).max() <= 2e-7 * p_ref.max()


@pytest.mark.parametrize("datatype,hop,window", [("cf64_le", 8192, sa.WIN_RECT), ("ci16_le", 8192, sa.WIN_HANN),
                                                 ("cf32_le", 16384, sa.WIN_RECT)])
def test_single_workgroup_and_four_step_paths_agree(svc, datatype, hop, window):
    """The kernel it replaces in the default dispatch ("large_single" = 0: the persistent team kernel from 64 lines on) gives
    the same lines to fp64 rounding: different radix plans, same transform."""
    import torch
    n_lines = 130
    iq = svc.synth_iq(datatype, 19, 0, (n_lines - 1) * hop + NFFT)
    try:
        one = svc.compute_waterfall(iq, 0, NFFT, datatype, n_lines, hop=hop, window=window, out_fmt=sa.OUT_POW_F64)
        torch.cuda.synchronize()
        svc.set_option("large_single", 0)
        four = svc.compute_waterfall(iq, 0, NFFT, datatype, n_lines, hop=hop, window=window, out_fmt=sa.OUT_POW_F64)
        torch.cuda.synchronize()
        rel = ((one - four).abs() / four.amax(dim=1, keepdim=True)).max().item()
        assert rel <= fp64_pow_tol(NFFT), rel
        assert not torch.equal(one, four)                  # (two different kernels did run)
    finally:
        svc.set_option("large_single", 1)


def test_extreme_magnitudes_take_the_general_form(svc, oracle):
    """Bins outside the table logarithm's range (silence, underflow, |X|^2 beyond 2^996, NaN) in either half."""
    hop, n_lines = NFFT, 4
    x = np.zeros((n_lines * NFFT, 2), np.float64)
    x[0:NFFT] = 0.0                                          # silence: exactly -200 dB
    t = np.arange(NFFT)
    x[NFFT:2 * NFFT, 0] = 1e-160 * np.cos(2 * np.pi * 37 * t / NFFT)      # |X|^2 underflows
    x[2 * NFFT:3 * NFFT, 0] = 1e152 * np.cos(2 * np.pi * 1001 * t / NFFT)  # |X|^2 ~ 1e312: overflows, |X| does not
    x[3 * NFFT:, 0] = 1.0
    x[3 * NFFT + 5, 1] = np.nan
    iq = x.reshape(-1).view(np.uint8)
    ref = oracle.waterfall(iq, 0, "cf64_le", NFFT, hop, n_lines)
    got = svc.compute_waterfall(iq, 0, NFFT, "cf64_le", n_lines, hop=hop, out_fmt=sa.OUT_DB20_F64)
    assert np.all(got[0] == -200.0)
    assert np.all(np.isnan(got[3]))
    assert np.all(np.isfinite(got[1:3]))
    check_fp64(got[1:3], ref[1:3])


def test_every_value_at_full_size(svc):
    """2^27 cf64 samples, 16 383 lines of 16384 points at 50 % overlap: all 16 383 x 16 384 power values of the single-workgroup
    kernel against the two-launch four-step path ("large_single" = 0, "large_team" = 0) on the device, |dP| <= the fp64 power
    tolerance of the line's peak; repeated launches bit-identical (the store hazard of DESIGN.md 4.4d showed up as one run in
    three differing in a few bins); Parseval on every 97th line."""
    import torch
    datatype, hop, S = "cf64_le", 8192, 1 << 27
    n_lines = (S - NFFT) // hop + 1
    iq = svc.synth_iq(datatype, 0x5EC7A11A, 0, S)
    try:
        one = svc.compute_waterfall(iq, 0, NFFT, datatype, n_lines, hop=hop, out_fmt=sa.OUT_POW_F64)
        torch.cuda.synchronize()
        for _ in range(3):
            again = svc.compute_waterfall(iq, 0, NFFT, datatype, n_lines, hop=hop, out_fmt=sa.OUT_POW_F64)
            torch.cuda.synchronize()
            assert torch.equal(one, again)
            del again
        svc.set_option("large_single", 0)
        svc.set_option("large_team", 0)
        two = svc.compute_waterfall(iq, 0, NFFT, datatype, n_lines, hop=hop, out_fmt=sa.OUT_POW_F64)
        torch.cuda.synchronize()
        worst, bad, tol = 0.0, 0, fp64_pow_tol(NFFT)
        for a in range(0, n_lines, 2048):
            t, w = one[a:a + 2048], two[a:a + 2048]
            rel = ((t - w).abs() / w.amax(dim=1, keepdim=True)).amax(dim=1)
            worst = max(worst, float(rel.max()))
            bad += int((rel > tol).sum())
        assert bad == 0 and worst <= tol, (bad, worst)
        assert bool(torch.isfinite(one).all())
        x = iq.view(torch.float64).view(-1, 2)
        for ln in range(0, n_lines, 97):                   # sum |X|^2 = N sum |x|^2
            e = float((x[ln * hop:ln * hop + NFFT] ** 2).sum()) * NFFT
            assert float(one[ln].sum()) == pytest.approx(e, rel=1e-11)
    finally:
        svc.set_option("large_team", 1)
        svc.set_option("large_single", 1)
        del iq
        torch.cuda.empty_cache()
