"""32768-point fp32 lines in ONE workgroup (spec_k_v2h.hip, round 4): the top of the reference's NFFT slider
(main-scene.fxml:129-132) around SpectralService.java:33-85.  A radix-2 decimation-in-frequency step in registers, then
two 16384-point transforms of the packed family through the same LDS buffer; until round 3 these lines took the
four-step team kernel (sixteen + sixteen workgroups per line, intermediate handed over in L2).

Against the oracle on the same bytes (every format, either byte order, the register-reuse hop, the reference's own
hop, odd hops, window, power output, lines past the end, several lines per workgroup), against the four-step path it
replaces, and -- every value of a full-size output -- against that path on the device."""
import numpy as np
import pytest

import spectral_analyzer_amd as sa
from test_gpu_parity import check_fp32

pytestmark = pytest.mark.gpu

NFFT = 32768
SMALL_DEFAULT = 2    # "small_single": the default rule


@pytest.mark.parametrize("datatype", ["cf32_le", "cf32_be", "ci16_le", "ci16_be", "cu8", "ci8"])
@pytest.mark.parametrize("hop,window", [(16384, sa.WIN_RECT), (32768, sa.WIN_RECT), (16384, sa.WIN_HANN), (12345, sa.WIN_RECT),
                                        (8192, sa.WIN_HANN), (40000, sa.WIN_RECT)])
def test_single_workgroup_lines_match_oracle(svc, oracle, datatype, hop, window):
    import torch
    assert svc.get_option("large_single") == 1 and svc.get_option("large_team") == 1    # the default dispatch
    n_lines = 11
    iq = oracle.synth_iq(datatype, seed=hop + window, first_sample=3, n_samples=(n_lines - 1) * hop + NFFT)
    ref = oracle.waterfall(iq, 0, datatype, NFFT, hop, n_lines + 2, window=window)      # two lines past the end
    d = torch.from_numpy(iq).cuda()
    try:
        for lpw in (0, 4):                                 # one line per workgroup; runs of four (tail run of three)
            svc.set_option("lines_per_wg", lpw)
            got = svc.compute_waterfall(d, 0, NFFT, datatype, n_lines + 2, hop=hop, window=window)
            torch.cuda.synchronize()
            got = got.cpu().numpy()
            assert np.all(got[n_lines:] == -150.0)         # MC:994-998
            check_fp32(got[:n_lines], ref[:n_lines], NFFT)
    finally:
        svc.set_option("lines_per_wg", 0)


@pytest.mark.parametrize("datatype", ["cf32_le", "ci16_le", "cu8"])
def test_power_output_start_byte_and_host_buffer(svc, oracle, datatype):
    bps = oracle.bytes_per_sample(datatype)
    hop, n_lines, start = 16384, 6, 5 * bps
    iq = oracle.synth_iq(datatype, 21, 0, 5 + (n_lines - 1) * hop + NFFT)
    p = svc.compute_waterfall(iq, start, NFFT, datatype, n_lines, hop=hop, out_fmt=sa.OUT_POW_F32).astype(np.float64)
    p_ref = oracle.waterfall(iq, start, datatype, NFFT, hop, n_lines, power=True)
    assert np.abs(p - p_ref).max() <= 2e-6 * p_ref.max()


@pytest.mark.parametrize("datatype,hop,window", [("cf32_le", 16384, sa.WIN_RECT), ("ci16_le", 16384, sa.WIN_HANN),
                                                 ("cu8", 32768, sa.WIN_RECT)])
def test_single_workgroup_and_four_step_paths_agree(svc, oracle, datatype, hop, window):
    """The kernel it replaces in the default dispatch ("large_single" = 0: the persistent team kernel from 64 lines on) gives
    the same lines to fp32 rounding: different radix plans, same transform."""
    import torch
    n_lines = 130
    iq = svc.synth_iq(datatype, 17, 0, (n_lines - 1) * hop + NFFT)
    try:
        one = svc.compute_waterfall(iq, 0, NFFT, datatype, n_lines, hop=hop, window=window, out_fmt=sa.OUT_POW_F32)
        torch.cuda.synchronize()
        svc.set_option("large_single", 0)
        four = svc.compute_waterfall(iq, 0, NFFT, datatype, n_lines, hop=hop, window=window, out_fmt=sa.OUT_POW_F32)
        torch.cuda.synchronize()
        rel = ((one - four).abs() / four.amax(dim=1, keepdim=True)).max().item()
        assert rel <= 4e-6, rel
        assert not torch.equal(one, four)                  # (two different kernels did run)
    finally:
        svc.set_option("large_single", 1)


def test_every_value_at_full_size(svc):
    """2^28 cf32 samples, 16 383 lines of 32768 points at 50 % overlap: all 16 383 x 32 768 power values of the
    single-workgroup kernel against the two-launch four-step path ("large_team" = 0) on the device, |dP| <= 4e-6 of the
    line's peak power; Parseval on every 97th line; repeated launches bit-identical."""
    import torch
    datatype, hop, S = "cf32_le", 16384, 1 << 28
    n_lines = (S - NFFT) // hop + 1
    iq = svc.synth_iq(datatype, 0x5EC7A11A, 0, S)
    try:
        one = svc.compute_waterfall(iq, 0, NFFT, datatype, n_lines, hop=hop, out_fmt=sa.OUT_POW_F32)
        torch.cuda.synchronize()
        again = svc.compute_waterfall(iq, 0, NFFT, datatype, n_lines, hop=hop, out_fmt=sa.OUT_POW_F32)
        torch.cuda.synchronize()
        assert torch.equal(one, again)
        del again
        svc.set_option("large_team", 0)
        two = svc.compute_waterfall(iq, 0, NFFT, datatype, n_lines, hop=hop, out_fmt=sa.OUT_POW_F32)
        torch.cuda.synchronize()
        worst, bad = 0.0, 0
        for a in range(0, n_lines, 2048):
            t, w = one[a:a + 2048], two[a:a + 2048]
            rel = ((t - w).abs() / w.amax(dim=1, keepdim=True)).amax(dim=1)
            worst = max(worst, float(rel.max()))
            bad += int((rel > 4e-6).sum())
        assert bad == 0 and worst <= 4e-6, (bad, worst)
        assert bool(torch.isfinite(one).all())
        x = iq.view(torch.float32).view(-1, 2).double()
        for ln in range(0, n_lines, 97):                   # sum |X|^2 = N sum |x|^2
            e = float((x[ln * hop:ln * hop + NFFT] ** 2).sum()) * NFFT
            assert float(one[ln].double().sum()) == pytest.approx(e, rel=2e-5)
    finally:
        svc.set_option("large_team", 1)
        del iq
        torch.cuda.empty_cache()


@pytest.mark.parametrize("datatype", ["cf32_le", "cf32_be", "ci16_le", "cu8", "ci8"])
@pytest.mark.parametrize("hop_num,window", [(2, sa.WIN_RECT), (4, sa.WIN_HANN), (1, sa.WIN_RECT), (0, sa.WIN_HANN)])
@pytest.mark.parametrize("nfft,knob", [(16384, "mid_single"), (8192, "small_single")])
def test_half_line_kernel_at_16384_and_8192_points(svc, oracle, datatype, hop_num, window, nfft, knob):
    """The same kernel one and two sizes down (256-thread workgroups, two / three per CU; the knob = 1 forces it, 2 takes it
    where it was measured faster than the family's kernel of that size): against the oracle, and the family's kernel
    (knob = 0) agrees to fp32 rounding."""
    import torch
    n_lines = 10
    hop = hop_num * nfft // 4 if hop_num else 5000
    iq = oracle.synth_iq(datatype, seed=hop + window + 1, first_sample=2, n_samples=(n_lines - 1) * hop + nfft)
    ref = oracle.waterfall(iq, 0, datatype, nfft, hop, n_lines + 1, window=window)
    d = torch.from_numpy(iq).cuda()
    try:
        out = {}
        for mode in (1, 0, 2):
            svc.set_option(knob, mode)
            got = svc.compute_waterfall(d, 0, nfft, datatype, n_lines + 1, hop=hop, window=window)
            torch.cuda.synchronize()
            out[mode] = got.cpu().numpy()
            assert np.all(out[mode][n_lines:] == -150.0)
            check_fp32(out[mode][:n_lines], ref[:n_lines], nfft)
        assert not np.array_equal(out[1], out[0])                       # two different kernels
        assert np.array_equal(out[2], out[1]) or np.array_equal(out[2], out[0])
    finally:
        svc.set_option(knob, 2 if knob == "mid_single" else SMALL_DEFAULT)
