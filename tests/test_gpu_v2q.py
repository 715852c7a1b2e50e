"""65536-point fp32 lines by PAIRS of single-workgroup kernels (spec_k_v2q.hip, round 5): the top of the reference's NFFT
slider (main-scene.fxml:129-132) around SpectralService.java:33-85.  A radix-4 decimation-in-frequency step in registers on
the way in; the even workgroup of a pair transforms y0 / y2 (bins 4k, 4k + 2), the odd one y1 / y3 (bins 4k + 1, 4k + 3), two
16384-point transforms of the packed family each; nothing waits for anything.  Until round 4 these lines took the four-step
team kernel (sixteen + sixteen workgroups per line, intermediate handed over in L2).

Against the oracle on the same bytes (every format, either byte order, the reference's own hop, 50 % overlap, odd hops,
window, power output, lines past the end, runs of several lines per pair, a run count that leaves the last group of
sixteen workgroups half empty), against the four-step path it replaces, and -- every value of a full-size output --
against the two-launch path on the device."""
import numpy as np
import pytest

import spectral_analyzer_amd as sa
from test_gpu_parity import check_fp32

pytestmark = pytest.mark.gpu

NFFT = 65536


@pytest.mark.parametrize("datatype", ["cf32_le", "cf32_be", "ci16_le", "ci16_be", "cu8", "ci8"])
@pytest.mark.parametrize("hop,window", [(32768, sa.WIN_RECT), (65536, sa.WIN_RECT), (32768, sa.WIN_HANN), (12345, sa.WIN_RECT),
                                        (16384, sa.WIN_HANN), (70000, sa.WIN_RECT)])
def test_paired_lines_match_oracle(svc, oracle, datatype, hop, window):
    import torch
    assert svc.get_option("large_pair") == 1 and svc.get_option("large_team") == 1     # the default dispatch
    n_lines = 11
    iq = oracle.synth_iq(datatype, seed=hop + window, first_sample=3, n_samples=(n_lines - 1) * hop + NFFT)
    ref = oracle.waterfall(iq, 0, datatype, NFFT, hop, n_lines + 2, window=window)      # two lines past the end
    d = torch.from_numpy(iq).cuda()
    try:
        for lpw in (0, 4):                                 # one line per pair; runs of four (tail run of three)
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
    hop, n_lines, start = 32768, 5, 5 * bps
    iq = oracle.synth_iq(datatype, 21, 0, 5 + (n_lines - 1) * hop + NFFT)
    p = svc.compute_waterfall(iq, start, NFFT, datatype, n_lines, hop=hop, out_fmt=sa.OUT_POW_F32).astype(np.float64)
    p_ref = oracle.waterfall(iq, start, datatype, NFFT, hop, n_lines, power=True)
    assert np.abs(p - p_ref).max() <= 2e-6 * p_ref.max()


def test_one_line_and_odd_run_counts(svc, oracle):
    """A single line (one pair, fourteen idle workgroups in its group of sixteen) and 9 / 17 runs of one line (a second group
    with one pair): the grid's rounding to sixteen leaves no line out and writes none twice."""
    datatype, hop = "ci16_le", 32768
    try:
        svc.set_option("lines_per_wg", 1)
        for n_lines in (1, 9, 17):
            iq = oracle.synth_iq(datatype, 40 + n_lines, 0, (n_lines - 1) * hop + NFFT)
            got = svc.compute_waterfall(iq, 0, NFFT, datatype, n_lines, hop=hop)
            check_fp32(got, oracle.waterfall(iq, 0, datatype, NFFT, hop, n_lines), NFFT)
    finally:
        svc.set_option("lines_per_wg", 0)


@pytest.mark.parametrize("datatype,hop,window", [("cf32_le", 32768, sa.WIN_RECT), ("ci16_le", 32768, sa.WIN_HANN),
                                                 ("cu8", 65536, sa.WIN_RECT)])
def test_paired_and_four_step_paths_agree(svc, oracle, datatype, hop, window):
    """The kernel it replaces in the default dispatch ("large_pair" = 0: the persistent team kernel from 64 lines on) gives
    the same lines to fp32 rounding: different radix plans, same transform."""
    import torch
    n_lines = 130
    iq = svc.synth_iq(datatype, 17, 0, (n_lines - 1) * hop + NFFT)
    try:
        one = svc.compute_waterfall(iq, 0, NFFT, datatype, n_lines, hop=hop, window=window, out_fmt=sa.OUT_POW_F32)
        torch.cuda.synchronize()
        svc.set_option("large_pair", 0)
        four = svc.compute_waterfall(iq, 0, NFFT, datatype, n_lines, hop=hop, window=window, out_fmt=sa.OUT_POW_F32)
        torch.cuda.synchronize()
        rel = ((one - four).abs() / four.amax(dim=1, keepdim=True)).max().item()
        assert rel <= 4e-6, rel
        assert not torch.equal(one, four)                  # (two different kernels did run)
    finally:
        svc.set_option("large_pair", 1)


def test_every_value_at_full_size(svc):
    """2^28 cf32 samples, 8 191 lines of 65536 points at 50 % overlap: all 8 191 x 65 536 power values of the paired kernel
    against the two-launch four-step path ("large_team" = 0) on the device, |dP| <= 4e-6 of the line's peak power; Parseval on
    every 97th line; repeated launches bit-identical (the pair's interleaved stores land where they belong whatever the order)."""
    import torch
    datatype, hop, S = "cf32_le", 32768, 1 << 28
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
        for a in range(0, n_lines, 1024):
            t, w = one[a:a + 1024], two[a:a + 1024]
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
