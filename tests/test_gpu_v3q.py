"""32768-point fp64 lines by PAIRS of single-workgroup kernels (spec_k_v3h.hip v3q_kernel, round 5): the strict-parity pipeline
(cf64 recordings, double outputs -- SpectralService.java:33-85 computes in double) two sizes above the fp64 family's largest
plan.  A radix-4 decimation-in-frequency step in registers; the first workgroup of a pair transforms y0 / y1 (bins 4k, 4k + 1),
the second y2 / y3, two 8192-point fp64 transforms each; nothing waits for anything.  Before, these lines took the four-step
team kernel.

Against the oracle on the same bytes (every format, either byte order, hops, window, every output format, lines past the end,
runs per pair), against the four-step path it replaces in the default dispatch, and against a long-double DFT
(tests/test_gpu_parity.py::test_fp64_lines_against_a_long_double_dft covers 32768 points through this kernel)."""
import numpy as np
import pytest

import spectral_analyzer_amd as sa
from test_gpu_parity import check_fp64, fp64_pow_tol

pytestmark = pytest.mark.gpu

NFFT = 32768


@pytest.mark.parametrize("datatype", ["cf64_le", "cf64_be", "cf32_le", "cf32_be", "ci16_le", "ci16_be", "cu8", "ci8"])
@pytest.mark.parametrize("hop,window", [(16384, sa.WIN_RECT), (32768, sa.WIN_RECT), (16384, sa.WIN_HANN), (5000, sa.WIN_HANN),
                                        (40000, sa.WIN_RECT)])
def test_paired_fp64_lines_match_oracle(svc, oracle, datatype, hop, window):
    import torch
    assert svc.get_option("large_pair") == 1 and svc.get_option("large_team") == 1    # the default dispatch
    svc.set_option("large_pair", 2)                        # (the pair kernel for little-endian cf64 too: the default leaves that format on the team kernel)
    n_lines = 9
    iq = oracle.synth_iq(datatype, seed=hop + window + 2, first_sample=1, n_samples=(n_lines - 1) * hop + NFFT)
    ref = oracle.waterfall(iq, 0, datatype, NFFT, hop, n_lines + 2, window=window)      # two lines past the end
    d = torch.from_numpy(iq).cuda()
    try:
        for lpw in (0, 4):                                 # one line per pair; runs of four
            svc.set_option("lines_per_wg", lpw)
            got = svc.compute_waterfall(d, 0, NFFT, datatype, n_lines + 2, hop=hop, window=window, out_fmt=sa.OUT_DB20_F64)
            torch.cuda.synchronize()
            got = got.cpu().numpy()
            assert got.dtype == np.float64 and np.all(got[n_lines:] == -150.0)         # MC:994-998
            check_fp64(got[:n_lines], ref[:n_lines])
    finally:
        svc.set_option("lines_per_wg", 0)
        svc.set_option("large_pair", 1)


@pytest.mark.parametrize("datatype", ["cf64_le", "cf32_le", "cu8"])
def test_every_output_format_start_byte_and_host_buffer(svc, oracle, datatype):
    bps = oracle.bytes_per_sample(datatype)
    hop, n_lines, start = 16384, 5, 3 * bps
    svc.set_option("large_pair", 2)
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
    svc.set_option("large_pair", 1)


@pytest.mark.parametrize("datatype,hop,window", [("cf64_le", 16384, sa.WIN_RECT), ("ci16_le", 16384, sa.WIN_HANN),
                                                 ("cf32_le", 32768, sa.WIN_RECT)])
def test_paired_and_four_step_paths_agree(svc, datatype, hop, window):
    """The kernel it replaces in the default dispatch ("large_pair" = 0: the persistent team kernel from 64 lines on) gives the
    same lines to fp64 rounding; 17 runs of one line per pair exercise the half-empty last group of sixteen workgroups."""
    import torch
    n_lines = 130
    iq = svc.synth_iq(datatype, 19, 0, (n_lines - 1) * hop + NFFT)
    try:
        svc.set_option("large_pair", 2)
        one = svc.compute_waterfall(iq, 0, NFFT, datatype, n_lines, hop=hop, window=window, out_fmt=sa.OUT_POW_F64)
        torch.cuda.synchronize()
        svc.set_option("lines_per_wg", 1)
        short = svc.compute_waterfall(iq, 0, NFFT, datatype, 17, hop=hop, window=window, out_fmt=sa.OUT_POW_F64)
        torch.cuda.synchronize()
        assert torch.equal(short, one[:17])
        svc.set_option("lines_per_wg", 0)
        svc.set_option("large_pair", 0)
        four = svc.compute_waterfall(iq, 0, NFFT, datatype, n_lines, hop=hop, window=window, out_fmt=sa.OUT_POW_F64)
        torch.cuda.synchronize()
        rel = ((one - four).abs() / four.amax(dim=1, keepdim=True)).max().item()
        assert rel <= fp64_pow_tol(NFFT), rel
        assert not torch.equal(one, four)                  # (two different kernels did run)
    finally:
        svc.set_option("large_pair", 1)
        svc.set_option("lines_per_wg", 0)
