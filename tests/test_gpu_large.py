"""Lines longer than the LDS holds (fp32 nfft >= 32768, fp64 nfft >= 16384; BASELINE configs[4] is the
65536-point cf64 case): the persistent "team" kernel that keeps the four-step intermediate in each XCD's
L2 (spec_k_team.hip) against the oracle, against the two-launch path, and through the default dispatch."""
import numpy as np
import pytest

import spectral_analyzer_amd as sa
from test_gpu_parity import check_fp32, check_fp64

pytestmark = pytest.mark.gpu


@pytest.fixture
def team(svc):
    """large_team = 2: the team kernel for any number of lines, no fall-back -- a timed-out wait is an error."""
    svc.set_option("large_team", 2)
    yield svc
    svc.set_option("large_team", 1)
    svc.set_option("large_ring", 2)


CASES = [  # datatype, nfft, hop, n_lines, window, fp64 output
    ("cf64_le", 65536, 32768, 5, sa.WIN_RECT, True),      # BASELINE configs[4] shape, a few lines
    ("cf64_le", 65536, 32768, 40, sa.WIN_RECT, True),     # several lines per team: ring reuse, register overlap
    ("cf64_be", 65536, 65536, 3, sa.WIN_RECT, True),      # the reference's hop, byte-swapped decode
    ("cf64_le", 65536, 50000, 4, sa.WIN_HANN, True),      # odd hop, window
    ("cf64_le", 32768, 16384, 21, sa.WIN_RECT, True),     # 128 x 256
    ("cf64_le", 16384, 8192, 150, sa.WIN_RECT, True),     # 128 x 128: teams of 16, many teams
    ("ci16_le", 16384, 4096, 33, sa.WIN_HANN, True),      # fp64 pipeline asked for by the output format
    ("cf32_le", 65536, 32768, 19, sa.WIN_RECT, False),    # fp32 members
    ("cf32_le", 32768, 16384, 70, sa.WIN_RECT, False),
    ("ci16_le", 65536, 65536, 4, sa.WIN_RECT, False),
    ("cu8", 32768, 10000, 9, sa.WIN_HANN, False),
]


@pytest.mark.parametrize("datatype,nfft,hop,n_lines,window,f64", CASES)
def test_team_kernel_matches_oracle(team, oracle, datatype, nfft, hop, n_lines, window, f64):
    import torch
    iq = oracle.synth_iq(datatype, seed=nfft // 1000 + n_lines, first_sample=11, n_samples=(n_lines - 1) * hop + nfft)
    ref = oracle.waterfall(iq, 0, datatype, nfft, hop, n_lines + 2, window=window)   # two lines past the end
    fmt = sa.OUT_DB20_F64 if f64 else sa.OUT_DB20_F32
    got = team.compute_waterfall(torch.from_numpy(iq).cuda(), 0, nfft, datatype, n_lines + 2, hop=hop, window=window,
                                 out_fmt=fmt)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    assert np.all(got[n_lines:] == -150.0)                 # MC:994-998
    if f64:
        check_fp64(got[:n_lines], ref[:n_lines])
    else:
        check_fp32(got[:n_lines], ref[:n_lines], nfft)


@pytest.mark.parametrize("ring", [1, 2, 3, 4])
def test_team_ring_depths(team, oracle, ring):
    import torch
    datatype, nfft, hop, n_lines = "cf64_le", 16384, 8192, 97
    team.set_option("large_ring", ring)
    iq = oracle.synth_iq(datatype, 3, 0, (n_lines - 1) * hop + nfft)
    ref = oracle.waterfall(iq, 0, datatype, nfft, hop, n_lines)
    got = team.compute_waterfall(torch.from_numpy(iq).cuda(), 0, nfft, datatype, n_lines, hop=hop, out_fmt=sa.OUT_DB20_F64)
    torch.cuda.synchronize()
    check_fp64(got.cpu().numpy(), ref)


def test_team_and_two_launch_paths_agree(svc, oracle):
    """The default dispatch (team kernel from 64 lines on, two-launch path below and as the guarded fall-back)
    and the two-launch path alone give the same lines to fp64 rounding; power output as well."""
    import torch
    datatype, nfft, hop, n_lines = "cf64_le", 65536, 32768, 80
    iq = torch.from_numpy(oracle.synth_iq(datatype, 9, 0, (n_lines - 1) * hop + nfft)).cuda()
    try:
        out = {}
        for mode in (0, 1, 2):
            svc.set_option("large_team", mode)
            out[mode] = svc.compute_waterfall(iq, 0, nfft, datatype, n_lines, hop=hop, out_fmt=sa.OUT_POW_F64)
            torch.cuda.synchronize()
        for mode in (1, 2):
            rel = ((out[mode] - out[0]).abs().max() / out[0].abs().max()).item()
            assert rel <= 1e-12, (mode, rel)
        assert torch.equal(out[1], out[2])                 # the fall-back kernels did not run over the result
    finally:
        svc.set_option("large_team", 1)


def test_team_kernel_repeated_launches_are_identical(team, oracle):
    # placement of workgroups may differ from launch to launch; the lines must not
    import torch
    datatype, nfft, hop, n_lines = "cf32_le", 32768, 16384, 200
    iq = team.synth_iq(datatype, 5, 0, (n_lines - 1) * hop + nfft)
    first = team.compute_waterfall(iq, 0, nfft, datatype, n_lines, hop=hop)
    for _ in range(5):
        again = team.compute_waterfall(iq, 0, nfft, datatype, n_lines, hop=hop)
        torch.cuda.synchronize()
        assert torch.equal(first, again)
    sample = [0, 1, 99, 199]
    host = iq.cpu().numpy()
    ref = np.stack([oracle.waterfall(host, l * hop * 8, datatype, nfft, hop, 1)[0] for l in sample])
    check_fp32(first[sample].cpu().numpy(), ref, nfft)
