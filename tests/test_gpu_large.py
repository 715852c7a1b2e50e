"""Lines longer than the LDS holds (fp32 nfft >= 32768, fp64 nfft >= 16384; BASELINE configs[4] is the
65536-point cf64 case): the persistent "team" kernel that keeps the four-step intermediate in each XCD's
L2 (spec_k_team.hip) against the oracle, against the two-launch path, and through the default dispatch."""
import os
import subprocess
import sys

import numpy as np
import pytest

import spectral_analyzer_amd as sa
from test_gpu_parity import check_fp32, check_fp64

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANT = os.environ.get("SPEC_LIB_VARIANT") == "teamvar"


@pytest.fixture(params=[512, 256], ids=["wg512", "wg256"])
def team(svc, request):
    """large_team = 2: the team kernel for any number of lines, no fall-back -- a timed-out wait is an error.
    Both geometries: one 512-thread workgroup per CU (16-wide tiles: the product's), two 256-thread workgroups per CU
    (8-wide tiles, one of either role on every CU: an experiment geometry that only the variant library
    lib/libspecgpu_teamvar.so carries -- test_experiment_geometries_in_the_variant_library runs these cases from it)."""
    if request.param != 512 and not VARIANT:
        pytest.skip("experiment geometry: run from the variant library (test_experiment_geometries_in_the_variant_library)")
    svc.set_option("large_team", 2)
    svc.set_option("large_wg", request.param)
    yield svc
    svc.set_option("large_team", 1)
    svc.set_option("large_ring", 0)
    svc.set_option("large_wg", 0)


def test_experiment_geometries_in_the_variant_library():
    """The product library carries only the team kernel the default dispatch reaches (512-thread workgroups).  The two
    geometries that were measured slower (large_wg = 256 / 1024) are compiled into lib/libspecgpu_teamvar.so
    (-DSPEC_TEAM_VARIANTS, built by __graft_entry__.build()); this test runs the wg256 cases of this file against that
    library in ONE child process, and checks that the product library refuses the knob instead of silently taking the
    default."""
    lib = os.path.join(ROOT, "spectral_analyzer_amd", "lib", "libspecgpu_teamvar.so")
    if VARIANT:
        pytest.skip("already inside the variant run")
    assert os.path.exists(lib), "variant library missing: python -m spectral_analyzer_amd.build --variant teamvar"
    env = dict(os.environ, SPEC_LIB_VARIANT="teamvar")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k", "wg256",
                        "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1].split("passed")[0]


def test_product_library_refuses_the_experiment_geometries(svc, oracle):
    import torch
    if VARIANT:
        pytest.skip("variant library")
    iq = torch.from_numpy(oracle.synth_iq("cf32_le", 1, 0, 65 * 16384 + 65536)).cuda()
    try:
        svc.set_option("large_wg", 256)
        svc.set_option("large_pair", 0)     # (fp32 lines of 32768 / 65536 points no longer take the team kernel by default)
        with pytest.raises(NotImplementedError, match="experiment geometry"):
            svc.compute_waterfall(iq, 0, 65536, "cf32_le", 65, hop=16384)
    finally:
        svc.set_option("large_wg", 512)
        svc.set_option("large_pair", 1)


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
    ("cf32_le", 32768, 12345, 9, sa.WIN_RECT, False),     # odd hop: 16-byte requests at 8-byte alignment
    ("cf32_le", 65536, 65536, 5, sa.WIN_HANN, False),     # window: plain column side, pipelined row side
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


@pytest.mark.parametrize("datatype,nfft,hop,n_lines,window,fmt", [
    ("cf64_le", 65536, 32768, 300, sa.WIN_RECT, sa.OUT_DB20_F64), ("cf64_be", 16384, 5000, 70, sa.WIN_HANN, sa.OUT_POW_F64),
    ("cf32_le", 65536, 32768, 257, sa.WIN_RECT, sa.OUT_DB20_F32), ("ci16_le", 32768, 32768, 64, sa.WIN_HANN, sa.OUT_POW_F32),
    ("cf64_le", 32768, 16384, 70, sa.WIN_RECT, sa.OUT_DB20_F64)])   # large_solo<double, 7, 8> (round-3 advisor)
def test_guarded_fallback_is_one_self_contained_launch(svc, oracle, datatype, nfft, hop, n_lines, window, fmt):
    """Behind the persistent team kernel sits ONE guarded launch (large_solo_kernel: every workgroup owns whole lines
    and an intermediate of its own, nothing waits for another workgroup) instead of round 2's launch pair per 1 GiB
    chunk.  "large_team" = 3 runs it as if the team kernel had timed out: its lines are the two-launch path's lines
    BIT FOR BIT (the same sub-transforms and inter-step twiddles), more lines than workgroups included; and with the
    default dispatch it leaves the team kernel's result alone (its guard reads the abort word, which stays 0)."""
    import torch
    iq = torch.from_numpy(oracle.synth_iq(datatype, 31, 3, (n_lines - 1) * hop + nfft)).cuda()
    svc.set_option("large_single", 0)      # (32768-point fp32 lines: the four-step paths are what this test is about)
    svc.set_option("large_pair", 0)        # (65536-point fp32 lines likewise)
    try:
        out = {}
        for mode in (0, 3, 2, 1):
            svc.set_option("large_team", mode)
            out[mode] = svc.compute_waterfall(iq, 0, nfft, datatype, n_lines + 1, hop=hop, window=window, out_fmt=fmt).clone()
            torch.cuda.synchronize()
        assert torch.equal(out[3], out[0])
        assert torch.equal(out[1], out[2])
        assert bool((out[3][n_lines] == -150.0).all())
    finally:
        svc.set_option("large_team", 1)
        svc.set_option("large_single", 1)
        svc.set_option("large_pair", 1)


def test_a_context_stops_trying_the_team_kernel_after_one_abort(oracle):
    """The default mode never waits for the team kernel's abort word (the call stays asynchronous; the guarded fall-back
    repairs the result).  The context looks at the PREVIOUS call's word at its next large-N call, once its stream is
    idle -- or in spec_sync, where the host waits anyway -- and after one abort takes the two-launch path instead of spinning
    to the 2 s limit in every call; setting the
    "large_team" knob re-arms the team kernel.  The abort is simulated ("large_team_fake_abort")."""
    import torch
    s = sa.SpectralService(0, stream=torch.cuda.Stream().cuda_stream)
    try:
        datatype, nfft, hop, n_lines = "cf64_le", 16384, 8192, 70
        iq = torch.from_numpy(oracle.synth_iq(datatype, 13, 0, (n_lines - 1) * hop + nfft)).cuda()
        s.set_option("large_single", 0)                    # (16384-point fp64 lines: the four-step paths are what this is about)

        def run():   # (the context has a stream of its own: wait for it before anything on torch's stream reads the tile)
            o = s.compute_waterfall(iq, 0, nfft, datatype, n_lines, hop=hop, out_fmt=sa.OUT_POW_F64)
            s.synchronize()
            return o.clone()
        def same(a, b):
            return float((a - b).abs().max() / b.abs().max()) <= 1e-12
        s.set_option("large_team", 0); two = run()
        s.set_option("large_team", 1)
        assert same(run(), two) and s.get_option("large_team_disabled") == 0    # default mode: the team kernel, no complaint
        assert same(run(), two) and s.get_option("large_team_disabled") == 0
        s.set_option("large_team_fake_abort", 1)
        o = s.compute_waterfall(iq, 0, nfft, datatype, n_lines, hop=hop, out_fmt=sa.OUT_POW_F64)   # "timed out": asynchronous, ...
        assert s.get_option("large_team_disabled") == 0    # ... nobody has looked at the abort word yet
        s.synchronize()                                    # spec_sync does (round 4: wherever the host synchronises anyway)
        assert same(o.clone(), two)                        # the guarded fall-back wrote the lines
        assert s.get_option("large_team_disabled") == 1    # two-launch path from now on
        s.set_option("large_team", 1)                      # re-armed; this time the word is found at the NEXT call's entry:
        s.set_option("large_team_fake_abort", 1)
        o = s.compute_waterfall(iq, 0, nfft, datatype, n_lines, hop=hop, out_fmt=sa.OUT_POW_F64)
        torch.cuda.synchronize()                           # (the device is idle, the context has not been asked)
        assert s.get_option("large_team_disabled") == 0
        assert same(run(), two)                            # looked at here, at the next large-N call's entry
        assert s.get_option("large_team_disabled") == 1
        assert same(run(), two) and s.get_option("large_team_disabled") == 1
        s.set_option("large_team", 1)                      # the knob re-arms the persistent launch
        assert s.get_option("large_team_disabled") == 0 and same(run(), two)
        with pytest.raises(ValueError):
            s.get_option("no_such_knob")
    finally:
        s.close()


# ---- team kernel vs two-launch path, EVERY value of a full-size output (the hand-off between workgroups rests on the
# hardware's L2 behaviour: a stale tile would be a rare wrong 16-column stripe, which sampled checks miss) -------------
@pytest.mark.parametrize("datatype,fmt,tol", [("cf64_le", sa.OUT_POW_F64, 1e-12), ("cf32_le", sa.OUT_POW_F32, 4e-6)])
def test_team_and_two_launch_agree_on_every_value_at_full_size(svc, datatype, fmt, tol):
    """BASELINE configs[4] at its per-GPU shape (65536-point cf64 -> f64, hop 32768, 2^30 samples, 32 767 lines) and
    the same shape in cf32: the whole output of the team kernel ("large_team" = 2: no fall-back, a timed-out wait is an
    error) against the whole output of the two-launch path, all 32 767 x 65 536 values compared on the device.
    |dP| <= tol * (the line's peak power): fp64 1e-12; fp32 4e-6 (the two paths use different radix plans -- 8 x 8 x 4
    against 16 x 16 -- so they differ by fp32 rounding, a stale stripe would differ by the stripe)."""
    import torch
    nfft, hop, S = 65536, 32768, 1 << 30
    n_lines = (S - nfft) // hop + 1
    iq = svc.synth_iq(datatype, 0x5EC7A11A, 0, S)
    try:
        svc.set_option("large_team", 2)
        team = svc.compute_waterfall(iq, 0, nfft, datatype, n_lines, hop=hop, out_fmt=fmt)
        torch.cuda.synchronize()
        svc.set_option("large_team", 0)
        two = svc.compute_waterfall(iq, 0, nfft, datatype, n_lines, hop=hop, out_fmt=fmt)
        torch.cuda.synchronize()
        worst, bad_lines = 0.0, 0
        for a in range(0, n_lines, 1024):                        # 1024 lines (up to 512 MiB) at a time
            t, w = team[a:a + 1024], two[a:a + 1024]
            peak = w.amax(dim=1, keepdim=True)
            rel = ((t - w).abs() / peak).amax(dim=1)
            worst = max(worst, float(rel.max()))
            bad_lines += int((rel > tol).sum())
            del t, w, peak, rel
        assert bad_lines == 0 and worst <= tol, (bad_lines, worst)
        assert bool(torch.isfinite(team).all())
    finally:
        svc.set_option("large_team", 1)
        del iq
        torch.cuda.empty_cache()


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


# ---- BASELINE configs[4] at its real per-GPU shape: 65536-pt cf64 -> f64, hop 32768, 2^30 samples ---------
@pytest.mark.parametrize("mode", [1, 0])
def test_cfg5_full_size_properties(svc, oracle, mode):
    """16 GiB in, 16 GiB out, 32 767 lines (+ 3 past the end): default dispatch (team kernel with the guarded
    fall-back behind it) and the two-launch path, which at this size crosses its 1024-line chunk loop."""
    import torch
    datatype, nfft, hop, S = "cf64_le", 65536, 32768, 1 << 30
    bps = 16
    n_lines = (S - nfft) // hop + 1
    svc.set_option("large_team", mode)
    try:
        iq = svc.synth_iq(datatype, 0x5EC7A11A, 0, S)
        out = svc.compute_waterfall(iq, 0, nfft, datatype, n_lines + 3, hop=hop, out_fmt=sa.OUT_DB20_F64)
        torch.cuda.synchronize()
        assert out.shape == (n_lines + 3, nfft) and out.dtype == torch.float64
        assert bool((out[n_lines:] == -150.0).all())                       # EOF lines (MC:994-998)
        assert bool(torch.isfinite(out).all())
        peak = out[:n_lines].argmax(dim=1)                                  # the 0.123 cycles/sample tone
        k = int(round(0.123 * nfft)) + nfft // 2
        assert bool(((peak - k).abs() <= 1).all())
        rng = np.random.default_rng(2)
        picks = [0, 1, 1023, 1024, 1025, n_lines - 1] + [int(x) for x in rng.integers(0, n_lines, 4)]
        for ln in picks:                                                    # chunk seams of the two-launch path included
            raw = iq[ln * hop * bps:(ln * hop + nfft) * bps].cpu().numpy()
            ref = oracle.waterfall(raw, 0, datatype, nfft, hop, 1)
            check_fp64(out[ln].cpu().numpy()[None, :], ref)
        # Parseval, every line: sum |X|^2 = N sum |x|^2 (fp64: 1e-12)
        lines = torch.arange(0, n_lines, 997, device=out.device)
        mag2 = torch.pow(10.0, out[lines] / 10.0).sum(dim=1)
        x = iq.view(torch.float64).view(-1, 2)
        for j, ln in enumerate(lines.tolist()):
            e = float((x[ln * hop:ln * hop + nfft] ** 2).sum()) * nfft
            assert float(mag2[j]) == pytest.approx(e, rel=1e-9)             # 20 log10(|X| + 1e-10): tiny bins bias
        del out, iq
        torch.cuda.empty_cache()
    finally:
        svc.set_option("large_team", 1)


# ---- BASELINE configs[3] at its real shape: Welch 16384-pt, Hann, hop 4096, 256 segments, batched ------------
def test_cfg4_full_size_properties(svc, oracle):
    import torch
    datatype, nfft, hop, n_seg, n_psd, fs = "cf32_le", 16384, 4096, 256, 64, 2.0e6
    per = (n_seg - 1) * hop + nfft                                          # 1 060 864 samples per PSD
    iq = svc.synth_iq(datatype, 0x5EC7A11A, 0, per * n_psd)
    f, p = svc.welch_psd(iq, 0, datatype, fs, nfft=nfft, hop=hop, n_seg=n_seg, window=sa.WIN_HANN, n_psd=n_psd,
                         psd_stride_bytes=per * 8)
    torch.cuda.synchronize()
    assert p.shape == (n_psd, nfft) and bool(torch.isfinite(p).all()) and bool((p > 0).all())
    assert np.allclose(f, (np.arange(nfft) - nfft // 2) * fs / nfft)
    for b in (0, 31, n_psd - 1):                                            # sampled PSDs against the oracle
        raw = iq[b * per * 8:(b + 1) * per * 8].cpu().numpy()
        _, ref = oracle.welch_psd(raw, 0, datatype, nfft, hop, n_seg, oracle.WIN_HANN, oracle.PSD_DENSITY, fs)
        assert np.abs(p[b].cpu().numpy() - ref).max() <= 5e-6 * ref.max()
    # Parseval for EVERY PSD: sum_k P[k] fs sum(w^2) = N mean_seg sum_n |w x|^2
    w = torch.from_numpy(oracle.np_window(nfft, oracle.WIN_HANN)).to(iq.device)
    s2 = float((w * w).sum())
    x = iq.view(torch.float32).view(n_psd, per, 2).double()
    for b in range(n_psd):
        seg = x[b].unfold(0, nfft, hop)                                     # [n_seg, 2, nfft] view
        e = float(((seg * w) ** 2).sum()) / n_seg * nfft
        assert float(p[b].double().sum()) * fs * s2 == pytest.approx(e, rel=2e-5)
    # one PSD alone equals the same PSD inside the batch (batching changes the run lengths, not the numbers'
    # tolerance), and the dB form
    _, one = svc.welch_psd(iq, 17 * per * 8, datatype, fs, nfft=nfft, hop=hop, n_seg=n_seg, window=sa.WIN_HANN)
    torch.cuda.synchronize()
    assert float((one[0] - p[17]).abs().max() / p[17].max()) <= 2e-6
    del iq, p, x
    torch.cuda.empty_cache()
