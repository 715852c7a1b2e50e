#!/usr/bin/env python3
"""bench.py -- spectrogram lines/s on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (IQ bytes resident in HBM -> fftshifted
20 log10|X| lines in HBM) over the rank's shard of the synthetic recording.
Workload at N = 1 is BASELINE config 2: 4096-pt FFT, 50 % overlap, cf32,
2^30 samples (524 287 lines).  With N > 1 every rank holds 2^30 samples of an
N * 2^30-sample recording (time-slice sharding, weak scaling, no collective in
the timed region -- lines are independent).

Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     -- algorithmic bytes / HIP-event kernel time vs 8 TB/s HBM peak
  cpu_baseline -- the CPU oracle (restated reference) timed on the host cores
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (datatype, nfft, hop, log2 samples per GPU, window)
    "cfg2": ("cf32_le", 4096, 2048, 30, 0),   # BASELINE configs[1] -- the metric's config
    "cfg3": ("ci16_le", 4096, 2048, 30, 0),   # BASELINE configs[2] per-GPU share (8 G samples / 8)
    "cfg1": ("cf32_le", 1024, 512, 20, 0),    # BASELINE configs[0] sizes (plumbing case)
    # development workloads (tools/ablate.sh): other line lengths at 50 % overlap
    "n1024": ("cf32_le", 1024, 512, 30, 0),
    "n8192": ("cf32_le", 8192, 4096, 30, 0),
    "n16384": ("cf32_le", 16384, 8192, 30, 0),
}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
SEED = 0x5EC7A11A


def usable_cores() -> int:
    """CPU threads this process may actually run on: affinity mask, capped by a cgroup quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)  # the clock needs ~10 launches to settle
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--log2-samples", type=int, default=None, help="override samples per GPU (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-log2-samples", type=int, default=28, help="CPU baseline sample: first 2^k samples")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    # SPEC_BENCH_REHEARSE=1: rehearsal of the N > 1 code path on a ONE-GPU box -- every rank on device 0,
    # gloo instead of RCCL (which refuses two ranks on one device).  Never used by the driver.
    rehearse = os.environ.get("SPEC_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: F811
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    red_dev = "cpu" if rehearse else None  # device of the two scalar reductions

    import spectral_analyzer_amd as sa
    from spectral_analyzer_amd import _lib as L

    datatype, nfft, hop, log2s, window = WORKLOADS[args.workload]
    if args.log2_samples is not None:
        log2s = args.log2_samples
    bps = sa.bytes_per_sample(datatype)
    per_gpu = 1 << log2s
    total_samples = per_gpu * world
    total_lines = (total_samples - nfft) // hop + 1
    l0, l1 = rank * total_lines // world, (rank + 1) * total_lines // world
    n_lines = l1 - l0
    first_sample = l0 * hop
    n_samples = (n_lines - 1) * hop + nfft  # includes the nfft-hop halo shared with the next rank

    # a real (non-null) stream shared by torch and the library, so that the HIP
    # events below bracket exactly the kernels the C ABI launches
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    svc = sa.SpectralService(local_rank, stream=stream.cuda_stream)
    iq = svc.synth_iq(datatype, SEED, first_sample, n_samples)
    out = torch.empty((n_lines, nfft), dtype=torch.float32, device=iq.device)

    def step():
        svc.compute_waterfall(iq, 0, nfft, datatype, n_lines, hop=hop, window=window,
                              out_fmt=L.OUT_DB20_F32, out=out)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        step()
        b.record(stream)
    fence()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev or iq.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        k = torch.tensor([kern_ms], dtype=torch.float64, device=red_dev or iq.device)
        dist.all_reduce(k, op=dist.ReduceOp.MAX)
        kern_ms = float(k.item())

    # spot check outside the timed region: a few of this rank's lines against the oracle
    checked = None
    cpu_baseline = None
    if rank == 0:
        from oracle import spec_oracle as so
        so.build()
        pick = sorted(set(int(x) for x in np.linspace(0, n_lines - 1, 6)))
        worst = 0.0
        for ln in pick:
            raw = iq[ln * hop * bps:(ln * hop + nfft) * bps].cpu().numpy()
            ref = so.waterfall(raw, 0, datatype, nfft, hop, 1, window)[0]
            got = out[ln].cpu().numpy().astype(np.float64)
            m_ref, m_got = 10 ** (ref / 20), 10 ** (got / 20)
            worst = max(worst, float(np.abs(m_got - m_ref).max() / (m_ref.max() * np.log2(nfft))))
        checked = {"lines": len(pick), "max_lin_err_over_M_log2N": worst, "tol": 4e-6, "ok": bool(worst <= 4e-6)}

        if world == 1 and not args.no_cpu_baseline:
            cs = min(1 << args.cpu_log2_samples, n_samples)
            host = iq[:cs * bps].cpu().numpy()
            cl = (cs - nfft) // hop + 1
            cores = usable_cores()
            secs, _ = so.time_waterfall(host, datatype, nfft, hop, cl, window, cores)
            cpu_baseline = {"value": cl / secs, "unit": "lines/s", "cores": cores, "kind": "port",
                            "sample": "first 2^%d samples of the same recording (%d lines), fp64 C oracle "
                                      "(restated SpectralService.computeMagnitudes), %d pthreads"
                                      % (int(np.log2(cs)), cl, cores)}

    # calibration outside the timed region ("calibrate with a device memcpy"): what a
    # plain device-to-device copy of the output tile moves per second on THIS box, read + write
    copy_gbps = None
    if rank == 0 and n_lines * nfft * 4 >= (1 << 28):
        dst = torch.empty_like(out)
        for _ in range(3):
            dst.copy_(out)
        cev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a, b in cev:
            a.record(stream)
            dst.copy_(out)
            b.record(stream)
        torch.cuda.synchronize()
        copy_gbps = 2.0 * out.numel() * 4 / (float(np.median([a.elapsed_time(b) for a, b in cev])) * 1e-3) / 1e9
        del dst

    if rank == 0:
        lines_all = total_lines * args.steps
        value = lines_all / elapsed
        b_line = hop * bps + nfft * 4  # every sample read once, every bin written once
        achieved = n_lines * b_line / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.workload if log2s == WORKLOADS[args.workload][3] else "")
            except Exception:
                traffic = None
        res = {
            "metric": "spectrogram lines/sec (4096-pt FFT, 50% overlap)" if nfft == 4096 else
                      "spectrogram lines/sec (%d-pt FFT)" % nfft,
            "value": value, "unit": "lines/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d-pt FFT, hop %d, %s, 2^%d samples per GPU, %d lines total, DB20_F32 out"
                                   % (args.workload, nfft, hop, datatype, log2s, total_lines),
                       "window": "rect" if window == 0 else "hann", "sharding": "time-slice x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel_ms": kern_ms, "bytes_per_line": b_line, "lines_per_launch": n_lines,
                         # extras: reads only (the north star is phrased on reads) and the box's own copy rate
                         "read_frac": n_lines * hop * bps / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "copy_GBps": copy_gbps,
                         "frac_of_copy": (achieved / copy_gbps) if copy_gbps else None},
            "cpu_baseline": cpu_baseline,
            "parity_spot_check": checked,
        }
        print(json.dumps(res), flush=True)
    # outside the timed region: the tile gather of SURVEY 8(e) on a bounded slice (the first
    # <= 4096 lines of every rank) through direct peer -> root sends; reported on stderr AFTER the bench line is out, never part of `value`
    if dist is not None:
        try:
            gl = min(n_lines, 4096)
            tile = out[:gl]
            torch.cuda.synchronize()
            dist.barrier()
            g0 = time.perf_counter()
            if rank == 0:
                full = torch.empty((gl * world, nfft), dtype=out.dtype, device=out.device)
                full[:gl].copy_(tile)
                ops = [dist.P2POp(dist.irecv, full[r * gl:(r + 1) * gl], r) for r in range(1, world)]
            else:
                ops = [dist.P2POp(dist.isend, tile, 0)]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            torch.cuda.synchronize()
            dist.barrier()
            gms = (time.perf_counter() - g0) * 1e3
            if rank == 0:
                print(json.dumps({"gather": {"lines_per_rank": gl, "ms": gms,
                                             "GBps_into_root": (world - 1) * gl * nfft * 4 / gms / 1e6}}),
                      file=sys.stderr, flush=True)
        except Exception as e:  # the demonstration must never break the run
            print("gather demonstration failed: %r" % (e,), file=sys.stderr, flush=True)

    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
