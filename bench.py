#!/usr/bin/env python3
"""bench.py -- spectrogram lines/s on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3 [--workload cfg2|cfg3|cfg4|cfg5|cfg1]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (IQ bytes resident in HBM -> results in HBM) over the rank's
shard of the synthetic recording.  Workloads (BASELINE.json configs):
  cfg2  4096-pt FFT, 50 % overlap, cf32, 2^30 samples per GPU (524 287 lines) -- the metric's config and the default
        at EVERY N, so that the per-N values of one scaling series are values of one workload (weak scaling)
  cfg3  4096-pt, ci16 (on-GPU int16 -> float), 2^30 samples per GPU = the per-GPU share of BASELINE's 8 G-sample,
        8-GPU recording (--workload cfg3); at N > 1 either one has the tile gather timed beside the compute-only figure
  cfg4  Welch PSD, 16384-pt, Hann, 75 % overlap, 256 segments per PSD, cf32; 1024 independent PSDs per step
        (one annotation each); a "line" is one segment; VALU-bound, so the roofline is the fp32 vector peak
  cfg5  65536-pt FFT, cf64 -> f64 (the whole pipeline in fp64), 2^30 samples per GPU (32 767 lines)
  cfg1  1024-pt, 1 M samples: the reference's own CPU-runnable case (2 047 lines: launch-bound)
With N > 1 every rank holds its own 2^30 samples of an N * 2^30-sample recording (time-slice sharding,
weak scaling; lines are independent, SURVEY 8e, so `value` has no collective in its timed region).

Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     -- algorithmic bytes (or flops) / HIP-event kernel time vs the peak that bounds the kernel
  cpu_baseline -- the CPU oracle (restated reference) timed on the host cores: all cores, and ONE thread
                  (the reference computes on the single JavaFX thread)
  gather       -- N > 1 only: the same step with every rank's tile sent to rank 0 in chunks on a second
                  stream while the next chunk is computed: compute+gather lines/s
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
    # spectrogram: datatype, nfft, hop, log2 samples per GPU, window, output
    "cfg2": dict(kind="spectro", datatype="cf32_le", nfft=4096, hop=2048, log2s=30, window=0, out="f32"),
    "cfg3": dict(kind="spectro", datatype="ci16_le", nfft=4096, hop=2048, log2s=30, window=0, out="f32"),
    "cfg1": dict(kind="spectro", datatype="cf32_le", nfft=1024, hop=512, log2s=20, window=0, out="f32"),
    "cfg5": dict(kind="spectro", datatype="cf64_le", nfft=65536, hop=32768, log2s=30, window=0, out="f64"),
    # Welch: n_psd independent PSDs of n_seg segments each, per GPU
    "cfg4": dict(kind="welch", datatype="cf32_le", nfft=16384, hop=4096, n_seg=256, n_psd=1024, window=1),
    # development workloads (tools/ablate.sh): other line lengths at 50 % overlap
    "n1024": dict(kind="spectro", datatype="cf32_le", nfft=1024, hop=512, log2s=30, window=0, out="f32"),
    "n8192": dict(kind="spectro", datatype="cf32_le", nfft=8192, hop=4096, log2s=30, window=0, out="f32"),
    "n16384": dict(kind="spectro", datatype="cf32_le", nfft=16384, hop=8192, log2s=30, window=0, out="f32"),
    "n65536f": dict(kind="spectro", datatype="cf32_le", nfft=65536, hop=32768, log2s=30, window=0, out="f32"),
    "n32768f": dict(kind="spectro", datatype="cf32_le", nfft=32768, hop=16384, log2s=30, window=0, out="f32"),
    "n16384d": dict(kind="spectro", datatype="cf64_le", nfft=16384, hop=8192, log2s=30, window=0, out="f64"),
    "n64": dict(kind="spectro", datatype="cf32_le", nfft=64, hop=32, log2s=28, window=0, out="f32"),
    "n128": dict(kind="spectro", datatype="cf32_le", nfft=128, hop=64, log2s=28, window=0, out="f32"),
}
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
GATHER_TIMEOUT_S = 240  # N > 1: the compute + gather phase is abandoned after this long (the headline is kept)
FP32_VECTOR_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 (vector)
SEED = 0x5EC7A11A


class ClockSampler:
    """Shader / memory clock, power and temperature of one GPU, read from sysfs (hwmon of the device's PCI function) by a host
    thread -- no child process, no HIP call, nothing on the stream that is being timed.  VERDICT r04 item 4: every number of the
    bench line carries the clock and power state it was measured at (the 4096-point kernels run at 5+ TB/s and the boxes of the
    pool read 0.63 ... 0.66 for cfg2: a power or thermal cap is part of the roofline story)."""

    FIELDS = (("sclk_mhz", "freq1_input", 1e-6), ("mclk_mhz", "freq2_input", 1e-6), ("power_w", "power1_average", 1e-6),
              ("power_w", "power1_input", 1e-6), ("temp_c", "temp2_input", 1e-3), ("temp_c", "temp1_input", 1e-3))

    def __init__(self, device_index: int):
        self.dir, self.err, self.samples, self._stop, self._thr = None, None, [], False, None
        try:
            p = torch.cuda.get_device_properties(device_index)
            addr = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
            base = os.path.join("/sys/bus/pci/devices", addr, "hwmon")
            hw = sorted(os.listdir(base))
            self.dir = os.path.join(base, hw[0])
            self.files = {}
            for key, name, scale in self.FIELDS:
                f = os.path.join(self.dir, name)
                if key not in self.files and os.path.exists(f):
                    self.files[key] = (f, scale)
            if not self.files:
                raise OSError("no clock / power files under " + self.dir)
            self.pci = addr
        except Exception as e:  # noqa: BLE001 -- a box without readable sysfs still benches; the line says why
            self.err = "%s: %s" % (type(e).__name__, e)

    def read(self):
        out = {}
        for key, (f, scale) in self.files.items():
            try:
                out[key] = float(open(f).read().strip()) * scale
            except (OSError, ValueError):
                pass
        return out

    def start(self, period_s: float = 0.001):
        if self.err:
            return
        import threading
        self.idle = self.read()

        def loop():
            while not self._stop:
                self.samples.append(self.read())
                time.sleep(period_s)
        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()

    def stop(self):
        if self._thr is not None:
            self._stop = True
            self._thr.join()
        if self.err:
            return {"source": "sysfs hwmon", "error": self.err}
        out = {"source": "sysfs hwmon of " + self.pci, "samples": len(self.samples), "idle_before": self.idle}
        try:
            out["power_cap_w"] = float(open(os.path.join(self.dir, "power1_cap")).read()) * 1e-6
        except (OSError, ValueError):
            pass
        for key in self.files:
            v = sorted(x[key] for x in self.samples if key in x)
            if v:
                out[key] = {"min": v[0], "median": v[len(v) // 2], "max": v[-1]}
        return out


def live_traffic(args):
    """HBM-side bytes per launch of THIS workload on THIS box, measured now: two short rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
    the guide's gfx950 corrections) of this same script as CHILD processes, started before this process has made any HIP call
    (roofline.traffic used to be a constant stamped from an earlier profile).  Returns (bytes, note) or
    (None, reason); never raises.  Skipped under a profiler, under torchrun, inside the children themselves and with
    --no-live-traffic (the stamped figure of profiles/pmc_traffic.json is then reported, and says so)."""
    import csv, glob, shutil, subprocess, tempfile
    if os.environ.get("SPEC_BENCH_CHILD") or "RANK" in os.environ or args.gpus != 1:
        return None, "not measured in this process"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "running under a profiler"
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not found"
    me = os.path.abspath(__file__)
    base = [sys.executable, me, "--workload", args.workload, "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    if args.log2_samples is not None:
        base += ["--log2-samples", str(args.log2_samples)]
    if args.n_psd is not None:
        base += ["--n-psd", str(args.n_psd)]
    for o in args.opt:
        base += ["--opt", o]
    env = dict(os.environ, SPEC_BENCH_CHILD="1", TMPDIR="/tmp")
    out = {}
    tmp = tempfile.mkdtemp(prefix="spec_traffic_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            r = subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--", *base], cwd="/tmp", env=env,
                               capture_output=True, text=True, timeout=240)
            files = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s failed (rc %d)" % (counter, r.returncode)
            per = {}
            for row in csv.DictReader(open(files[0])):
                k = row["Kernel_Name"]
                if row["Counter_Name"] == counter and not any(x in k for x in ("synth", "copyBuffer", "fill")):
                    per.setdefault(k, []).append(float(row["Counter_Value"]))
            if not per:
                return None, "no kernel in the %s pass" % counter
            if counter == "FETCH_SIZE":
                out["kernel"] = max(per, key=lambda k: sum(per[k]))       # the dominant kernel: the one that reads most
            v = per.get(out["kernel"]) or max(per.values(), key=sum)
            out[counter] = sum(v) / len(v)
        # MI355X_MICROARCH.md, HBM: both counters in KiB; FETCH_SIZE reads exactly half of a wide coalesced stream on gfx950 -> x2
        return out["FETCH_SIZE"] * 1024 * 2 + out["WRITE_SIZE"] * 1024, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run (3 launches each), kernel " + out["kernel"].replace("(anonymous namespace)::", "").replace("specgpu::", "").split("(")[0][:80]
    except Exception as e:  # noqa: BLE001
        return None, "%s: %s" % (type(e).__name__, e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def stream_probe(args, read_bytes: float, write_bytes: float):
    """What THIS box's memory system delivers for a plain stream with this workload's read : write mix (16-byte coalesced
    accesses, no arithmetic): tools/membench_rw.hip, built by build() into lib/membench_rw and run as a CHILD process before this
    process touches the GPU.  HBM3E does not add its directions: on these boxes a read-only stream moves 7.0 TB/s, a write-only
    one 5.3, 1 : 1 5.5, 1 : 2 (ci16 lines) 4.8, 1 : 4 (cu8) 4.7 -- the ceiling a spectrogram kernel can reach depends on its
    bytes in per byte out.  Returns a dict or None (never raises; skipped like live_traffic)."""
    import subprocess
    if os.environ.get("SPEC_BENCH_CHILD") or "RANK" in os.environ or args.gpus != 1 or args.no_stream_probe:
        return None
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    exe = os.path.join(ROOT, "spectral_analyzer_amd", "lib", "membench_rw")
    if not os.path.exists(exe):
        return None
    ratio = read_bytes / write_bytes if write_bytes > 0 else float("inf")
    mix = (4, 0) if ratio > 6 else (2, 1) if ratio >= 1.5 else (2, 2) if ratio >= 0.75 else (1, 2) if ratio >= 0.375 else (1, 4)
    try:
        r = subprocess.run([exe, str(mix[0]), str(mix[1])], capture_output=True, text=True, timeout=120)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        if "GBps" not in d:
            return None
        d["mix"] = "read %d : write %d (workload: %.2f bytes read per byte written)" % (mix[0], mix[1], ratio) if write_bytes > 0 else "read only"
        d["how"] = "tools/membench_rw.hip as a child process of this run: 4 GiB in, 4 GiB out, 16-byte non-temporal accesses, best of 8 launches"
        return d
    except Exception:  # noqa: BLE001
        return None


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


def cpu_baseline_for(so, host: np.ndarray, datatype: str, nfft: int, hop: int, window: int, lines_all: int,
                     lines_one: int, what: str):
    """The fp64 C oracle (= restated SpectralService.computeMagnitudes, SS:33-85) over the first lines of the
    same recording: every usable core (pthreads over lines) and ONE thread (the reference's FX thread)."""
    cores = usable_cores()
    secs, _ = so.time_waterfall(host, datatype, nfft, hop, lines_all, window, cores)
    secs1, _ = so.time_waterfall(host, datatype, nfft, hop, lines_one, window, 1)
    return {"value": lines_all / secs, "unit": "lines/s", "cores": cores, "kind": "port",
            "single_thread": {"value": lines_one / secs1, "unit": "lines/s", "cores": 1,
                              "sample": "first %d %s" % (lines_one, what)},
            "sample": "first %d %s of the same recording, fp64 C oracle (restated "
                      "SpectralService.computeMagnitudes), %d pthreads; single_thread = the same on one thread, "
                      "as the reference's JavaFX thread" % (lines_all, what, cores)}


def relaunch_under_torchrun(n: int) -> None:
    """`python bench.py --gpus N` started WITHOUT a launcher: start the one-rank-per-GPU job the contract describes
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    ...`) as a CHILD process, relay its output and leave with its exit code.  Called before this process has touched
    the GPU (no HIP call yet: importing torch and counting devices do not initialise it), and it starts a child -- it
    never replaces this process."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and os.environ.get("SPEC_BENCH_REHEARSE") != "1":
        print(json.dumps({"error": "bench.py --gpus %d: this node shows %d GPU(s)" % (n, have), "n_gpus": n}), flush=True)
        sys.exit(2)
    with socket.socket() as sk:  # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    print("bench.py: --gpus %d without a launcher, starting: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    sys.exit(subprocess.call(cmd, env=env))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)  # the clock needs ~10 launches to settle
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: cfg2 (the metric's configuration) at every N; cfg3 = the per-GPU share of BASELINE's 8-GPU recording")
    ap.add_argument("--log2-samples", type=int, default=None, help="override samples per GPU (debug)")
    ap.add_argument("--n-psd", type=int, default=None, help="cfg4: PSDs per GPU and step (default 1024)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stream-probe", action="store_true", help="skip roofline.stream (the box's plain-stream ceiling for the workload's read : write mix)")
    ap.add_argument("--no-live-traffic", action="store_true", help="report the stamped traffic figure instead of measuring it (two rocprofv3 child runs)")
    ap.add_argument("--gather-steps", type=int, default=3, help="N > 1: steps of the compute+gather timing")
    ap.add_argument("--gather-chunks", type=int, default=8)
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="spec_set_option knob for experiments, e.g. --opt large_team=0 (repeatable)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            relaunch_under_torchrun(args.gpus)  # does not return
        args.gpus = world
    if args.workload is None:
        args.workload = "cfg2"
    # BEFORE anything touches the GPU: the counter passes run as children of this process
    live_bytes, live_note = (None, "--no-live-traffic") if args.no_live_traffic else live_traffic(args)
    _w = WORKLOADS[args.workload]
    _bps_in = {"cf32_le": 8, "ci16_le": 4, "cf64_le": 16}[_w["datatype"]]
    stream_ceiling = stream_probe(args, float(_w["hop"] * _bps_in),
                          0.0 if _w["kind"] == "welch" else float(_w["nfft"] * (8 if _w.get("out") == "f64" else 4)))
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
    red_dev = "cpu" if rehearse else None  # device of the scalar reductions

    import spectral_analyzer_amd as sa
    from spectral_analyzer_amd import _lib as L
    from spectral_analyzer_amd import dist as sd

    w = dict(WORKLOADS[args.workload])
    datatype, nfft, hop, window = w["datatype"], w["nfft"], w["hop"], w["window"]
    bps = sa.bytes_per_sample(datatype)
    welch = w["kind"] == "welch"

    # a real (non-null) stream shared by torch and the library, so that the HIP
    # events below bracket exactly the kernels the C ABI launches
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    svc = sa.SpectralService(local_rank, stream=stream.cuda_stream)
    for kv in args.opt:
        key, _, val = kv.partition("=")
        svc.set_option(key, int(val))

    if welch:
        n_seg = w["n_seg"]
        n_psd = args.n_psd or w["n_psd"]
        per_psd = (n_seg - 1) * hop + nfft                      # samples of one PSD (one annotation)
        n_samples = per_psd * n_psd
        first_sample = rank * n_samples
        iq = svc.synth_iq(datatype, SEED, first_sample, n_samples)
        out = torch.empty((n_psd, nfft), dtype=torch.float32, device=iq.device)
        n_lines = n_psd * n_seg                                   # segments ("lines") per GPU and step
        total_lines = n_lines * world
        log2s = None
        fs = 1.0e6

        def step():
            svc.welch_psd(iq, 0, datatype, fs, nfft=nfft, hop=hop, n_seg=n_seg, window=window, n_psd=n_psd,
                          psd_stride_bytes=per_psd * bps, out=out)
    else:
        log2s = args.log2_samples if args.log2_samples is not None else w["log2s"]
        per_gpu = 1 << log2s
        total_lines = (per_gpu * world - nfft) // hop + 1
        l0, l1 = sd.shard_lines(total_lines, world, rank)
        n_lines = l1 - l0
        first_sample, n_samples = sd.shard_span(l0, l1, nfft, hop)  # includes the nfft - hop halo
        iq = svc.synth_iq(datatype, SEED, first_sample, n_samples)
        out_fmt = L.OUT_DB20_F64 if w["out"] == "f64" else L.OUT_DB20_F32
        out = torch.empty((n_lines, nfft), dtype=torch.float64 if w["out"] == "f64" else torch.float32,
                          device=iq.device)

        def step():
            svc.compute_waterfall(iq, 0, nfft, datatype, n_lines, hop=hop, window=window, out_fmt=out_fmt, out=out)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    clocks = ClockSampler(local_rank) if rank == 0 else None
    for _ in range(args.warmup):
        step()
    fence()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if clocks is not None:
        clocks.start()   # a host thread reading sysfs while the timed steps run
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        step()
        b.record(stream)
    fence()
    elapsed = time.perf_counter() - t0
    clock_state = clocks.stop() if clocks is not None else None
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    if dist is not None:
        t = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device=red_dev or iq.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kern_ms = float(t[0].item()), float(t[1].item())

    # ---- outside the timed region: spot check against the oracle, CPU baseline ----
    checked = None
    cpu_baseline = None
    # EVERY rank checks lines of its OWN shard against the oracle (rank 0's lines say nothing about the other devices);
    # the verdicts are reduced below.  Rank 0 builds the checker first, the others load the finished library.
    if rank == 0:
        from oracle import spec_oracle as so
        so.build()
    if dist is not None:
        dist.barrier()
    from oracle import spec_oracle as so
    so.build()
    if welch:
        pick = sorted(set(int(x) for x in np.linspace(0, n_psd - 1, 3)))
        worst = 0.0
        for p in pick:
            raw = iq[p * per_psd * bps:(p + 1) * per_psd * bps].cpu().numpy()
            _, ref = so.welch_psd(raw, 0, datatype, nfft, hop, n_seg, window, so.PSD_DENSITY, fs)
            worst = max(worst, float(np.abs(out[p].cpu().numpy() - ref).max() / ref.max()))
        checked = {"psds": len(pick), "max_err_over_peak": worst, "tol": 5e-6, "ok": bool(worst <= 5e-6)}
    else:
        pick = sorted(set(int(x) for x in np.linspace(0, n_lines - 1, 6)))
        worst = 0.0
        for ln in pick:
            raw = iq[ln * hop * bps:(ln * hop + nfft) * bps].cpu().numpy()
            ref = so.waterfall(raw, 0, datatype, nfft, hop, 1, window)[0]
            got = out[ln].cpu().numpy().astype(np.float64)
            m_ref, m_got = 10 ** (ref / 20), 10 ** (got / 20)
            worst = max(worst, float(np.abs(m_got - m_ref).max() / (m_ref.max() * np.log2(nfft))))
        # fp64: against the reference's own transform (commons-math3: twiddles by recurrence, off by up to 4e-17 N M
        # itself) -- the N-dependent statement of tests/test_gpu_parity.py fp64_tol
        tol = (max(8e-15 * np.log2(nfft) + 5e-14, 4e-17 * nfft) / np.log2(nfft)) if w["out"] == "f64" else 4e-6
        checked = {"lines": len(pick), "max_lin_err_over_M_log2N": worst, "tol": tol, "ok": bool(worst <= tol)}
    if dist is not None:
        # ok_all_ranks: the MINIMUM of the per-rank verdicts; worst_all_ranks: the MAXIMUM of the per-rank errors
        t = torch.tensor([1.0 if checked["ok"] else 0.0, -worst], dtype=torch.float64, device=red_dev or iq.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        checked["ok_all_ranks"] = bool(t[0].item() >= 1.0)
        checked["worst_all_ranks"] = float(-t[1].item())
        checked["ranks_checked"] = world
    else:
        checked["ok_all_ranks"] = checked["ok"]

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # bounded samples: roughly 1e9 butterfly-points for the all-core run, 1.5e8 for the single thread
        per_line = nfft * np.log2(nfft)
        la = int(max(8, min(n_lines, 1.6e9 / per_line)))
        l1t = int(max(4, min(n_lines, 2.0e8 / per_line)))
        host = iq[:((la - 1) * hop + nfft) * bps].cpu().numpy()
        cpu_baseline = cpu_baseline_for(so, host, datatype, nfft, hop, window, la, l1t,
                                        "segments (Hann, hop %d)" % hop if welch else "lines")

    # calibration outside the timed region ("calibrate with a device memcpy"): what a
    # plain device-to-device copy of the output tile moves per second on THIS box, read + write
    copy_gbps = None
    if rank == 0 and out.numel() * out.element_size() >= (1 << 28):
        dst = torch.empty_like(out)
        for _ in range(3):
            dst.copy_(out)
        cev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a, b in cev:
            a.record(stream)
            dst.copy_(out)
            b.record(stream)
        torch.cuda.synchronize()
        copy_gbps = 2.0 * out.numel() * out.element_size() / (float(np.median([a.elapsed_time(b) for a, b in cev])) * 1e-3) / 1e9
        del dst

    res = None
    if rank == 0:
        value = total_lines * args.steps / elapsed
        # roofline.traffic is NOT measured in this run (PMC counters need rocprofv3 passes of their own): it is the
        # figure of the last profile of this workload, profiles/pmc_traffic.json, and says so: traffic_measured_at
        # names the commit and profile the number comes from (tools/summarize_*profile.py write the stamp)
        traffic, traffic_stamp = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and args.log2_samples is None and args.n_psd is None and not args.opt:
            try:
                ent = json.load(open(tpath)).get(args.workload)
                if isinstance(ent, dict):
                    traffic, traffic_stamp = ent.get("bytes"), {k: ent[k] for k in ("commit", "profile", "date") if k in ent}
                else:
                    traffic = ent
            except Exception:
                traffic = None
        if live_bytes is not None:
            traffic, traffic_stamp = live_bytes, {"measured": "live", "how": live_note}
        elif traffic_stamp is not None:
            traffic_stamp = dict(traffic_stamp, measured="stamped", live_skipped=live_note)
        if welch:
            # B_psd = S bps_in + 4 N; 5 N log2 N flops per segment.  35 flop per byte: the vector
            # ALUs bound this kernel, not HBM -- the roofline is the fp32 vector peak, the HBM figure rides along
            b_psd = per_psd * bps + nfft * 4
            flops = 5.0 * nfft * np.log2(nfft) * n_lines
            tfl = flops / (kern_ms * 1e-3) / 1e12
            hbm = n_psd * b_psd / (kern_ms * 1e-3) / 1e9
            roof = {"bound": "valu", "achieved": tfl, "peak": FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
                    "frac": tfl / FP32_VECTOR_TFLOPS, "traffic": traffic, "traffic_measured_at": traffic_stamp,
                    "kernel_ms": kern_ms, "clocks": clock_state,
                    "flops_per_segment": 5.0 * nfft * np.log2(nfft), "segments_per_launch": n_lines,
                    "hbm_achieved_GBps": hbm, "hbm_frac": hbm / HBM_PEAK_GBPS, "bytes_per_psd": b_psd,
                    "psd_per_s": n_psd * world * args.steps / elapsed}
            metric = "Welch PSD segments/sec (%d-pt FFT, Hann, %d %% overlap, %d-segment average)" % (
                nfft, round(100 * (1 - hop / nfft)), n_seg)
            workload = "%s: Welch %d-pt, hop %d, %s, %d segments per PSD, %d PSDs per GPU and step, fp32 PSD out" % (
                args.workload, nfft, hop, datatype, n_seg, n_psd)
        else:
            osz = out.element_size()
            b_line = hop * bps + nfft * osz  # every sample read once, every bin written once
            achieved = n_lines * b_line / (kern_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_measured_at": traffic_stamp,
                    "kernel_ms": kern_ms, "clocks": clock_state, "bytes_per_line": b_line, "lines_per_launch": n_lines,
                    # extras: reads only (the north star is phrased on reads) and the box's own copy rate
                    "read_frac": n_lines * hop * bps / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "copy_GBps": copy_gbps,
                    "frac_of_copy": (achieved / copy_gbps) if copy_gbps else None,
                    # the box's own ceiling for a plain stream of this read : write mix, measured in this run (stream_probe)
                    "stream": dict(stream_ceiling, frac_of_stream=achieved / stream_ceiling["GBps"]) if stream_ceiling else None}
            metric = ("spectrogram lines/sec (4096-pt FFT, 50% overlap)" if nfft == 4096 else
                      "spectrogram lines/sec (%d-pt FFT, %d %% overlap)" % (nfft, round(100 * (1 - hop / nfft))))
            workload = "%s: %d-pt FFT, hop %d, %s, 2^%d samples per GPU, %d lines total, %s out" % (
                args.workload, nfft, hop, datatype, log2s, total_lines, "DB20_F64" if w["out"] == "f64" else "DB20_F32")
        res = {
            "metric": metric,
            "value": value, "unit": "lines/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64" if (not welch and w["out"] == "f64") else "f32", "data": "synthetic",
            "config": {"workload": workload, "window": "rect" if window == 0 else "hann",
                       "sharding": "time-slice x%d" % world, **({"options": args.opt} if args.opt else {})},
            "roofline": roof,
            "cpu_baseline": cpu_baseline,
            "parity_spot_check": checked,
        }


    # ---- N > 1: the same step with the tiles gathered on rank 0, chunk sends overlapped with compute.  LAST, and under
    # a watchdog: whatever happens in here (a rank that cannot allocate, a transport that stalls), rank 0 still prints the
    # headline line and every rank leaves.
    gather = None
    if dist is not None and not welch:
        import threading

        def bail():
            # the compute-only headline is kept (rank 0 prints it, with the marker), but an abandoned gather -- a stalled
            # transport, a hung or faulted GPU -- is NOT a clean run: every rank leaves with a non-zero status
            msg = "gather phase abandoned after %d s" % GATHER_TIMEOUT_S
            if rank == 0:
                res["gather"] = {"error": msg}
                print(json.dumps(res), flush=True)
            print("bench.py rank %d: %s" % (rank, msg), file=sys.stderr, flush=True)
            os._exit(3)

        fence()
        dog = threading.Timer(GATHER_TIMEOUT_S, bail)
        dog.daemon = True
        dog.start()
        # every rank must take the same decision: a root that cannot hold the gathered tile would otherwise leave its
        # peers waiting in their sends
        full, comm, ready = None, None, 1.0
        try:
            comm = torch.cuda.Stream()
            if rank == 0:
                full = torch.empty((total_lines, nfft), dtype=out.dtype, device=out.device)
        except Exception:
            ready = 0.0
        t = torch.tensor([ready], dtype=torch.float64, device=red_dev or iq.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if float(t.item()) < 1.0:
            gather = {"error": "skipped: the root could not allocate the gathered tile (%d x %d %s)" % (total_lines, nfft, out.dtype)}
        else:
            try:
                chunks = max(1, args.gather_chunks)

                def compute_rows(a, b, view):
                    svc.compute_waterfall(iq, (a - l0) * hop * bps, nfft, datatype, b - a, hop=hop, window=window,
                                          out_fmt=out_fmt, out=view)

                def gstep():
                    sd.sharded_waterfall_overlapped(compute_rows, total_lines, nfft, n_chunks=chunks, dst=0,
                                                    out=full if rank == 0 else out, comm_stream=comm)

                gstep()
                fence()
                g0 = time.perf_counter()
                for _ in range(max(1, args.gather_steps)):
                    gstep()
                fence()
                gsec = (time.perf_counter() - g0) / max(1, args.gather_steps)
                t = torch.tensor([gsec], dtype=torch.float64, device=red_dev or iq.device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                gsec = float(t.item())
                ok = None
                if rank == 0:  # the root's own rows of the gathered tile are the lines the plain step produced
                    ok = bool(torch.equal(full[l0:l1], out))
                # the rows that CROSSED the transport: every peer's per-chunk checksums of the tile it sent (all-gathered)
                # against the root's checksums of the rows it received -- a mis-ordered chunk or a transfer that ran ahead
                # of its kernels shows here, not in the root's own rows
                ver = sd.verify_gathered(full, full[l0:l1] if rank == 0 else out, total_lines, chunks, dst=0, via_cpu=rehearse)
                peer_bytes = (total_lines - n_lines) * nfft * out.element_size()
                gather = {"value": total_lines / gsec, "unit": "lines/s", "ms_per_step": gsec * 1e3, "chunks": chunks,
                          "steps": max(1, args.gather_steps), "GBps_into_root": peer_bytes / gsec / 1e9,
                          "root_rows_equal_plain_step": ok,
                          "peer_rows_verified": ver["peer_rows_verified"] if ver else None,
                          "peer_chunks_checked": ver["chunks_checked"] if ver else None,
                          "peer_chunk_mismatches": ver["mismatches"] if ver else None,
                          "what": "compute + gather of every rank's tile on rank 0: %d chunk sends per rank on a second "
                                  "stream behind the chunk's kernels (dist.sharded_waterfall_overlapped)" % chunks}
            except Exception as e:  # the gather timing must never take the headline down with it
                gather = {"error": repr(e)}
        dog.cancel()
        del full

    if rank == 0:
        if gather is not None:
            res["gather"] = gather
        print(json.dumps(res), flush=True)

    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
