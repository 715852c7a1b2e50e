#!/usr/bin/env python3
"""The single-process multi-device path at the C ABI, timed: spec_waterfall_multi over one context per visible GPU
(what a one-JVM host uses), device-resident shards -> a tile on device 0, the peers' pieces sent with
hipMemcpyPeerAsync behind their kernels.  Ready for the day a multi-GPU node is at hand; on a one-GPU box
``--contexts N`` puts N contexts on device 0 (a rehearsal of the code path, not an xGMI number).

    python tools/bench_multi.py [--contexts N] [--workload cfg2|cfg3] [--log2-samples 28] [--steps 5] [--chunks 8]

Prints one JSON line: compute + gather lines/s, the path every peer's copies took ("multi_peer_access": 1 direct --
peer access enabled --, 0 staged by the runtime, 2 same device), and the verdict of one "multi_verify" pass (every
piece checksummed on the peer's device before it leaves and where it landed on the consumer's)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spectral_analyzer_amd as sa

W = {"cfg2": ("cf32_le", 4096, 2048), "cfg3": ("ci16_le", 4096, 2048), "n32768f": ("cf32_le", 32768, 16384)}
ap = argparse.ArgumentParser()
ap.add_argument("--contexts", type=int, default=0, help="default: one per visible GPU")
ap.add_argument("--workload", default="cfg2", choices=sorted(W))
ap.add_argument("--log2-samples", type=int, default=30, help="samples per context")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--chunks", type=int, default=8)
a = ap.parse_args()

n_dev = torch.cuda.device_count()
n = a.contexts or n_dev
datatype, nfft, hop = W[a.workload]
bps = sa.bytes_per_sample(datatype)
total_samples = n << a.log2_samples
L = (total_samples - nfft) // hop + 1
svcs, shards = [], []
for r in range(n):
    dev = r % n_dev
    torch.cuda.set_device(dev)
    s = sa.SpectralService(dev, stream=torch.cuda.Stream(device=dev).cuda_stream)
    l0, l1 = sa.shard_lines(L, n, r)
    first, nb = sa.shard_span(l0, l1, datatype, nfft, hop)
    shards.append(s.synth_iq(datatype, 0x5EC7A11A, first // bps, nb // bps) if nb else None)   # every device generates its own span
    s.synchronize()
    svcs.append(s)
torch.cuda.set_device(0)
out = torch.empty((L, nfft), dtype=torch.float32, device="cuda:0")

def step():
    sa.compute_waterfall_multi(svcs, shards, 0, nfft, datatype, L, hop=hop, out=out, n_bytes=total_samples * bps, n_chunks=a.chunks)

step()
t0 = time.perf_counter()
for _ in range(a.steps):
    step()                                   # returns when the whole tile is on device 0
sec = (time.perf_counter() - t0) / a.steps
svcs[0].set_option("multi_verify", 1)
verdict, err = True, None
try:
    step()
except RuntimeError as e:
    verdict, err = False, str(e)
svcs[0].set_option("multi_verify", 0)
# the root's own rows against a plain single-context call on the same shard (they never travelled)
l0, l1 = sa.shard_lines(L, n, 0)
own = svcs[0].compute_waterfall(shards[0], 0, nfft, datatype, l1 - l0, hop=hop)
torch.cuda.synchronize()
peer_bytes = (L - (l1 - l0)) * nfft * 4
print(json.dumps({
    "what": "spec_waterfall_multi, %d contexts on %d device(s), %s: %d-pt, hop %d, %s, 2^%d samples per context, %d lines, device tile on ctx[0]" % (
        n, n_dev, a.workload, nfft, hop, datatype, a.log2_samples, L),
    "value": L / sec, "unit": "lines/s (compute + gather)", "ms_per_step": sec * 1e3, "GBps_into_root": peer_bytes / sec / 1e9,
    "chunks": a.chunks, "multi_peer_access": [s.get_option("multi_peer_access") for s in svcs],
    "multi_verified_pieces": [s.get_option("multi_verified") for s in svcs], "peer_rows_verified": verdict, "verify_error": err,
    "root_rows_equal_plain_call": bool(torch.equal(out[l0:l1], own))}))
for s in svcs:
    s.close()
