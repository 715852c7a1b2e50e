#!/bin/bash
# round-4: the reference's own call shape re-recorded on the final library (profiles/r04_latency.txt, r04_host_path.txt)
mkdir -p gpurun_out
timeout -k 10 500 python tools/latency.py > gpurun_out/r04_latency.txt 2> gpurun_out/r04_latency.err || { tail -5 gpurun_out/r04_latency.err; exit 1; }
cat gpurun_out/r04_latency.txt
timeout -k 10 400 python tools/bench_host.py 28 > gpurun_out/r04_host_path.txt 2> gpurun_out/r04_host_path.err || { tail -5 gpurun_out/r04_host_path.err; exit 1; }
cat gpurun_out/r04_host_path.txt
