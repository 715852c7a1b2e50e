mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/s13_pytest.txt 2>&1; tail -3 gpurun_out/s13_pytest.txt
bash tools/ab_variant.sh v2nowait cfg2 cfg3 n8192 n16384 n1024 > gpurun_out/s13_ab_v2nowait.txt 2>&1; cat gpurun_out/s13_ab_v2nowait.txt
