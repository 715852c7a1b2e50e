mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/s10_pytest.txt 2>&1; tail -3 gpurun_out/s10_pytest.txt
timeout -k 10 400 python tools/bench_coop.py 256 > gpurun_out/s10_coop256.txt 2>&1; cat gpurun_out/s10_coop256.txt
