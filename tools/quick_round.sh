mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/s8_pytest.txt 2>&1; tail -3 gpurun_out/s8_pytest.txt
bash tools/ab_variant.sh v2wfall cfg2 n8192 n16384 cfg4 > gpurun_out/s8_ab_v2wfall.txt 2>&1; cat gpurun_out/s8_ab_v2wfall.txt
bash tools/ab_variant.sh slpnorm cfg4 cfg3 > gpurun_out/s8_ab_slpnorm.txt 2>&1; cat gpurun_out/s8_ab_slpnorm.txt
