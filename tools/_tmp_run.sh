timeout -k 10 400 python -m pytest tests/test_gpu_large.py -m gpu -x -q -k "not full_size" 2>&1 | tail -3
for v in "" nosx "" nosx; do
  SPEC_LIB_VARIANT=$v python bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b_$v.json
  python -c "
import json;d=json.loads(open('gpurun_out/b_$v.json').read().strip().splitlines()[-1]);print('[$v]', d['roofline']['kernel_ms'], d['roofline']['frac'], d['parity_spot_check']['ok'])"
done
