for r in 2 3 2 1 2 3; do
  python bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline --opt large_ring=$r > gpurun_out/b_r$r.json
  python -c "
import json;d=json.loads(open('gpurun_out/b_r$r.json').read().strip().splitlines()[-1]);print('ring $r', d['roofline']['kernel_ms'], d['roofline']['frac'], d['parity_spot_check']['ok'])"
done
