#!/bin/bash
for rep in 1 2; do
for n in v23 v10; do
  FERMIFLOW_LIB=$PWD/fermiflow_amd/libfermiflow_hip_$n.so timeout 900 python bench.py --steps 20 --warmup 5 --no-pmc --cpu-walkers 0 2>/dev/null | grep '^{' > gpurun_out/r06_m_bench_${n}_$rep.json
  python - <<PY
import json
d = json.loads(open('gpurun_out/r06_m_bench_${n}_$rep.json').read())
print('$n', 'head %.4f ms (pass %.4f)' % (d['ms_per_step'], d['roofline']['avg_launch_ms']), '| long %.4f (evals %.2f, pass %.4f)' % (d['long_window_leg']['ms_per_step'], d['long_window_leg']['rhs_evals_per_walker'], d['long_window_leg']['eloc_pass_ms']),
      '| trained %.4f (evals %.2f, pass %.4f)' % (d['trained_leg']['ms_per_step'], d['trained_leg']['rhs_evals_per_walker'], d['trained_leg']['eloc_kernel_ms']),
      '| driver 100: %.4f (pass %.4f) 300: %.4f (evals %.2f, pass %.4f)' % (d['driver_leg']['iter_100']['ms_per_step'], d['driver_leg']['iter_100']['eloc_pass_ms'], d['driver_leg']['iter_300']['ms_per_step'], d['driver_leg']['iter_300']['rhs_evals_per_walker'], d['driver_leg']['iter_300']['eloc_pass_ms']))
PY
done
done
