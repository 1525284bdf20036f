#!/bin/bash
# Round 6, GPU call 4: the retry queue -- its own test first (under a timeout: a protocol bug would hang), then the whole suite, then A/B against v0.
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -s -k "retry_queue" 2>&1 | grep -v amdgpu.ids | tail -5
echo "retry test rc ${PIPESTATUS[0]}"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -4
bash tools/probes/ab_libs.sh 2 100 v0 v5 | tee gpurun_out/r06_d_ab_retry_head.txt
for n in v0 v5; do
  FERMIFLOW_LIB=$PWD/fermiflow_amd/libfermiflow_hip_$n.so timeout 900 python bench.py --steps 20 --warmup 5 --no-pmc --cpu-walkers 0 2>/dev/null | grep '^{' > gpurun_out/r06_d_bench_$n.json
  python - <<PY
import json
d = json.loads(open('gpurun_out/r06_d_bench_$n.json').read())
print('$n', 'head %.4f ms' % d['ms_per_step'], '| long %.4f (evals %.2f, pass %.4f)' % (d['long_window_leg']['ms_per_step'], d['long_window_leg']['rhs_evals_per_walker'], d['long_window_leg']['eloc_pass_ms']),
      '| trained %.4f (evals %.2f, pass %.4f)' % (d['trained_leg']['ms_per_step'], d['trained_leg']['rhs_evals_per_walker'], d['trained_leg']['eloc_kernel_ms']),
      '| driver 300: %.4f (evals %.2f, pass %.4f)' % (d['driver_leg']['iter_300']['ms_per_step'], d['driver_leg']['iter_300']['rhs_evals_per_walker'], d['driver_leg']['iter_300']['eloc_pass_ms']))
PY
done
