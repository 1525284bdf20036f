#!/bin/bash
# Round 6, GPU call 14: adjoint row masks -- the adjoint / sweep parity tests, then the config-2 bench line with counters of the adjoint kernel.
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "adjoint or gsvmc or betavmc or determin or table" 2>&1 | grep -v amdgpu.ids | tail -4
timeout 900 python bench.py --steps 20 --warmup 5 --cpu-walkers 0 2>/dev/null | grep '^{' > gpurun_out/r06_n_bench.json
python - <<PY
import json
d = json.loads(open('gpurun_out/r06_n_bench.json').read())
print('head %.4f ms' % d['ms_per_step'], 'pass %.4f' % d['roofline']['avg_launch_ms'], 'frac %.4f' % d['roofline']['frac'], 'stages', d['stages_ms'])
print('adjoint', {k: d['roofline_adjoint'][k] for k in ('frac', 'avg_launch_ms', 'rhs_evals_per_walker')})
print('trained', d['trained_leg']['ms_per_step'], 'long', d['long_window_leg']['ms_per_step'])
PY
timeout 900 python tools/pmc_kernels.py gpurun_out/r06_n_kernels_pmc.json 2>&1 | grep -v amdgpu | tail -8
