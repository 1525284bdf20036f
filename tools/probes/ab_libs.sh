#!/bin/bash
# A/B of variant builds on ONE box: bench.py --no-extras per library, alternating, R rounds.  usage: ab_libs.sh R steps lib1 lib2 ...
# (libraries: names of fermiflow_amd/libfermiflow_hip_<name>.so made with `make variant` or by hand)
R=$1; K=$2; shift 2
mkdir -p gpurun_out
for r in $(seq 1 $R); do
  for n in "$@"; do
    FERMIFLOW_LIB=$PWD/fermiflow_amd/libfermiflow_hip_$n.so python bench.py --steps $K --warmup 20 --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$n round $r: ms_per_step %.4f  eloc pass %.4f ms  evals %.2f  frac %.4f  E %.6f' % (d['ms_per_step'], r['avg_launch_ms'], r['rhs_evals_per_walker'], r['frac'], d['E']))
"
  done
done
