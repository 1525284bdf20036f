#!/bin/bash
# the routing threshold of the local-energy pass: bench.py --no-extras per value of FERMIFLOW_HEAVY_CLASS (args; -1: no routing) -> GSVMC.heavy_class -> ff_ode.heavy_class
for c in "$@"; do
  FERMIFLOW_HEAVY_CLASS=$c python bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('heavy class', sys.argv[1], 'ms/iter', round(d['ms_per_step'], 3), 'sensitivity pass', round(d['roofline']['avg_launch_ms'], 3), 'evals', round(d['roofline']['rhs_evals_per_walker'], 2))" $c
done
