#!/bin/bash
# run-to-run spread of the headline: bench.py --no-extras several times per value of FERMIFLOW_PREFETCH_DELAY_US (args: the values)
for d in "$@"; do
  FERMIFLOW_PREFETCH_DELAY_US=$d python bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('delay_us', sys.argv[1], 'ms/iter', round(d['ms_per_step'], 3), 'adjoint stage', round(d['stages_ms']['adjoint'], 3), 'eloc stage', round(d['stages_ms']['eloc'], 3))" $d
done
