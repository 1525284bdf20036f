#!/bin/bash
# does the headline depend on the number of untimed warm-up iterations?  args: "steps warmup" pairs
while [ $# -ge 2 ]; do
  python bench.py --steps $1 --warmup $2 --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('steps', sys.argv[1], 'warmup', sys.argv[2], 'ms/iter', round(d['ms_per_step'], 3), 'evals', round(d['roofline']['rhs_evals_per_walker'], 2), {k: round(v, 3) for k, v in d['stages_ms'].items()})" $1 $2
  shift 2
done
