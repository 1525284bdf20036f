# where / when the sampler's stream is released: ms per iteration over 40 steps (the overlap is bimodal per iteration)
for rep in 1 2; do
for cfg in "adj 10" "est 10" "est 0" "est 25" "adj 0"; do
  set -- $cfg
  FERMIFLOW_PREFETCH_GO=$1 FERMIFLOW_PREFETCH_DELAY_US=$2 python bench.py --workload gsvmc --no-extras --steps 40 --warmup 10 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"go=$1 delay=$2:\", round(d[\"ms_per_step\"],4), {k: round(v,4) for k,v in d[\"stages_ms\"].items()})"
done; done
