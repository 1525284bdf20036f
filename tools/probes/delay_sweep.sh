# prefetch delay sweep at config 2 (FERMIFLOW_PREFETCH_DELAY_US): ms per iteration and the wait for the prefetched walkers
for rep in 1 2; do
for d in 0 3 10 25; do
  FERMIFLOW_PREFETCH_DELAY_US=$d python bench.py --workload gsvmc --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"delay $d us:\", round(d[\"ms_per_step\"],4), {k: round(v,4) for k,v in d[\"stages_ms\"].items()})"
done; done
