python -c "import torch; print('stream priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
for rep in 1 2 3; do
for p in 0 1 -1; do
  FERMIFLOW_PREFETCH_PRIORITY=$p python bench.py --workload gsvmc --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"priority $p:\", round(d[\"ms_per_step\"],4), {k: round(v,4) for k,v in d[\"stages_ms\"].items()})"
done; done
