# same-box A/B: the working tree against a worktree of HEAD (_ab_head), alternating
for rep in 1 2 3; do
for t in . _ab_head; do
( cd $t; python bench.py --workload gsvmc --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"gsvmc $t\", round(d[\"ms_per_step\"],4), {k: round(v,4) for k,v in d[\"stages_ms\"].items()})" )
done; done
