# same-box A/B of library variants (FERMIFLOW_LIB) at config 2; args: variant names ("" = the default build)
for rep in 1 2 3; do
for v in "$@"; do
  if [ "$v" = base ]; then unset FERMIFLOW_LIB; else export FERMIFLOW_LIB=$PWD/fermiflow_amd/libfermiflow_hip_$v.so; fi
  python bench.py --workload gsvmc --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"gsvmc $v\", round(d[\"ms_per_step\"],4), {k: round(v,4) for k,v in d[\"stages_ms\"].items()})"
done; done
