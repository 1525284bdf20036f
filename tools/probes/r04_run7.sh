python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide.py -x -q -k "mcmc or sampler or philox or metropolis or three_dim or driver or config5" 2>&1 | tail -3
python bench.py --workload n12 --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"n12\", round(d[\"ms_per_step\"],4), {k: round(v,4) for k,v in d[\"stages_ms\"].items()})"
python bench.py --workload c5 --no-extras --steps 5 --warmup 2 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"c5\", round(d[\"ms_per_step\"],4), {k: round(v,4) for k,v in d[\"stages_ms\"].items()})"
