set -x
python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide.py tests/test_gpu_dist.py -x -q 2>&1 | tail -12 > gpurun_out/r04i_tests.log
cat gpurun_out/r04i_tests.log
for wl in gsvmc beta n12; do python bench.py --workload $wl --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$wl\", round(d[\"ms_per_step\"],4), d[\"stages_ms\"])"; done
python bench.py --workload c5 --no-extras --steps 5 --warmup 2 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"c5\", round(d[\"ms_per_step\"],4), d[\"stages_ms\"])"
