set -x
python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide.py tests/test_gpu_dist.py -x -q 2>&1 | tail -12 > gpurun_out/r04d_tests.log
python tools/kbench.py 2>&1 | tail -1 > gpurun_out/r04d_kb.json
python bench.py --no-pmc > gpurun_out/r04d_bench.json 2> gpurun_out/r04d_bench.err
python bench.py --workload beta --no-extras > gpurun_out/r04d_bench_beta.json 2>/dev/null
python bench.py --workload n12 --no-extras > gpurun_out/r04d_bench_n12.json 2>/dev/null
cat gpurun_out/r04d_tests.log gpurun_out/r04d_kb.json
python - <<'PY'
import json
for f in ("r04d_bench","r04d_bench_beta","r04d_bench_n12"):
    d=json.load(open(f"gpurun_out/{f}.json")); print(f, round(d["ms_per_step"],4), d["stages_ms"], d.get("trained_leg",{}).get("ms_per_step"))
PY
