# the round's evidence set: bench lines of all configs, kernel tables, per-kernel counters  ->  gpurun_out/r04n_*
set -x
python bench.py > gpurun_out/r04n_bench.json 2> gpurun_out/r04n_bench.err
python bench.py --workload beta > gpurun_out/r04n_bench_beta.json 2>/dev/null
python bench.py --workload n12 > gpurun_out/r04n_bench_n12.json 2>/dev/null
python bench.py --workload c5 --steps 5 --warmup 2 > gpurun_out/r04n_bench_c5.json 2>/dev/null
bash tools/prof_stats.sh r04n > /dev/null 2>&1
bash tools/prof_stats.sh r04n_beta --workload beta > /dev/null 2>&1
bash tools/prof_stats.sh r04n_c5 --workload c5 --steps 3 --warmup 1 > /dev/null 2>&1
python tools/pmc_kernels.py gpurun_out/r04n_kernels_pmc.json > gpurun_out/r04n_pmc.log 2>&1
python - <<'PY'
import json
for f in ("r04n_bench","r04n_bench_beta","r04n_bench_n12","r04n_bench_c5"):
    try:
        d=json.load(open(f"gpurun_out/{f}.json")); print(f, round(d["ms_per_step"],4), d["value"], d["stages_ms"], "frac", round(d["roofline"]["frac"],4), "traffic", d["roofline"].get("traffic"))
    except Exception as e: print(f, "failed", e)
PY
