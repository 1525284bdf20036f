# round 6 evidence set: bench lines of all four workloads, kernel tables, per-kernel counters  ->  gpurun_out/<tag>_*
tag=${1:-r06z}
set -x
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python bench.py --workload beta --steps 20 --warmup 5 > gpurun_out/${tag}_bench_beta.json 2>/dev/null
python bench.py --workload n12 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_n12.json 2>/dev/null
python bench.py --workload c5 --steps 5 --warmup 2 > gpurun_out/${tag}_bench_c5.json 2>/dev/null
bash tools/prof_stats.sh ${tag} > /dev/null 2>&1
bash tools/prof_stats.sh ${tag}_beta --workload beta > /dev/null 2>&1
bash tools/prof_stats.sh ${tag}_n12 --workload n12 > /dev/null 2>&1
bash tools/prof_stats.sh ${tag}_c5 --workload c5 --steps 3 --warmup 1 > /dev/null 2>&1
python tools/pmc_kernels.py gpurun_out/${tag}_kernels_pmc.json > gpurun_out/${tag}_pmc.log 2>&1
# the long window (the flow trains: three steps per walker where two did), the learned first-step table and the adjoint over it, kernel timeline of its last iterations
python bench.py --steps 200 --warmup 20 --no-extras > gpurun_out/${tag}_bench_200steps.json 2>/dev/null
python tools/probes/h_table_drift.py 2e-5 300 20 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_h_table_drift.txt
( R=$PWD; cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_trace -- python3 $R/tools/probes/long_run.py 2e-5 260 > /dev/null 2>&1; f=$(ls $R/gpurun_out/${tag}_trace/*/*kernel_trace.csv | head -1); python3 $R/tools/probes/pass_timeline.py $f 30 > $R/gpurun_out/${tag}_iteration_timeline_late.txt; rm -rf $R/gpurun_out/${tag}_trace )
python - <<PY
import json
for f in ("${tag}_bench","${tag}_bench_beta","${tag}_bench_n12","${tag}_bench_c5","${tag}_bench_200steps"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1]); print(f, round(d["ms_per_step"],4), d["value"], d["stages_ms"], "frac", round(d["roofline"]["frac"],4), "traffic", d["roofline"].get("traffic"))
    except Exception as e: print(f, "failed", e)
PY
