python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide.py tests/test_gpu_dist.py -x -q 2>&1 | tail -4
for wl in gsvmc beta n12; do python bench.py --workload $wl --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$wl\", round(d[\"ms_per_step\"],4), {k: round(v,4) for k,v in d[\"stages_ms\"].items()}, d.get(\"stages_note\",\"\")[-60:])"; done
python bench.py --workload c5 --no-extras --steps 5 --warmup 2 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"c5\", round(d[\"ms_per_step\"],4), {k: round(v,4) for k,v in d[\"stages_ms\"].items()})"
bash tools/prof_stats.sh r04j > /dev/null 2>&1
head -22 gpurun_out/r04j_kernel_stats.csv | cut -d, -f1-4 | cut -c1-130
