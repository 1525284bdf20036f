python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/r04l_bench.json 2> gpurun_out/r04l_bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r04l_bench.json")); print("bench", round(d["ms_per_step"],4), d["value"], "frac", round(d["roofline"]["frac"],4), "traffic", d["roofline"].get("traffic"), "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"])
PY
