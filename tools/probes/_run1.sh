{
for r in 1 2; do for n in head rows6; do
for sh in "5 5" "5 4" "4 3"; do
set -- $sh
FERMIFLOW_LIB=$PWD/fermiflow_amd/libfermiflow_hip_$n.so python bench.py --nup $1 --ndown $2 --walkers-per-gpu 32768 --steps 20 --warmup 3 --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$n nup=$1 ndown=$2 round $r: ms_per_step %.4f  eloc pass %.4f ms  evals %.2f  E %.6f stages %s' % (d['ms_per_step'], r['avg_launch_ms'], r['rhs_evals_per_walker'], d['E'], {k: round(v,3) for k,v in d['stages_ms'].items()}))
"
done; done; done
} > gpurun_out/r06_y23_ab_rows_static_stages.txt 2>&1
cat gpurun_out/r06_y23_ab_rows_static_stages.txt
