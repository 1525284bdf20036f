for r in 1 2; do for n in w0 w1; do for wl in n12 c5; do
FERMIFLOW_LIB=$PWD/fermiflow_amd/libfermiflow_hip_$n.so python bench.py --workload $wl --steps $([ $wl = c5 ] && echo 5 || echo 20) --warmup 3 --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$n $wl round $r: ms_per_step %.4f  eloc pass %.4f ms  evals %.2f  E %.6f' % (d['ms_per_step'], r['avg_launch_ms'], r['rhs_evals_per_walker'], d['E']))
"
done; done; done
