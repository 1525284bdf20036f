python -m pytest tests/test_gpu_wide.py -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c5x -- python3 $GRAFT_REPO_ROOT/bench.py --workload c5 --steps 3 --warmup 1 --no-extras > $GRAFT_REPO_ROOT/gpurun_out/c5x.json 2>/dev/null
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_c5x -name "*kernel_stats.csv" | head -1); head -8 $f | cut -d, -f1-4 | cut -c1-110
