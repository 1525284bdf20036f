#!/bin/bash
# Round 6, GPU call 1: shape-trained weight sets + tolerance-policy error by sens_tol; the box's baseline bench line; consume ticks by stage.
mkdir -p gpurun_out
timeout 1500 python tools/probes/train_fixtures_r06.py soak,n12,c5 3 > gpurun_out/r06_a_fixtures.txt 2>&1
echo "fixtures rc $?"
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_a_bench.json 2> gpurun_out/r06_a_bench.err
echo "bench rc $?"
FERMIFLOW_LIB=$PWD/fermiflow_amd/libfermiflow_hip_stamps.so FF_STATS_WORDS=65700 timeout 300 python tools/kbench.py > gpurun_out/r06_a_kbench_stamps.json 2>&1
echo "kbench rc $?"
tail -30 gpurun_out/r06_a_fixtures.txt
