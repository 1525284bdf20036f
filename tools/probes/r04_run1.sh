set -x
python -m pytest tests/test_gpu_parity.py -x -q -k "mcmc or philox or particle_split or heavy_walker or persistent or ho3d" 2>&1 | tail -15 > gpurun_out/r04a_tests1.log
python -m pytest tests/test_gpu_wide.py -x -q -k "config5_local_energy or fp32" 2>&1 | tail -15 > gpurun_out/r04a_tests2.log
python tools/kbench.py > gpurun_out/r04a_kb_new.json 2>&1
FF_MCMC_CLASSIC=1 python tools/kbench.py > gpurun_out/r04a_kb_classic.json 2>&1
python tools/kbench.py --nup 3 --ndown 0 > gpurun_out/r04a_kb_new_30.json 2>&1
python bench.py > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err
python bench.py --workload beta --no-extras > gpurun_out/r04a_bench_beta.json 2> gpurun_out/r04a_bench_beta.err
cat gpurun_out/r04a_tests1.log gpurun_out/r04a_tests2.log gpurun_out/r04a_kb_new.json gpurun_out/r04a_kb_classic.json
