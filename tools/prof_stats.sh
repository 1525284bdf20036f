#!/bin/bash
# kernel table of one bench run: tools/prof_stats.sh <tag> [bench args]  ->  gpurun_out/<tag>_kernel_stats.csv (+ the bench line)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-extras "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench_under_profiler.json 2> $out.err
echo "rc=$?"
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv && head -12 $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv | cut -c1-160
