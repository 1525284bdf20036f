#!/bin/bash
# usage: tools_prof.sh <tag> "<pmc counters>"   -- runs bench under rocprofv3 with PMC counters (own pass, no tracing)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout ${PROF_TIMEOUT:-600} rocprofv3 --pmc $@ --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-extras > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
echo "rc=$?"
