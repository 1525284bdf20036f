# same-box kernel tables of the working tree and of a worktree of HEAD (_ab_head)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for t in new head; do
  d=$R; [ $t = head ] && d=$R/_ab_head
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_$t -- python3 $d/bench.py --workload gsvmc --no-extras --steps 10 --warmup 5 > $R/gpurun_out/ab_$t.log 2>&1
  f=$(find $R/gpurun_out/ab_$t -name "*kernel_stats.csv" | head -1)
  echo "== $t"; head -14 $f | cut -d, -f1-4 | cut -c1-150
done
