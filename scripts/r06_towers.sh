#!/bin/bash
# round 6 opening measurement: the towers alone (kernel tables), what they cost the step (tower_cost.py), and the headline on this box
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-r06a}
for t in sam clip; do
  rm -rf gpurun_out/prof_$t
  TOWER=$t timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$t -- python scripts/r06_tower_trace.py 10 > gpurun_out/${tag}_${t}_alone.log 2>&1
  db=$(ls gpurun_out/prof_$t/*/*.db | head -1)
  python scripts/rocpd_stats.py $db 13 gpurun_out/${tag}_${t}_alone_kernel_stats.md > /dev/null
  tail -2 gpurun_out/${tag}_${t}_alone.log
  rm -rf gpurun_out/prof_$t
done
timeout 900 python scripts/tower_cost.py > gpurun_out/${tag}_tower_cost.txt 2>&1; cat gpurun_out/${tag}_tower_cost.txt | tail -5
timeout 900 python scripts/tower_bench.py > gpurun_out/${tag}_tower_bench.json 2>&1; tail -30 gpurun_out/${tag}_tower_bench.json
