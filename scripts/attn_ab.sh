#!/bin/bash
# kernel-time A/B of the attention forward (rocprofv3 kernel trace: durations, not launch-bound wall time)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for t in 1 0; do
  rm -rf /tmp/attn_prof_$t
  MP_ATTN_TUNED=$t rocprofv3 --kernel-trace --stats -d /tmp/attn_prof_$t -- python scripts/attn_bench.py > /tmp/attn_$t.log 2>&1
  echo "== MP_ATTN_TUNED=$t"; grep -E "llama causal|clip  " /tmp/attn_$t.log | head -3
  db=$(ls /tmp/attn_prof_$t/*/*.db | head -1)
  python scripts/rocpd_stats.py $db 1 /tmp/attn_stats_$t.md > /dev/null 2>&1; grep -E "attn_fwd" /tmp/attn_stats_$t.md | head -8
done
