#!/bin/bash
# kernel-time A/B of the attention forward (rocprofv3 kernel trace: durations, not launch-bound wall time)
# usage: attn_ab.sh "ENV=VAL ..." "ENV=VAL ..."   (one profile per argument; default: pipelined v3 against the v2 loop)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
[ $# -eq 0 ] && set -- "MP_ATTN_PIPE=1" "MP_ATTN_PIPE=0" "MP_ATTN_PIPE=0 MP_ATTN_TUNED=0"
i=0
for envs in "$@"; do
  i=$((i+1)); rm -rf /tmp/attn_prof_$i
  env $envs rocprofv3 --kernel-trace --stats -d /tmp/attn_prof_$i -- python scripts/attn_bench.py > /tmp/attn_$i.log 2>&1
  echo "== $envs"; grep -E "llama|clip" /tmp/attn_$i.log | sed -n 4,6p
  db=$(ls /tmp/attn_prof_$i/*/*.db | head -1)
  python scripts/rocpd_stats.py $db 1 /tmp/attn_stats_$i.md > /dev/null 2>&1; grep -E "attn_fwd[23]" /tmp/attn_stats_$i.md | head -4
done
