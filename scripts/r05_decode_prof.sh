#!/bin/bash
# decode layer timeline (dense and MoE): gpurun_out/<tag>_decode_timeline_{dense,moe}.md
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-r05}
for kind in dense moe; do
  rm -rf /tmp/dprof
  flag=""; [ $kind = dense ] && flag="--dense"
  timeout 600 rocprofv3 --kernel-trace -d /tmp/dprof -- python scripts/decode_bench.py --new 16 $flag > /tmp/dprof.log 2> /tmp/dprof.err
  tail -1 /tmp/dprof.log; tail -2 /tmp/dprof.err
  db=$(ls /tmp/dprof/*/*.db | head -1); python scripts/decode_timeline.py $db gpurun_out/${tag}_decode_timeline_$kind.md
  python scripts/decode_bench.py --new 32 $flag | tail -1
done
