#!/bin/bash
# rocprofv3 kernel table of the decode bench (MoE): gpurun_out/<tag>_decode_kernel_stats.md
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-r04}
rm -rf /tmp/dprof
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/dprof -- python scripts/decode_bench.py --new 16 > /tmp/dprof.log 2> /tmp/dprof.err
tail -1 /tmp/dprof.log; tail -2 /tmp/dprof.err
db=$(ls /tmp/dprof/*/*.db | head -1); python scripts/rocpd_stats.py $db 1 gpurun_out/${tag}_decode_kernel_stats.md > /dev/null
head -20 gpurun_out/${tag}_decode_kernel_stats.md | cut -c1-150
