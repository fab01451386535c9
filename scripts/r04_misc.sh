#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests/test_gpu_bench_ep.py -x -q 2>&1 | tail -5
rm -rf gpurun_out/prof_dec
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_dec -- python scripts/decode_bench.py > /dev/null 2> gpurun_out/dec.err
db=$(ls gpurun_out/prof_dec/*/*.db | head -1); python scripts/rocpd_stats.py $db 1 gpurun_out/r04_decode_kernel_stats.md | head -14
rm -rf gpurun_out/prof_dec
