#!/bin/bash
# Round wrap-up on the GPU box: [tests] + bench line + rocprofv3 kernel table (+ forward-only bench).  Usage: bash scripts/final_run.sh <tag> [notests]
tag=${1:-x}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ "$2" != "notests" ]; then bash scripts/gpu_tests.sh; echo "== tests rc=$?"; fi
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -2 gpurun_out/${tag}_bench.err; cut -c1-300 gpurun_out/${tag}_bench.json
rm -rf gpurun_out/prof_$tag
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_bench_prof.json 2> gpurun_out/${tag}_bench_prof.err
db=$(ls gpurun_out/prof_$tag/*/*.db | head -1); python scripts/rocpd_stats.py $db 4 gpurun_out/${tag}_kernel_stats.md; head -16 gpurun_out/${tag}_kernel_stats.md
rm -rf gpurun_out/prof_$tag
python scripts/forward_bench.py > gpurun_out/${tag}_forward_bench.json 2> gpurun_out/${tag}_forward.err; tail -3 gpurun_out/${tag}_forward.err; cat gpurun_out/${tag}_forward_bench.json | cut -c1-600
