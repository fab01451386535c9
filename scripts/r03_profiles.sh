#!/bin/bash
# round-3 evidence at HEAD: the default bench line, the rocprofv3 kernel table of the same command (short), HBM-side traffic and MFMA-pipe
# counters (each its own --pmc pass, kernel-trace only).  Usage (GPU box): bash scripts/r03_profiles.sh <tag>
tag=${1:-r03}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -2 gpurun_out/${tag}_bench.err; cut -c1-300 gpurun_out/${tag}_bench.json
rm -rf gpurun_out/prof_$tag
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-lora-line > gpurun_out/${tag}_bench_prof.json 2> gpurun_out/${tag}_bench_prof.err
db=$(ls gpurun_out/prof_$tag/*/*.db | head -1); python scripts/rocpd_stats.py $db 10 gpurun_out/${tag}_kernel_stats.md; head -16 gpurun_out/${tag}_kernel_stats.md
rm -rf gpurun_out/prof_$tag
TAG=$tag bash scripts/bench_pmc.sh > gpurun_out/${tag}_pmc.log 2>&1; tail -30 gpurun_out/${tag}_pmc.log
TAG=$tag bash scripts/bench_mfma_pmc.sh > gpurun_out/${tag}_mfma.log 2>&1; tail -30 gpurun_out/${tag}_mfma.log
