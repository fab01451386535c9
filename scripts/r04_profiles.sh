#!/bin/bash
# round 4 profiles of the driver's step at HEAD: kernel table of the TIMED steps only (cut marks), CU x time per kernel and queue,
# fabric traffic and MFMA-pipe counters in their own passes (kernel-trace only).  Outputs under gpurun_out/${tag}_*; copy to profiles/.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-r04p}
rm -rf gpurun_out/prof_$tag
MP_BENCH_MARKERS=1 timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-lora-line --no-secondary --roofline-steps 0 > /dev/null 2> gpurun_out/${tag}_bench_prof.err
db=$(ls gpurun_out/prof_$tag/*/*.db | head -1)
python scripts/rocpd_stats.py $db 8 gpurun_out/${tag}_kernel_stats.md | head -16
(cd scripts && python rocpd_cutime.py ../$db 8 ../gpurun_out/${tag}_cu_time.md | tail -8)
rm -rf gpurun_out/prof_$tag
TAG=$tag bash scripts/bench_pmc.sh > gpurun_out/${tag}_pmc.log 2>&1; tail -3 gpurun_out/${tag}_pmc.log
TAG=$tag bash scripts/bench_mfma_pmc.sh > gpurun_out/${tag}_mfma.log 2>&1; tail -3 gpurun_out/${tag}_mfma.log
rm -rf gpurun_out/bench_pmc gpurun_out/mfma_pmc
