#!/bin/bash
# what the forced one-rank RCCL path costs: kernel tables of the same bench with and without it
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for f in 1 0; do
  rm -rf gpurun_out/prof_d$f
  MP_BENCH_FORCE_DIST=$f timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_d$f -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --roofline-steps 0 > /dev/null 2> gpurun_out/r03_d$f.err
  grep "gpu leg" gpurun_out/r03_d$f.err
  db=$(ls gpurun_out/prof_d$f/*/*.db | head -1); python scripts/rocpd_stats.py $db 8 gpurun_out/r03_dist${f}_kernel_stats.md; head -14 gpurun_out/r03_dist${f}_kernel_stats.md
  rm -rf gpurun_out/prof_d$f
done
