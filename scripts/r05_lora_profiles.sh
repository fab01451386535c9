#!/bin/bash
# round 5: kernel table + CU x time of the TIMED LoRA stage-III steps (cut marks), the table the round-4 review asked for before any change.
# Usage (on the GPU box): bash scripts/r05_lora_profiles.sh <tag>.  Outputs gpurun_out/<tag>_lora_kernel_stats.md, <tag>_lora_cu_time.md; copy to profiles/.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-r05a}
rm -rf gpurun_out/prof_$tag
MP_BENCH_MARKERS=1 timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -- python bench.py --lora --steps 6 --warmup 3 --no-kernel-timer > gpurun_out/${tag}_lora_bench_prof.json 2> gpurun_out/${tag}_lora_bench_prof.err
db=$(ls gpurun_out/prof_$tag/*/*.db | head -1)
python scripts/rocpd_stats.py $db 6 gpurun_out/${tag}_lora_kernel_stats.md | head -40
(cd scripts && python rocpd_cutime.py ../$db 6 ../gpurun_out/${tag}_lora_cu_time.md | tail -12)
rm -rf gpurun_out/prof_$tag
