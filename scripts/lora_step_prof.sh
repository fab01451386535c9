#!/bin/bash
# rocprofv3 kernel table of bench.py --lora (the shipped stage-III LoRA configuration).  Usage: bash scripts/lora_step_prof.sh <tag>
tag=${1:-x}
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/prof_$tag
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -- python bench.py --lora --steps 3 --warmup 1 --no-kernel-timer > /dev/null 2> gpurun_out/${tag}_lora_prof.err
db=$(ls gpurun_out/prof_$tag/*/*.db | head -1); python scripts/rocpd_stats.py $db 4 gpurun_out/${tag}_lora_kernel_stats.md > /dev/null; sed -n 1,45p gpurun_out/${tag}_lora_kernel_stats.md | cut -c1-150
rm -rf gpurun_out/prof_$tag
