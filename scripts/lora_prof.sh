#!/bin/bash
# rocprofv3 kernel table of the LoRA training step (scripts/lora_bench.py).  Usage: bash scripts/lora_prof.sh <tag> [lora_bench args]
tag=${1:-x}; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python scripts/lora_bench.py "$@" > gpurun_out/${tag}_lora_bench.json 2> gpurun_out/${tag}_lora_bench.err; tail -2 gpurun_out/${tag}_lora_bench.err; cat gpurun_out/${tag}_lora_bench.json
rm -rf gpurun_out/prof_$tag
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -- python scripts/lora_bench.py --steps 3 --warmup 1 "$@" > /dev/null 2> gpurun_out/${tag}_lora_prof.err
db=$(ls gpurun_out/prof_$tag/*/*.db | head -1); python scripts/rocpd_stats.py $db 4 gpurun_out/${tag}_lora_kernel_stats.md > /dev/null; sed -n 5,40p gpurun_out/${tag}_lora_kernel_stats.md | cut -c1-160
rm -rf gpurun_out/prof_$tag
