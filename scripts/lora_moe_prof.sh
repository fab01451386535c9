#!/bin/bash
# rocprofv3 kernel table of the stage-IV LoRA configuration (MoE E = 2, per-expert adapters).  Usage: bash scripts/lora_moe_prof.sh <tag>
tag=${1:-x}
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/prof_$tag
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -- python scripts/lora_bench.py --steps 3 --warmup 1 --moe --targets gate_proj,up_proj,down_proj,q_proj,v_proj --sft mask_decoder,text_hidden_fcs,lm_head,embed_tokens,wg > /dev/null 2> gpurun_out/${tag}_prof.err
db=$(ls gpurun_out/prof_$tag/*/*.db | head -1); python scripts/rocpd_stats.py $db 4 gpurun_out/${tag}_kernel_stats.md > /dev/null; sed -n 1,42p gpurun_out/${tag}_kernel_stats.md | cut -c1-150
rm -rf gpurun_out/prof_$tag
