#!/bin/bash
# The secondary measurements at HEAD in one call: BASELINE configs 1 and 4 (config_bench), evaluate()'s decode rate (MoE and dense),
# the LoRA configurations every shipped script trains.  One JSON line each under gpurun_out/.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for c in 1 4; do timeout 600 python scripts/config_bench.py --config $c 2>gpurun_out/r03_config$c.err | tail -1 > gpurun_out/r03_config$c.json; echo "config $c: $(cut -c1-300 gpurun_out/r03_config$c.json)"; done
timeout 600 python scripts/decode_bench.py 2>gpurun_out/r03_decode_moe.err | tail -1 > gpurun_out/r03_decode_moe.json; echo "decode moe: $(cut -c1-300 gpurun_out/r03_decode_moe.json)"
timeout 600 python scripts/decode_bench.py --dense 2>gpurun_out/r03_decode_dense.err | tail -1 > gpurun_out/r03_decode_dense.json; echo "decode dense: $(cut -c1-300 gpurun_out/r03_decode_dense.json)"
i=0
while read -r line; do
  case "$line" in python*) i=$((i+1)); timeout 900 bash -c "$line" > gpurun_out/r03_lora_cfg$i.json; echo "lora cfg $i: $(cut -c1-260 gpurun_out/r03_lora_cfg$i.json)";; esac
done < scripts/lora_configs.sh
