#!/bin/bash
# The last layer's MLP on the read rows only (MP_PRUNE_LAST_MLP): tests, then the headline and the LoRA step with it off / on, same box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_prune_last_mlp.py -x -q -s -m gpu 2>&1 | tail -25 > gpurun_out/prune_tests.log
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "lora or model_forward or lisa_golden or full_depth" 2>&1 | tail -15 >> gpurun_out/prune_tests.log
run() { tag=$1; shift
  env "$@" python bench.py $ARGS --steps 12 --warmup 3 --no-kernel-timer --no-cpu-baseline --no-lora-line --no-secondary --no-live-traffic > gpurun_out/prune_$tag.json 2> gpurun_out/prune_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/prune_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["ms_per_step"], "ms/step; host issue", d.get("host_issue_ms_per_step"))
except Exception as e:
    print("$tag failed", e)
PY
}
ARGS="--lora";  run lora_off MP_PRUNE_LAST_MLP=0; run lora_on; run lora_off2 MP_PRUNE_LAST_MLP=0; run lora_on2
ARGS="";        run moe_off MP_PRUNE_LAST_MLP=0;  run moe_on;  run moe_off2 MP_PRUNE_LAST_MLP=0;  run moe_on2
