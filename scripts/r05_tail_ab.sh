#!/bin/bash
# A/B of the mask-tail program in the step: headline (tail hidden on its stream) and LoRA (tail exposed), program off / on at several grids
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { # tag, env...
  tag=$1; shift
  env "$@" python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-lora-line --no-secondary --roofline-steps 0 $EXTRA > gpurun_out/tail_ab_$tag.json 2> gpurun_out/tail_ab_$tag.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/tail_ab_$tag.json").read().strip().splitlines()[-1])
print("$tag", d["ms_per_step"], "ms/step; host issue", d.get("host_issue_ms_per_step"), "tail_backward_us", (d.get("dp_bucket") or {}).get("tail_backward_us"))
PY
}
EXTRA=""
run head_off MP_TAIL_PROGRAM=0
run head_on256 MP_TAIL_PROGRAM=1 MP_TAIL_GRID=256
run head_on64 MP_TAIL_PROGRAM=1 MP_TAIL_GRID=64
run head_on32 MP_TAIL_PROGRAM=1 MP_TAIL_GRID=32
run head_off2 MP_TAIL_PROGRAM=0
EXTRA="--lora"
run lora_off MP_TAIL_PROGRAM=0
run lora_on256 MP_TAIL_PROGRAM=1 MP_TAIL_GRID=256
run lora_off2 MP_TAIL_PROGRAM=0
