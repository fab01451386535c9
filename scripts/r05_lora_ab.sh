#!/bin/bash
# LoRA stage-III step: this round's switches one by one against the round-4 form, same box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { tag=$1; shift
  env "$@" python bench.py --lora --steps 10 --warmup 3 --no-kernel-timer > gpurun_out/lab_$tag.json 2> gpurun_out/lab_$tag.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/lab_$tag.json").read().strip().splitlines()[-1])
print("$tag", d["ms_per_step"], "ms/step; host issue", d.get("host_issue_ms_per_step"))
PY
}
run r4form MP_TAIL_PROGRAM=0 MP_FUSE_UP_SWIGLU=0 MP_LORA_PACK_BATCHED=0 MP_LORA_UNPACK_PARTIALS=0
run all_on
run no_partials MP_LORA_UNPACK_PARTIALS=0
run r4form2 MP_TAIL_PROGRAM=0 MP_FUSE_UP_SWIGLU=0 MP_LORA_PACK_BATCHED=0 MP_LORA_UNPACK_PARTIALS=0
run all_on2
