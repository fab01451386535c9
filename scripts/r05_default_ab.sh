#!/bin/bash
# the driver's default command shape (5 steps after 2) with the tail program off / on, same box, alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { tag=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-lora-line --no-secondary $EXTRA > gpurun_out/dab_$tag.json 2> gpurun_out/dab_$tag.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/dab_$tag.json").read().strip().splitlines()[-1])
print("$tag", d["ms_per_step"], "ms/step; host issue", d.get("host_issue_ms_per_step"), "frac", (d.get("roofline") or {}).get("frac"), "tail_bwd_us", (d.get("dp_bucket") or {}).get("tail_backward_us"))
PY
}
EXTRA=""
run off1 MP_TAIL_PROGRAM=0
run on1 MP_TAIL_PROGRAM=1
run off2 MP_TAIL_PROGRAM=0
run on2 MP_TAIL_PROGRAM=1
run on_nofuse MP_TAIL_PROGRAM=1 MP_TAIL_FUSED_UPSAMPLER=0
EXTRA="--lora --steps 8 --warmup 3"
run lora_base MP_TAIL_PROGRAM=0 MP_FUSE_UP_SWIGLU=0
run lora_new MP_TAIL_PROGRAM=1 MP_FUSE_UP_SWIGLU=1
run lora_new_noprog MP_TAIL_PROGRAM=0 MP_FUSE_UP_SWIGLU=1
