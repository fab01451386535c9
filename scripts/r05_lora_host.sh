#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for mode in 0 1; do
  MP_TAIL_PROGRAM=$mode python bench.py --lora --steps 8 --warmup 3 --no-kernel-timer > gpurun_out/lora_host_$mode.json 2> gpurun_out/lora_host_$mode.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/lora_host_$mode.json").read().strip().splitlines()[-1])
print("program=$mode", d["ms_per_step"], "ms/step; host issue", d.get("host_issue_ms_per_step"), (d.get("dp_bucket") or {}).get("tail_backward_us"))
PY
done
MP_TAIL_PROGRAM=1 bash scripts/r05_lora_profiles.sh r05b > /dev/null 2>&1
head -30 gpurun_out/r05b_lora_kernel_stats.md | cut -c1-150
