#!/bin/bash
# Non-default configurations of the bench, one line each: a robustness sweep (every line must finish at a sane rate).
set -u
mkdir -p gpurun_out
run() {
  tag=$1; shift
  timeout 400 python bench.py --no-cpu-baseline --no-lora-line --roofline-steps 0 --steps 6 --warmup 3 "$@" > gpurun_out/sweep_$tag.json 2> gpurun_out/sweep_$tag.err
  rc=$?
  python - "$tag" $rc <<'PY'
import json, sys
tag, rc = sys.argv[1], sys.argv[2]
try:
    d = json.load(open(f"gpurun_out/sweep_{tag}.json"))
    print(tag, "rc", rc, d["value"], d["unit"], d["ms_per_step"], "ms", d.get("loss_last"))
except Exception as e:
    print(tag, "rc", rc, "NO LINE", e)
    print(open(f"gpurun_out/sweep_{tag}.err").read()[-600:])
PY
}
run b1 --batch 1
run b2 --batch 2
run b5 --batch 5
run b20 --batch 20
run b24 --batch 24
run lora_b4 --lora --batch 4
run lora_b12 --lora --batch 12
run lora_b16 --lora --batch 16
run l8_b16 --layers 8 --batch 16
run nss_b16 --no-side-streams --batch 16
