#!/bin/bash
# round 6 mid-round checkpoint: the whole GPU suite (one process per module), then the driver's bench command
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-r06e}
bash scripts/gpu_tests.sh > gpurun_out/${tag}_tests.log 2>&1; echo "tests rc=$?"; grep -E "^==|passed|failed" gpurun_out/${tag}_tests.log | paste - - | cut -c1-140
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_bench.json").read().strip().splitlines()[-1])
print("headline", d["ms_per_step"], "value", d["value"], "host", d.get("host_issue_ms_per_step"), "frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"))
l=d.get("lora_stage3") or {}
print("lora", l.get("ms_per_step"), "host", l.get("host_issue_ms_per_step"))
u=d.get("roofline_upsampler") or {}
print("ups", {k:(v.get("frac") if isinstance(v,dict) else v) for k,v in u.items()})
print("decode", {k: (v.get("ms_per_token"), v.get("frac_of_8TBps")) for k, v in (d.get("decode") or {}).items() if isinstance(v, dict)})
print("configs", {k:(v.get("frac_of_mfma_peak") if isinstance(v,dict) else v) for k,v in (d.get("configs") or {}).items()})
p=d.get("parity") or {}
print({k:p.get(k) for k in ("hidden_p999_rel_err","hidden_bad_rows","flipped_tokens_total","rows_agreeing_in_every_layer","max_abs_dloss_over_10")}, (p.get("mask") or {}).get("max_abs_dlogit"))
PY
