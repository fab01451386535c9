#!/bin/bash
# round 6 closing run: the driver's bench command, kernel table + CU x time + PMC passes of the headline step, LoRA kernel table, decode timelines
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
tag=${1:-r06}
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("headline", d["ms_per_step"], "value", d["value"], "host", d.get("host_issue_ms_per_step"), "frac", r["frac"], "clk", r.get("effective_clock_ghz"), r.get("frac_at_effective_clock"), "traffic", r.get("traffic"))
l=d.get("lora_stage3") or {}
print("lora", l.get("ms_per_step"), "host", l.get("host_issue_ms_per_step"))
u=d.get("roofline_upsampler") or {}
print("ups", u.get("sam1024"), u.get("sam256"), {k:v for k,v in u.items() if "copy" in k})
print("decode", {k: (v.get("ms_per_token"), v.get("frac_of_8TBps")) for k, v in (d.get("decode") or {}).items() if isinstance(v, dict)})
print("configs", json.dumps(d.get("configs"))[:600])
p=d.get("parity") or {}
print({k:p.get(k) for k in ("hidden_p999_rel_err","hidden_bad_rows","flipped_tokens_total","rows_agreeing_in_every_layer","max_abs_dloss_over_10","distinct_weights")}, (p.get("mask") or {}).get("max_abs_dlogit"))
print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("mask_tail"), d.get("dp_bucket"))
PY
bash scripts/r04_profiles.sh $tag > gpurun_out/${tag}_profiles.log 2>&1; tail -5 gpurun_out/${tag}_profiles.log
bash scripts/r05_lora_profiles.sh $tag > /dev/null 2>&1; head -14 gpurun_out/${tag}_lora_kernel_stats.md | cut -c1-130
bash scripts/r05_decode_prof.sh $tag 2>&1 | grep -E "metric" | cut -c1-200
