#!/bin/bash
# round 5 closing run: GPU suite, the driver's bench command, kernel table + CU x time + PMC passes of the headline step, LoRA kernel table
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out

python bench.py > gpurun_out/r05k_bench.json 2> gpurun_out/r05k_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05k_bench.json").read().strip().splitlines()[-1])
print("headline", d["ms_per_step"], "host", d.get("host_issue_ms_per_step"), "frac", d["roofline"]["frac"], "lora", (d.get("lora_stage3") or {}).get("ms_per_step"),
      "ups", (d.get("roofline_upsampler") or {}).get("sam1024", {}).get("frac"), "decode", {k: v.get("ms_per_token") for k, v in (d.get("decode") or {}).items() if isinstance(v, dict)})
print("tail", d.get("mask_tail"), d.get("dp_bucket"))
p=d.get("parity") or {}
print({k:p.get(k) for k in ("hidden_p999_rel_err","hidden_bad_rows","flipped_tokens_total","routing_layer_local","max_abs_dloss_over_10")}, p.get("mask",{}).get("max_abs_dlogit"))
PY
bash scripts/r04_profiles.sh r05k > gpurun_out/r05k_profiles.log 2>&1; tail -5 gpurun_out/r05k_profiles.log
bash scripts/r05_lora_profiles.sh r05k > /dev/null 2>&1; head -14 gpurun_out/r05k_lora_kernel_stats.md | cut -c1-130
bash scripts/r05_decode_prof.sh r05k 2>&1 | grep -E "metric" | cut -c1-200
python scripts/r05_skinny_bench.py > gpurun_out/r05k_skinny_bench.txt 2>&1; tail -22 gpurun_out/r05k_skinny_bench.txt
