cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
python bench.py > gpurun_out/r05m_bench.json 2> gpurun_out/r05m_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05m_bench.json").read().strip().splitlines()[-1])
print("headline", d["ms_per_step"], "value", d["value"], "frac", d["roofline"]["frac"], "lora", (d.get("lora_stage3") or {}).get("ms_per_step"), (d.get("lora_stage3") or {}).get("last_layer_mlp_rows"),
      "ups", (d.get("roofline_upsampler") or {}).get("sam1024", {}).get("frac"), "decode", {k: v.get("ms_per_token") for k, v in (d.get("decode") or {}).items() if isinstance(v, dict)})
print(d["config"].get("last_layer_mlp_rows"), d.get("model_tflops_per_gpu"), d["roofline"].get("traffic"))
PY
bash scripts/r05_lora_profiles.sh r05m > /dev/null 2>&1; head -40 gpurun_out/r05m_lora_kernel_stats.md | cut -c1-120 | grep -E "window|rope|lora_down_finish|swiglu_bwd|rmsnorm_bwd|tn_skinny|sum"
