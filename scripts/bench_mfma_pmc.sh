#!/bin/bash
# MFMA-pipe occupancy of the training step's GEMM / attention kernels from PMC counters (kernel-trace only, its own run):
# SQ_VALU_MFMA_BUSY_CYCLES (cycles the MFMA pipe is busy, summed over the chip's 1024 SIMDs) against GRBM_GUI_ACTIVE (the launch's
# duration in shader clocks, SUMMED over the 8 XCDs): util = MFMA_BUSY / (GUI_ACTIVE / 8 x 1024).  -> gpurun_out/${TAG:-r02}_mfma_pmc.json (copy to profiles/).
export TMPDIR=/tmp
out=gpurun_out/mfma_pmc; rm -rf $out; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $out/p1 -- \
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-lora-line --no-secondary --no-kernel-timer > $out/p1.log 2>&1
python - <<'PY'
import csv, glob, collections, json
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/mfma_pmc/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        fam = ("gemm256v3" if "gemm256v3" in n else "gemm320" if "gemm320" in n else "gemm128" if "gemm_bf16_nt_kernel" in n else "attn_fwd2" if "attn_fwd2" in n else None)
        if fam:
            res[fam][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "gemm320" in n and any("kernel<%d>" % e in n for e in (1, 2, 3)):      # epilogue families 1-3 (qkv + RoPE, gate|up, down): decoder launches only
            res["gemm320_decoder"][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"note": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 over bench.py (3 steps, kernels "
               "serialised by the counter collection); per-launch averages; mfma_pipe_util = MFMA_BUSY / (GUI_ACTIVE / 8 XCDs x 1024 SIMDs); "
               "mops_flops = MOPS_BF16 x 512 (one MOP = 512 flops) when the counter is populated"}
for fam, cs in res.items():
    a = {k: sum(v) / max(len(v), 1) for k, v in cs.items()}
    busy, act = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), a.get("GRBM_GUI_ACTIVE", 0.0)
    out[fam] = {"launches": len(cs.get("GRBM_GUI_ACTIVE", [])), "avg": {k: round(v, 1) for k, v in a.items()},
                "shader_clocks_per_launch": round(act / 8), "mfma_pipe_util": round(busy / (act / 8 * 1024), 4) if act else None}
json.dump(out, open("gpurun_out/" + __import__("os").environ.get("TAG", "r02") + "_mfma_pmc.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
PY
