#!/bin/bash
# SQ / LDS counters of the attention backward kernels on the Llama shape.  Usage: bash scripts/attn_bwd_pmc.sh
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/attn_bwd_pmc; rm -rf $out; mkdir -p $out
cat > /tmp/attn_bwd_one.py <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from medplib_amd import ops
dev = torch.device("cuda:0")
B, S, H, D = 8, 639, 32, 128
qkv = torch.randn(B, S, 3, H, D, device=dev).to(torch.bfloat16)
d_out = torch.randn(B, S, H * D, device=dev).to(torch.bfloat16)
out, lse2 = ops.attention_fwd_lse(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], causal=True)
for _ in range(4):
    ops.attention_bwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], out, d_out, lse2, causal=True)
torch.cuda.synchronize()
PY
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d $out/p1 -- python /tmp/attn_bwd_one.py > $out/p1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $out/p2 -- python /tmp/attn_bwd_one.py > $out/p2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $out/p3 -- python /tmp/attn_bwd_one.py > $out/p3.log 2>&1
python - <<'PY'
import csv, glob, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/attn_bwd_pmc/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn_bwd" in r["Kernel_Name"]:
            res[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kname, cs in res.items():
    print(kname)
    for k, v in sorted(cs.items()):
        print(f"   {k:28s} {sum(v) / len(v):16.1f}")
PY
python - <<'PY'
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("gpurun_out/attn_bwd_pmc/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    if "attn" in k: print(f"{k:72s} n={len(v)} avg {sum(v)/len(v):8.1f} us  min {min(v):8.1f}")
PY
