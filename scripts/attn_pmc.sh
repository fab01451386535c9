#!/bin/bash
# SQ counters of the attention forward kernel on the Llama shape (kernel-trace + pmc only).  Usage: bash scripts/attn_pmc.sh
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/attn_pmc; rm -rf $out; mkdir -p $out
cat > /tmp/attn_one.py <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from medplib_amd import ops
dev = torch.device("cuda:0")
B, S, H, D = 8, 639, 32, 128
qkv = torch.randn(B, S, 3, H, D, device=dev).to(torch.bfloat16)
out = torch.empty(B, S, H * D, dtype=torch.bfloat16, device=dev)
for _ in range(6):
    ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], out=out, causal=True)
torch.cuda.synchronize()
PY
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d $out/p1 -- python /tmp/attn_one.py > $out/p1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $out/p2 -- python /tmp/attn_one.py > $out/p2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY --output-format csv -d $out/p3 -- python /tmp/attn_one.py > $out/p3.log 2>&1
python - <<'PY'
import csv, glob, collections
res = collections.defaultdict(list)
for f in glob.glob("gpurun_out/attn_pmc/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn_fwd2" in r["Kernel_Name"]:
            res[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(res.items()):
    print(f"{k:32s} {sum(v) / len(v):16.1f}  (n={len(v)})")
PY
