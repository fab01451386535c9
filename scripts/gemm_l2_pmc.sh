#!/bin/bash
# L2 (TCC) hit / miss counts of the 256x256 GEMM on the Llama shapes (kernel-trace + pmc only).  Usage: bash scripts/gemm_l2_pmc.sh
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/gemm_l2; rm -rf $out; mkdir -p $out
cat > /tmp/gemm_one.py <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from medplib_amd import ops
dev = torch.device("cuda:0")
for (M, N, K) in [(5112, 12288, 4096), (5112, 4096, 4096), (5112, 22016, 4096), (5112, 4096, 11008), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    ws = [(torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(3)]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for i in range(4):
        ops.gemm(a, ws[i % 3], out=out)
    torch.cuda.synchronize()
PY
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/p1 -- python /tmp/gemm_one.py > $out/p1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d $out/p2 -- python /tmp/gemm_one.py > $out/p2.log 2>&1
python - <<'PY'
import csv, glob, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/gemm_l2/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm256v3" in r["Kernel_Name"]:
            res[(r["Grid_Size"], r.get("Dispatch_Id", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for (grid, _), cs in res.items():
    for k, v in cs.items():
        agg[grid][k] += v
for grid, cs in sorted(agg.items()):
    a = {k: sum(v) / len(v) for k, v in cs.items()}
    hit, miss = a.get("TCC_HIT_sum", 0), a.get("TCC_MISS_sum", 0)
    print(f"grid {grid}: " + "  ".join(f"{k} {v:.3e}" for k, v in sorted(a.items())) + (f"  hit rate {hit / (hit + miss):.3f}" if hit + miss else ""))
PY
tail -3 $out/p1.log
