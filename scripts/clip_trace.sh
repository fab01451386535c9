#!/bin/bash
# kernel timeline of the CLIP tower alone (batch 8): per-kernel table of one pass, the span it covers and the idle time inside it.  bash scripts/clip_trace.sh
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/clip_trace; rm -rf $out; mkdir -p $out
cat > /tmp/clip_one.py <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from medplib_amd.model.config import MedPLIBConfig
from medplib_amd.model.medplib import MedPLIBForCausalLM
dev = torch.device("cuda:0")
model = MedPLIBForCausalLM(MedPLIBConfig.medplib_7b(num_hidden_layers=1), device=dev).eval()
img = torch.randn(8, 3, 336, 336).to(dev).to(torch.bfloat16)
marker = torch.zeros(1, device=dev)
with torch.no_grad():
    for _ in range(4):
        marker.add_(1.0); torch.cuda.synchronize()
        model.model.vision_tower.encode_images(img)
        torch.cuda.synchronize()
PY
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/p -- python /tmp/clip_one.py > $out/log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/clip_trace/p/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
marks = [i for i, r in enumerate(rows) if "add" in r[2].lower() and "elementwise" in r[2].lower() and r[1] - r[0] < 20000]
# last pass = kernels after the last marker
seg = rows[marks[-1] + 1:]
span = seg[-1][1] - seg[0][0]; busy = sum(e - s for s, e, _ in seg)
print(f"{len(seg)} kernels, span {span/1e3:.1f} us, sum of durations {busy/1e3:.1f} us, idle {(span-busy)/1e3:.1f} us")
by = collections.OrderedDict()
for s, e, n in seg:
    k = n[:90]; by.setdefault(k, []).append((e - s) / 1e3)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v):8.1f} us  n={len(v):3d}  avg {sum(v)/len(v):6.1f}  {k}")
gaps = sorted(((seg[i+1][0] - seg[i][1]) / 1e3 for i in range(len(seg) - 1)), reverse=True)
print("largest gaps (us):", [round(g, 1) for g in gaps[:8]], "median", round(sorted(gaps)[len(gaps)//2], 2))
PY
