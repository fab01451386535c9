#!/bin/bash
# Are the 320-row tile kernel's PMC-pass durations representative?  Kernel durations of scripts/gemm_tile_ab.py with and without counter collection.
export TMPDIR=/tmp
out=gpurun_out/pmc_tile; rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/plain -- python scripts/gemm_tile_ab.py > $out/plain.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc -- python scripts/gemm_tile_ab.py > $out/pmc.log 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("plain", "pmc"):
    by = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_tile/{d}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            fam = "gemm256v3" if "gemm256v3" in n else "gemm320" if "gemm320" in n else None
            if fam: by[(fam, r.get("Grid_Size_X", r.get("Grid_Size", "")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(by.items()): print(d, k, len(v), "avg us %.1f" % (sum(v) / len(v)))
PY
