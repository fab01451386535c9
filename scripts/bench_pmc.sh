#!/bin/bash
# HBM traffic of the training step's kernels from PMC counters: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes
# (kernel-trace only, no sys-trace), as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes.  Summarised per kernel family into
# gpurun_out/${TAG:-r02}_hbm_traffic.json (copy to profiles/).  Units: FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; on
# gfx950 FETCH_SIZE counts 128-B read requests as 64 B, so read bytes = 2 x FETCH_SIZE (guide's correction); WRITE_SIZE is
# uncalibrated and reported raw.
export TMPDIR=/tmp
out=gpurun_out/bench_pmc; rm -rf $out; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-lora-line --no-secondary --no-kernel-timer > $out/$c.log 2>&1
done
python - <<'PY'
import csv, glob, collections, json, re
res = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/bench_pmc/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            n = r["Kernel_Name"]
            fam = ("gemm256v3" if "gemm256v3" in n else "gemm320" if "gemm320" in n else "gemm128" if "gemm_bf16_nt_kernel" in n else "attn_fwd2" if "attn_fwd2" in n else
                   "rmsnorm" if "rmsnorm" in n else "moe_combine" if "moe_combine" in n else "rope" if "rope_qk" in n else None)
            if fam:
                res[fam][c].append(float(r["Counter_Value"]))
            if "gemm320" in n and any("kernel<%d>" % e in n for e in (1, 2, 3)):      # epilogue families 1-3 (qkv + RoPE, gate|up, down): decoder launches only
                res["gemm320_decoder"][c].append(float(r["Counter_Value"]))
import sys; sys.path.insert(0, ".")
from bench import kernel_source_sha
out = {"kernel_source_sha": kernel_source_sha(), "note": "rocprofv3 --kernel-trace --pmc <counter> over bench.py (3 steps); per-launch averages; read_bytes = 2 x FETCH_SIZE KiB (gfx950 correction, MI355X_MICROARCH.md HBM section), write_bytes = WRITE_SIZE KiB raw (uncalibrated)"}
for fam, cs in res.items():
    f, w = cs.get("FETCH_SIZE", []), cs.get("WRITE_SIZE", [])
    out[fam] = {"launches": len(f), "read_bytes_per_launch": round(2 * 1024 * sum(f) / max(len(f), 1)), "write_bytes_per_launch": round(1024 * sum(w) / max(len(w), 1))}
json.dump(out, open("gpurun_out/" + __import__("os").environ.get("TAG", "r02") + "_hbm_traffic.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
PY
