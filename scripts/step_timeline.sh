#!/bin/bash
# where one training step's time goes on the MAIN stream: kernel trace of bench.py, last step, per queue; prints the main queue's phases and idle gaps.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/step_tl; rm -rf $out; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $out/p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timer > $out/log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/step_tl/p/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in csv.DictReader(open(f))]
rows.sort()
ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
a, b = ends[-2] + 1, ends[-1] + 1
seg = rows[a:b]
t0 = seg[0][0]
byq = collections.defaultdict(list)
for r in seg: byq[r[3]].append(r)
print("queues:", {q: (len(v), round(sum(e - s for s, e, _, _ in v) / 1e6, 2)) for q, v in byq.items()})
mainq = max(byq, key=lambda q: sum(e - s for s, e, _, _ in byq[q]))
m = byq[mainq]
print(f"step span {(seg[-1][1] - t0) / 1e6:.2f} ms; main queue {mainq}: first kernel at {(m[0][0] - t0) / 1e6:.2f} ms, last ends at {(m[-1][1] - t0) / 1e6:.2f} ms, busy {sum(e - s for s, e, _, _ in m) / 1e6:.2f} ms")
def short(n): return n.split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:48]
# phases: first gemm320<1> (qkv rope) = decoder start; last gemm (down) before lm_head = decoder end
first_qkv = next(r for r in m if "gemm320" in r[2] and "<1>" in r[2])
last_dec = [r for r in m if "gemm256v3" in r[2] or "gemm320" in r[2]][-1]
print(f"decoder starts at {(first_qkv[0] - t0) / 1e6:.2f} ms; last tile GEMM on the main queue ends at {(last_dec[1] - t0) / 1e6:.2f} ms")
print("main-queue gaps > 15 us:")
for i in range(len(m) - 1):
    g = m[i + 1][0] - m[i][1]
    if g > 15000: print(f"  {g / 1e3:7.1f} us at {(m[i][1] - t0) / 1e6:6.2f} ms after {short(m[i][2])} before {short(m[i + 1][2])}")
pre = [r for r in m if r[0] < first_qkv[0]]
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, q in pre: agg[short(n)][0] += 1; agg[short(n)][1] += (e - s) / 1e3
print("main queue before the decoder:", sorted(((k, v[0], round(v[1], 1)) for k, v in agg.items()), key=lambda x: -x[2])[:12])
post = [r for r in m if r[0] > last_dec[1]]
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, q in post: agg[short(n)][0] += 1; agg[short(n)][1] += (e - s) / 1e3
print("main queue after the decoder:", sorted(((k, v[0], round(v[1], 1)) for k, v in agg.items()), key=lambda x: -x[2])[:12])
PY
