#!/bin/bash
# The decoder's own queue over one training step: how much of the span between the first qkv GEMM and the last expert GEMM is kernel time,
# how much is gaps between consecutive kernels of that queue (kernel-to-kernel dependency bubbles), by predecessor kernel.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/step_gaps; rm -rf $out; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $out/p -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-lora-line --roofline-steps 0 > $out/log 2>&1
python - <<'PY'
import csv, glob, collections, re
f = glob.glob("gpurun_out/step_gaps/p/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in csv.DictReader(open(f))]
rows.sort()
def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:44]
ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
a, b = ends[-2] + 1, ends[-1] + 1
seg = rows[a:b]
byq = collections.defaultdict(list)
for r in seg: byq[r[3]].append(r)
mainq = max(byq, key=lambda q: sum(e - s for s, e, n, _ in byq[q] if "gemm320" in n))
m = byq[mainq]
dec = [i for i, r in enumerate(m) if "gemm320_bf16_nt_kernel<1>" in r[2] or "gemm320_bf16_nt_kernel<3>" in r[2]]
i0, i1 = dec[0], dec[-1]
d = m[i0:i1 + 1]
span = d[-1][1] - d[0][0]
busy = sum(e - s for s, e, _, _ in d)
gaps = collections.defaultdict(lambda: [0, 0.0])
for x, y in zip(d, d[1:]):
    g = y[0] - x[1]
    if g > 0:
        k = short(x[2]) + " -> " + short(y[2])
        gaps[k][0] += 1; gaps[k][1] += g / 1e3
print(f"step span {(seg[-1][1] - seg[0][0]) / 1e6:.2f} ms; decoder queue {mainq}: {len(d)} kernels over {span / 1e6:.2f} ms, kernel time {busy / 1e6:.2f} ms, gaps {(span - busy) / 1e6:.2f} ms")
for k, (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {t / 1e3:6.2f} ms  x{n:3d}  avg {t / n:5.1f} us   {k}")
# what the decoder's queue does outside the decoder: from the previous step's last decoder GEMM to this step's first one
prev = [r for r in byq[mainq] if r[1] <= d[0][0]]
allq = rows[:a + 0]
pm = [r for r in rows if r[3] == mainq and r[0] < d[0][0]]
pdec = [i for i, r in enumerate(pm) if "gemm320_bf16_nt_kernel<3>" in r[2]]
if pdec:
    w = pm[pdec[-1] + 1:]
    t_a, t_b = pm[pdec[-1]][1], d[0][0]
    busy = sum(e - s for s, e, _, _ in w)
    print(f"between two decoders on queue {mainq}: {(t_b - t_a) / 1e6:.2f} ms, {len(w)} kernels, kernel time {busy / 1e6:.2f} ms, idle {(t_b - t_a - busy) / 1e6:.2f} ms")
    agg = collections.defaultdict(lambda: [0, 0.0])
    for s_, e_, n_, _ in w: agg[short(n_)][0] += 1; agg[short(n_)][1] += (e_ - s_) / 1e3
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
        print(f"  {t / 1e3:6.2f} ms  x{n:3d}   {k}")
    big = sorted(((y[0] - x[1], short(x[2]), short(y[2])) for x, y in zip([pm[pdec[-1]]] + w, w + [d[0]]) if y[0] - x[1] > 20000), reverse=True)[:8]
    for g, x, y in big: print(f"  gap {g / 1e3:7.1f} us after {x} before {y}")
PY
