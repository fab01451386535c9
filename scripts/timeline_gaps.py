"""GPU idle time of the training step from a rocprofv3 --kernel-trace CSV: union of the kernel intervals of the last step vs its
span, the longest gaps and what precedes them.  python scripts/timeline_gaps.py <kernel_trace.csv> [steps]"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", ""))))
rows.sort()
# the step boundary: adamw kernels end a step
ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
if len(ends) < 2:
    sys.exit("need >= 2 optimizer steps in the trace")
a, b = ends[-2] + 1, ends[-1] + 1
seg = rows[a:b]
t0, t1 = seg[0][0], max(r[1] for r in seg)
busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
gaps = []
last_name = seg[0][2]
for s, e, n, q in seg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name[:60], n[:60]))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e:
        last_name = n
busy += cur_e - cur_s
print(f"step span {(t1 - t0) / 1e6:.2f} ms, kernels running {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms in {len(gaps)} gaps, {len(seg)} launches")
gaps.sort(reverse=True)
for g, p, n in gaps[:12]:
    print(f"  gap {g / 1e3:8.1f} us   after {p}   before {n}")
small = sum(g for g, _, _ in gaps if g < 5000)
print(f"  gaps < 5 us: {sum(1 for g, _, _ in gaps if g < 5000)} totalling {small / 1e6:.2f} ms")
