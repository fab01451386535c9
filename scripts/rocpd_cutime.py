"""What each kernel of a traced step costs in CU x time — the currency of work that runs BESIDE the decoder's GEMMs (DESIGN.md section 10:
nothing co-resides with a 147 KB / 245-register GEMM workgroup, so a side stream's workgroup displaces decoder work for its own duration).
From a rocprofv3 --kernel-trace rocpd database, for every dispatch between bench.py's cut marks (MP_BENCH_MARKERS=1):
    chip_us = duration x min(1, workgroups / (256 CUs x workgroups resident per CU))
with the residency from the dispatch's own LDS bytes, register counts and workgroup size; summed per (queue, kernel).
    python scripts/rocpd_cutime.py <db> <steps> <out.md>"""
import re
import sqlite3
import sys

from rocpd_stats import short, timed_region

CUS = 256


def resident(lds, vgpr, agpr, wg_threads):
    waves = max(1, (wg_threads + 63) // 64)
    regs = max(8, -(-(vgpr + agpr) // 8) * 8)
    by_regs = (min(8, 512 // regs) * 4) // waves if regs <= 512 else 0
    by_lds = (160 * 1024) // lds if lds > 0 else 32
    by_waves = 32 // waves
    return max(1, min(by_regs if by_regs > 0 else 1, by_lds, by_waves))


def main(db_path, steps, out):
    db = sqlite3.connect(db_path)
    win = timed_region(db)
    where, args = ("where start >= ? and end <= ?", win) if win else ("", ())
    rows = db.execute(f"select name, queue, start, end, grid_x * grid_y * grid_z, workgroup_x * workgroup_y * workgroup_z, lds_size, vgpr_count, "
                      f"accum_vgpr_count from kernels {where}", args)
    agg = {}
    for name, queue, s, e, grid, wg, lds, vg, ag in rows:
        n_wg = max(1, grid // max(wg, 1))
        r = resident(lds or 0, vg or 0, ag or 0, wg)
        dur = (e - s) / 1e3
        k = (queue, short(name))
        a = agg.setdefault(k, [0, 0.0, 0.0, 0, r])
        a[0] += 1; a[1] += dur; a[2] += dur * min(1.0, n_wg / (CUS * r)); a[3] += n_wg
    lines = [f"# CU x time per kernel and queue ({db_path.split('/')[-1]}; {'timed steps only' if win else 'whole run'}, {steps} steps)", "",
             "chip us = duration x min(1, workgroups / (256 x resident workgroups per CU)): what a kernel takes from whatever else wants the CUs.", "",
             "| queue | kernel | calls / step | avg us | avg workgroups | resident / CU | duration ms / step | chip ms / step |", "|---|---|---|---|---|---|---|---|"]
    per_q = {}
    for (q, k), (c, d, cu, wgs, r) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
        lines.append(f"| {q} | `{k}` | {c / steps:.1f} | {d / c:.1f} | {wgs / c:.0f} | {r} | {d / 1e3 / steps:.3f} | {cu / 1e3 / steps:.3f} |")
        pq = per_q.setdefault(q, [0.0, 0.0]); pq[0] += d; pq[1] += cu
    lines += ["", "| queue | duration ms / step | chip ms / step |", "|---|---|---|"]
    for q, (d, cu) in sorted(per_q.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| {q} | {d / 1e3 / steps:.3f} | {cu / 1e3 / steps:.3f} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:40]))


if __name__ == "__main__":
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3])
