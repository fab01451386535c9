"""Per-position timeline of a decode layer from a rocprofv3 --kernel-trace rocpd database: the dispatches of the replayed decode graph in
start order, folded by their position inside the layer (the launch sequence repeats every layer and every token): average duration per
position, average gap to the next dispatch.  python scripts/decode_timeline.py <db> <out.md>"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:60]


def main(db_path, out):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
    names = [short(r[0]) for r in rows]
    # the layer's first launch: the norm-folded qkv GEMV
    first = [i for i, n in enumerate(names) if n.startswith("gemv_rmsnorm_rope_kernel")]
    if len(first) < 64:
        print("no decode layers found"); return
    period = first[1] - first[0]
    pos = defaultdict(lambda: [0, 0.0, 0.0, ""])
    n_layers = 0
    for a, b in zip(first[:-1], first[1:]):
        if b - a != period:
            continue                                             # a token boundary (final norm, lm_head, argmax, embedding) sits between
        n_layers += 1
        for p in range(period):
            r, nxt = rows[a + p], rows[a + p + 1]
            e = pos[p]
            e[0] += 1; e[1] += (r[2] - r[1]) / 1e3; e[2] += (nxt[1] - r[2]) / 1e3
            e[3] = f"{names[a + p]} grid {r[3] // max(r[4], 1)} x {r[4]}"
    lines = [f"# decode layer timeline ({db_path.split('/')[-1]}; {n_layers} layer passes folded; us)", "",
             "| position | kernel | avg duration us | avg gap to next us |", "|---|---|---|---|"]
    tot_d = tot_g = 0.0
    for p in range(period):
        c, d, g, nm = pos[p]
        lines.append(f"| {p} | `{nm}` | {d / c:.2f} | {g / c:.2f} |")
        tot_d += d / c; tot_g += g / c
    lines.append(f"| **layer** | | {tot_d:.2f} | {tot_g:.2f} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
