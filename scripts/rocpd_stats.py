"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats) into a markdown table: per-kernel calls, total, average, share."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    if "distribution_elementwise_grid_stride_kernel" in name:
        return "at::native::distribution_elementwise (torch.randn weight init, outside the timed step)"
    if "vectorized_elementwise_kernel" in name or "elementwise_kernel_manual_unroll" in name:
        m = re.search(r"(MulFunctor|bfloat16_copy|CUDAFunctor_add|FillFunctor|direct_copy)", name)
        return "at::native elementwise (" + (m.group(1) if m else "misc") + ") [init / autograd plumbing]"
    name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)
    return name.split("(")[0][:90]


def timed_region(db):
    """(start, end) of the dispatch window bench.py marked with mp_profile_marker tags 1 / 2 (MP_BENCH_MARKERS=1), or None."""
    marks = list(db.execute("select grid_x / workgroup_x, start, end from kernels where name like '%mp_profile_marker_kernel%' order by start"))
    b = [m for m in marks if m[0] == 1]
    e = [m for m in marks if m[0] == 2]
    return (b[0][2], e[0][1]) if b and e else None


def main(db_path, steps, out):
    """steps = the steps the table's "ms / step" column divides by: with cut marks in the trace, the TIMED steps (only dispatches between
    the marks are listed, so the column sums to the step: no weight initialisation, no warm-up, no micro-benchmark loops, no LoRA / parity
    legs); without marks, every dispatch of the run (round <= 3 tables) over steps incl. warm-up."""
    db = sqlite3.connect(db_path)
    win = timed_region(db)
    if win is not None:
        rows = list(db.execute("select name, count(*), sum(end - start) / 1000.0 from kernels where start >= ? and end <= ? group by name", win))
        total = sum(r[2] for r in rows) or 1.0
        rows = [(n, c, t, t / c, 100.0 * t / total) for n, c, t in rows]
        head = (f"# rocprofv3 --kernel-trace summary of the TIMED steps only ({db_path.split('/')[-1]}; dispatches between bench.py's cut marks, "
                f"{steps} steps, window {(win[1] - win[0]) / 1e6 / steps:.2f} ms per step; the column `ms / step` sums to the GPU time of a step over all queues)")
    else:
        rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        head = f"# rocprofv3 --kernel-trace --stats summary ({db_path.split('/')[-1]}; {steps} steps incl. warm-up; the whole run: init and micro-benchmarks included)"
    agg = {}
    for n, c, t, a, p in rows:
        k = short(n)
        e = agg.setdefault(k, [0, 0.0, 0.0])
        e[0] += c; e[1] += t; e[2] += p
    lines = [head, "", "| kernel | calls | total ms | avg us | % GPU time | ms / step |", "|---|---|---|---|---|---|"]
    for k, (c, t, p) in sorted(agg.items(), key=lambda kv: -kv[1][1]):           # every kernel of the run: the short ones are part of the path too
        lines.append(f"| `{k}` | {c} | {t / 1e3:.2f} | {t / c:.1f} | {p:.2f} | {t / 1e3 / steps:.3f} |")
    if win is not None:
        lines.append(f"| **sum** | {sum(v[0] for v in agg.values())} | {sum(v[1] for v in agg.values()) / 1e3:.2f} | | 100.00 | {sum(v[1] for v in agg.values()) / 1e3 / steps:.3f} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:14]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3])
