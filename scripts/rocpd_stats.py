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


def main(db_path, steps, out):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    agg = {}
    for n, c, t, a, p in rows:
        k = short(n)
        e = agg.setdefault(k, [0, 0.0, 0.0])
        e[0] += c; e[1] += t; e[2] += p
    lines = [f"# rocprofv3 --kernel-trace --stats summary ({db_path.split('/')[-1]}; {steps} steps incl. warm-up)", "",
             "| kernel | calls | total ms | avg us | % GPU time | ms / step |", "|---|---|---|---|---|---|"]
    for k, (c, t, p) in sorted(agg.items(), key=lambda kv: -kv[1][1]):           # every kernel of the run: the short ones are part of the path too
        lines.append(f"| `{k}` | {c} | {t / 1e3:.2f} | {t / c:.1f} | {p:.2f} | {t / 1e3 / steps:.3f} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:14]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3])
