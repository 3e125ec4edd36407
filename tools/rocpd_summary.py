#!/usr/bin/env python3
"""Turn a rocprofv3 `--kernel-trace --stats` result (rocpd sqlite .db, or *_kernel_stats.csv) into a short CSV summary.

usage: tools/rocpd_summary.py <results.db> [out.csv]
Kernel names are shortened to their function name; rows sorted by total time.
"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?((?:at::native::)?[A-Za-z_0-9:]+(?:<[^(]{0,80})?)", name)
    s = name if name.startswith("Cijk") else (m.group(1) if m else name)
    if name.startswith("Cijk"):
        mt = re.search(r"MT\d+x\d+x\d+", name)
        s = name.split("_S_")[0] + "_" + (mt.group(0) if mt else "")
    if "vectorized_elementwise_kernel" in name or "elementwise_kernel" in name or "reduce_kernel" in name:
        f = re.search(r"(launch_clamp_scalar|threshold_kernel_impl|sum_functor<\w+|MulFunctor|CUDAFunctor_add|FillFunctor<\w+>|direct_copy_kernel_cuda|CompareEqFunctor|uniform_kernel|random_from_to)", name)
        s = s.split("<")[0] + ("[" + f.group(1) + "]" if f else "")
    return s[:110]


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
    con = sqlite3.connect(db)
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    w = csv.writer(out)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
    for name, calls, total, avg, pct in rows:
        w.writerow([short(name), calls, f"{total:.1f}", f"{avg:.2f}", f"{pct:.3f}"])


if __name__ == "__main__":
    main()
