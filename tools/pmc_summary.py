#!/usr/bin/env python3
"""Per-kernel means of a rocprofv3 `--pmc ... --kernel-trace` pass.

usage: tools/pmc_summary.py <dir with *_counter_collection.csv> [kernel-name regex]
Prints CSV: kernel, launches, mean duration (us), then mean / min / max of every counter collected (raw counter units:
FETCH_SIZE and WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reads 1/2 of the bytes of a wide streaming read, see
/opt/skills/guides/MI355X_MICROARCH.md "HBM").
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([A-Za-z_0-9:]+(?:<[^(]{0,60})?)", name)
    return (m.group(1) if m else name)[:90]


def main():
    d = sys.argv[1]
    rx = re.compile(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] else None
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("no counter_collection.csv under", d)
        return 1
    per = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> values (one per dispatch)
    dur = defaultdict(dict)                        # kernel -> dispatch id -> us
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                if rx and not rx.search(k):
                    continue
                per[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                dur[k][row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
    counters = sorted({c for k in per for c in per[k]})
    w = csv.writer(sys.stdout)
    head = ["kernel", "launches", "mean_duration_us"]
    for c in counters:
        head += [c + "_mean", c + "_min", c + "_max"]
    w.writerow(head)
    for k in sorted(per, key=lambda k: -sum(dur[k].values())):
        n = len(dur[k])
        row = [k, n, f"{sum(dur[k].values()) / max(n, 1):.2f}"]
        for c in counters:
            v = per[k].get(c, [])
            row += [f"{sum(v) / len(v):.1f}", f"{min(v):.1f}", f"{max(v):.1f}"] if v else ["", "", ""]
        w.writerow(row)
    return 0


if __name__ == "__main__":
    sys.exit(main())
