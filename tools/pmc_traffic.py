#!/usr/bin/env python3
"""HBM bytes per launch of the step's kernels from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; each in its own run with
--kernel-trace only, tools/pmc_run.sh), corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE reads
1/2 of the bytes of a wide coalesced read, so it is doubled; both counters are in KiB.

usage: tools/pmc_traffic.py <fetch.csv> <write.csv> <out.json> [source text]
"""
import csv
import json
import re
import sys


def load(path, counter):
    out = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = re.sub(r"<.*", "", row["kernel"])
            out[name] = (float(row[counter + "_mean"]), float(row["mean_duration_us"]), int(row["launches"]))
    return out


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    source = sys.argv[4] if len(sys.argv) > 4 else ""
    out = {}
    for k in sorted(set(fetch) & set(write)):
        f_kib, f_us, n = fetch[k]
        w_kib, w_us, _ = write[k]
        out[k] = {
            "FETCH_SIZE_raw_KiB_per_launch": f_kib, "WRITE_SIZE_KiB_per_launch": w_kib,
            "traffic_bytes_per_launch": int(2 * f_kib * 1024 + w_kib * 1024),
            "mean_duration_us_in_the_counter_passes": [f_us, w_us], "launches": n,
            "source": source or "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); traffic = 2 x FETCH_SIZE + WRITE_SIZE",
        }
    with open(sys.argv[3], "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v["traffic_bytes_per_launch"] for k, v in out.items()}))


if __name__ == "__main__":
    main()
