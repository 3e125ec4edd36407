#!/usr/bin/env python3
"""profiles/<tag>_k1_pmc.json from the two K1 counter passes of tools/round_artifacts.sh (tools/k1_pmc.py under rocprofv3 --pmc
FETCH_SIZE and --pmc WRITE_SIZE): per-launch HBM bytes of k_observe at B = 2^20 (launches with a 2^20-thread grid only).

usage: tools/k1_pmc_json.py <gpurun_out dir> <tag> <out.json>
"""
import csv
import glob
import json
import os
import sys


def collect(d, counter):
    vals, durs = [], []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            if "k_observe" in r["Kernel_Name"] and r["Counter_Name"] == counter and int(r["Grid_Size"]) == 1 << 20:
                vals.append(float(r["Counter_Value"]))
                durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return vals, durs


def main():
    root, tag, out = sys.argv[1:4]
    f, fd = collect(os.path.join(root, f"pmc_{tag}_k1_fetch"), "FETCH_SIZE")
    w, wd = collect(os.path.join(root, f"pmc_{tag}_k1_write"), "WRITE_SIZE")
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    res = {
        "kernel": "k_observe<3,*,float,true>, B=2^20, c2 tree, one launch per env step of a real rollout (T=12), 3 repetitions",
        "command": "tools/pmc_run.sh <tag>_k1_fetch FETCH_SIZE ... -- python tools/k1_pmc.py ; tools/pmc_run.sh <tag>_k1_write WRITE_SIZE ... -- python tools/k1_pmc.py",
        "launches": [len(f), len(w)],
        "FETCH_SIZE_raw_KiB_per_launch": fk, "FETCH_SIZE_corrected_KiB_per_launch": 2 * fk, "WRITE_SIZE_KiB_per_launch": wk,
        "traffic_bytes_per_launch": int((2 * fk + wk) * 1024),
        "mean_duration_us_in_the_counter_passes": [sum(fd) / len(fd), sum(wd) / len(wd)],
        "algorithmic_bytes_per_launch_survey_8d": 160 << 20,
        "note": "FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md (it tallies 128-B requests at 64 B); writes = 72 B obs + 1 B "
                "mask bits per lane; fetches = the 4 B index stream plus L2 misses on the 48 B node rows; ev/legal rows are served from L2 and "
                "legal travels as bits, so the traffic is below SURVEY 8d's 160 B/step model",
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res)[:400])


if __name__ == "__main__":
    main()
