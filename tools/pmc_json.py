#!/usr/bin/env python3
"""One JSON with the counter evidence bench.py's roofline reads, from the per-kernel summaries of separate rocprofv3 passes
(tools/pmc_run.sh: FETCH_SIZE, WRITE_SIZE, SQ issue counters -- each in its own run with --kernel-trace only):

    tools/pmc_json.py <fetch.csv> <write.csv> <sq.csv> <out.json> [<k1_fetch.csv> <k1_write.csv> [<mfma.csv>]]

Per kernel: HBM traffic = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; FETCH_SIZE doubled as /opt/skills/guides/MI355X_MICROARCH.md
prescribes for gfx950), SQ_INSTS_VALU (wave-instructions), SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_WAIT_INST_ANY (quad-cycles: resident, parked on
s_waitcnt, issue-stalled), SQ_INSTS_LDS, the durations of the passes.  `source_hash` names the build (rnad_hip.source_hash()): bench.py uses the file only while it matches."""
import csv
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "r-nad_amd"))


def load(path):
    out = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            out[re.sub(r"<.*", "", row["kernel"])] = row
    return out


def main():
    import rnad_hip

    fetch, write, sq = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
    if len(sys.argv) > 6:  # K1 (k_observe) is not in the step: its passes run over tools/k1_pmc.py
        for dst, path in ((fetch, sys.argv[5]), (write, sys.argv[6])):
            dst.update({k: v for k, v in load(path).items() if k.startswith("k_observe")})
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
    except Exception:
        head = None  # the GPU box has no .git: the source hash is what identifies the build
    out = {"source_hash": rnad_hip.source_hash(), "git_head_when_written": head,
           "how": "rocprofv3 --kernel-trace --pmc <counters>, one pass per counter group (tools/pmc_run.sh) over tools/step_probe.py --no-graph; "
                  "traffic = 2 x FETCH_SIZE + WRITE_SIZE", "kernels": {}}
    for k in sorted(set(fetch) | set(write) | set(sq)):
        e = {}
        if k in fetch and k in write:
            f_kib, w_kib = float(fetch[k]["FETCH_SIZE_mean"]), float(write[k]["WRITE_SIZE_mean"])
            e.update(FETCH_SIZE_raw_KiB=f_kib, WRITE_SIZE_KiB=w_kib, traffic_bytes_per_launch=int(2 * f_kib * 1024 + w_kib * 1024),
                     duration_us_fetch_pass=float(fetch[k]["mean_duration_us"]), duration_us_write_pass=float(write[k]["mean_duration_us"]))
        if k in sq:
            r = sq[k]
            e.update(duration_us_sq_pass=float(r["mean_duration_us"]), launches=int(r["launches"]))
            for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if r.get(c + "_mean"):
                    e[c] = float(r[c + "_mean"])
        out["kernels"][k] = e
    if len(sys.argv) > 8:  # vector-L1 (TCP) counters of the bucket kernels (their own pass)
        for k, r in load(sys.argv[8]).items():
            e = out["kernels"].setdefault(k, {})
            for c in ("TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_GATE_EN1_sum", "TCP_TCC_READ_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum"):
                if r.get(c + "_mean"):
                    e[c] = float(r[c + "_mean"])
            e["duration_us_tcp_pass"] = float(r["mean_duration_us"])
    if len(sys.argv) > 7:  # matrix-pipe busy cycles of the MLP kernels (their own pass)
        for k, r in load(sys.argv[7]).items():
            e = out["kernels"].setdefault(k, {})
            for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU"):
                if r.get(c + "_mean"):
                    e[c + "_mfma_pass"] = float(r[c + "_mean"])
            e["duration_us_mfma_pass"] = float(r["mean_duration_us"])
    with open(sys.argv[4], "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: {"traffic": v.get("traffic_bytes_per_launch"), "valu": v.get("SQ_INSTS_VALU")} for k, v in out["kernels"].items()}))


if __name__ == "__main__":
    main()
