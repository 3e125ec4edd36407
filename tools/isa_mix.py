#!/usr/bin/env python3
"""The static instruction mix of the step's integer / gather kernels and the measured class costs, as bench.py's issue roof reads them:

    python tools/isa_mix.py [--issue gpurun_out/r06_valu_issue.json]   ->  profiles/r06_isa_mix.json (+ profiles/r06_valu_issue.json)

r06_isa_mix.json: per kernel the share of every issue class among the VALU instructions of its loops (tools/isa_hist.py on `hipcc -S`, no
GPU needed), keyed by rnad_hip.source_hash().  r06_valu_issue.json: the microbenchmark's per-instruction cycles (tools/micro/valu_issue.hip on
the MI355X) plus `class_cycles`, the mean over a class's members -- what a class share is multiplied by."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "r-nad_amd"))
import isa_hist  # noqa: E402

KERNELS = [
    ("bucket.hip", "k_bucket_learn_c<3, unsigned char, false, false>"),
    ("bucket.hip", "k_bucket_learn_c<5, unsigned char, false, false>"),
    ("bucket.hip", "k_bucket_play_learn<3, unsigned char, false>"),
    ("bucket.hip", "k_bucket_rollout_items<3, unsigned char, 1>"),
    ("bucket.hip", "k_bucket_rollout_items<5, unsigned char, 1>"),
    ("bucket.hip", "k_bucket_keys_lds<3, 1, 4096>"),
    ("bucket.hip", "k_bucket_keys_hybrid<5, 1, 4096>"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--issue", help="the microbenchmark's JSON (gpurun_out/r06_valu_issue.json): also writes profiles/r06_valu_issue.json")
    a = ap.parse_args()
    import rnad_hip

    out = {"source_hash": rnad_hip.source_hash(), "how": "tools/isa_hist.py --loop (hipcc -S --offload-arch=gfx950, the Makefile's flags): "
           "share of each issue class among the VALU instructions inside the kernel's loops", "kernels": {}}
    asm_cache = {}
    for src, pattern in KERNELS:
        path = os.path.join(ROOT, "r-nad_amd", "csrc", src)
        if path not in asm_cache:
            asm = isa_hist.compile_asm(path)
            asm_cache[path] = isa_hist.kernels_of(asm)
            os.unlink(asm)
        kernels = asm_cache[path]
        name = isa_hist.pick(kernels, pattern)
        body, meta = kernels[name]
        hist, _ = isa_hist.histogram(body, True)
        valu = sum(hist[c] for c in isa_hist.VALU_CLASSES)
        out["kernels"][pattern] = {"valu": valu, "share": {c: hist[c] / valu for c in isa_hist.VALU_CLASSES if hist[c]}, "resources": meta,
                                   "other": {c: hist[c] for c in ("salu", "smem", "vmem", "lds", "wait", "branch") if hist[c]}}
    with open(os.path.join(ROOT, "profiles", "r06_isa_mix.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("profiles/r06_isa_mix.json", out["source_hash"], {k: v["valu"] for k, v in out["kernels"].items()})
    if a.issue:
        m = json.load(open(a.issue))
        by_class = {}
        for name, e in m["cycles"].items():
            if e["class"] in ("lds", "other", "pair"):
                continue
            by_class.setdefault(e["class"], []).append(e["cycles"])
        m["class_cycles"] = {c: sum(v) / len(v) for c, v in by_class.items()}
        m["class_cycles_note"] = ("mean over the class's measured opcodes; `pair` (v_cmp + v_cndmask through vcc) = the sum of its two classes; "
                                  "a class a kernel uses but the microbenchmark has no opcode of is priced as slow32")
        with open(os.path.join(ROOT, "profiles", "r06_valu_issue.json"), "w") as f:
            json.dump(m, f, indent=1)
        print("profiles/r06_valu_issue.json", {c: round(v, 2) for c, v in m["class_cycles"].items()})


if __name__ == "__main__":
    main()
