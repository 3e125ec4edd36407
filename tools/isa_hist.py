#!/usr/bin/env python3
"""Histogram of a gfx950 kernel's instructions by issue class, from `hipcc -S` of its source (no GPU needed).

    python tools/isa_hist.py r-nad_amd/csrc/bucket.hip 'k_bucket_learn<3, true, false>' [--loop] [--json out.json] [--dump out.s]

The classes are the ones tools/micro/valu_issue.hip measures the issue cost of (cycles a SIMD is busy per wave64 instruction);
`bench.py` multiplies this static mix by the measured costs and by SQ_INSTS_VALU (dynamic wave-instructions per launch) to get the
"issue" roof of a kernel:  issue_cycles = SQ_INSTS_VALU * sum_c share_c * cycles_c.

--loop restricts the histogram to the basic blocks that sit inside a backward branch (the loops: where the dynamic instructions
are), weighting every block once.  Flags mirror r-nad_amd/csrc/Makefile.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden",
         "-I" + os.path.join(ROOT, "include"), "-Wno-unused-function", "-S", "--cuda-device-only"]

# issue classes, first match wins (the class labels of tools/micro/valu_issue.hip, which measures their cost)
FAST32 = ("v_fma_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshrrev_b32",
          "v_add_u32", "v_sub_u32", "v_subrev_u32")
CLASSES = [
    ("trans32", re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_(f32|f16|legacy_f32|iflag_f32)")),
    ("cvt64", re.compile(r"^v_cvt_f64_|^v_cvt_\w+_f64")),
    ("f64", re.compile(r"^v_\w+_f64")),
    ("mad_u64", re.compile(r"^v_mad_[ui]64_[ui]32")),
    ("int64", re.compile(r"^v_(lshl_add_u64|lshlrev_b64|lshrrev_b64|ashrrev_i64|add_u64|sub_u64|add_i64)")),
    ("mov64", re.compile(r"^v_mov_b64")),
    ("pk32", re.compile(r"^v_pk_")),
    ("mul32", re.compile(r"^v_mul_(lo|hi)_[ui]32")),
    ("mfma", re.compile(r"^v_mfma|^v_smfmac")),
    ("dpp_move", re.compile(r"^v_(readlane|readfirstlane|writelane|permlane|swap)")),
    ("mov32", re.compile(r"^v_(mov_b32|accvgpr)")),
    ("fmac", re.compile(r"^v_fmac_f32")),
    ("cndmask", re.compile(r"^v_cndmask_b32")),
    ("fast32", re.compile(r"^(" + "|".join(FAST32) + r")(_e32|_e64|_sdwa|_dpp)?$")),
    ("slow32", re.compile(r"^v_")),
    ("lds", re.compile(r"^ds_")),
    ("vmem", re.compile(r"^(global|buffer|flat|scratch)_")),
    ("smem", re.compile(r"^s_(load|buffer_load|store|atomic)")),
    ("wait", re.compile(r"^s_(waitcnt|nop|sleep|barrier)")),
    ("branch", re.compile(r"^s_(branch|cbranch|setpc|swappc|endpgm)")),
    ("salu", re.compile(r"^s_")),
]
VALU_CLASSES = ("trans32", "cvt64", "f64", "mad_u64", "int64", "mov64", "pk32", "mul32", "dpp_move", "mov32", "fmac", "cndmask", "fast32", "slow32")


def classify(op):
    for name, rx in CLASSES:
        if rx.match(op):
            return name
    return "other"


def compile_asm(source, defines=()):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    extra = []
    base = os.path.basename(source)
    if base == "mlp_fwd.hip" or base == "mlp_bwd_t.hip":
        extra = ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"]
    elif base == "mlp_bwd.hip":
        extra = ["-fno-honor-nans"]
    cmd = [HIPCC] + FLAGS + extra + ["-D" + d for d in defines] + ["-I" + os.path.dirname(os.path.abspath(source)), source, "-o", out]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return out


def kernels_of(asm_path):
    """{demangled name: [lines of the body]} for every kernel symbol of the file, and its resource metadata."""
    text = open(asm_path).read().split("\n")
    symbols = [m.group(1) for m in (re.match(r"^(_Z\w+):", ln) for ln in text) if m]
    demangled = subprocess.run(["c++filt"], input="\n".join(symbols), capture_output=True, text=True).stdout.split("\n")
    names = dict(zip(symbols, demangled))
    bodies, cur = {}, None
    for ln in text:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1)
            bodies[cur] = []
            continue
        if cur is not None:
            if ln.startswith(".Lfunc_end"):  # (not the first s_endpgm: a kernel with an early return has several)
                cur = None
            else:
                bodies[cur].append(ln)
    meta = {}
    blob = "\n".join(text)
    # resource usage: the `.set <symbol>.num_vgpr, N` lines that follow each kernel
    for m in re.finditer(r"\.set (_Z\w+)\.(num_vgpr|num_agpr|numbered_sgpr|private_seg_size), (\d+)", blob):
        meta.setdefault(m.group(1), {})[m.group(2)] = int(m.group(3))
    return {names[s]: (bodies[s], meta.get(s, {})) for s in bodies}


def blocks_of(body):
    """[(label, [opcodes])] in layout order."""
    blocks, cur = [], ("entry", [])
    for ln in body:
        s = ln.strip()
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            blocks.append(cur)
            cur = (m.group(1), [])
            continue
        if not s or s.startswith(";") or s.startswith(".") or s.startswith("//"):
            continue
        cur[1].append(s)
    blocks.append(cur)
    return blocks


def loop_blocks(blocks):
    """indices of blocks that lie between the target of a backward branch and the branch (natural loops by layout)."""
    pos = {label: i for i, (label, _) in enumerate(blocks)}
    inside = set()
    for i, (_, ops) in enumerate(blocks):
        for s in ops:
            m = re.match(r"^s_c?branch\w*\s+(\.LBB\w+)", s)
            if m and m.group(1) in pos and pos[m.group(1)] <= i:
                inside.update(range(pos[m.group(1)], i + 1))
    return inside


def histogram(body, loops_only=False):
    blocks = blocks_of(body)
    keep = loop_blocks(blocks) if loops_only else set(range(len(blocks)))
    hist, ops_seen = collections.Counter(), collections.Counter()
    for i, (_, ops) in enumerate(blocks):
        if i not in keep:
            continue
        for s in ops:
            op = s.split()[0]
            hist[classify(op)] += 1
            ops_seen[op] += 1
    return hist, ops_seen


def pick(kernels, pattern):
    hits = [n for n in kernels if pattern in n]
    if not hits:
        raise SystemExit(f"no kernel matches {pattern!r}; have e.g. {sorted(kernels)[:5]}")
    exact = [n for n in hits if n.split("(")[0].endswith(pattern)]
    return (exact or hits)[0]


def kernel_mix(source, pattern, loops_only=True, defines=()):
    """{'classes': {class: count}, 'valu': n, 'share': {class: fraction of VALU}, 'vgprs': ...} of one kernel."""
    asm = compile_asm(source, defines)
    try:
        kernels = kernels_of(asm)
    finally:
        os.unlink(asm)
    name = pick(kernels, pattern)
    body, meta = kernels[name]
    hist, ops = histogram(body, loops_only)
    valu = sum(hist[c] for c in VALU_CLASSES)
    return {"kernel": name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], "loops_only": loops_only, "classes": dict(hist), "valu": valu,
            "share": {c: hist[c] / valu for c in VALU_CLASSES if hist[c]} if valu else {}, "resources": meta,
            "top_ops": ops.most_common(25)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("kernel", help="substring of the demangled name, e.g. 'k_bucket_learn<3, true, false>'")
    ap.add_argument("--loop", action="store_true", help="only the basic blocks inside loops")
    ap.add_argument("--json")
    ap.add_argument("--dump", help="write the kernel's assembly here")
    ap.add_argument("-D", dest="defines", action="append", default=[])
    a = ap.parse_args()
    if a.dump:
        asm = compile_asm(a.source, a.defines)
        ks = kernels_of(asm)
        os.unlink(asm)
        open(a.dump, "w").write("\n".join(ks[pick(ks, a.kernel)][0]))
    mix = kernel_mix(a.source, a.kernel, a.loop, a.defines)
    if a.json:
        json.dump(mix, open(a.json, "w"), indent=1)
    print(mix["kernel"], "(loops only)" if a.loop else "(whole kernel)", mix["resources"])
    for c, n in sorted(mix["classes"].items(), key=lambda kv: -kv[1]):
        print(f"  {c:10s} {n:6d}" + (f"  {n / mix['valu']:.3f} of VALU" if c in VALU_CLASSES else ""))
    print("  VALU total", mix["valu"])
    print("  top opcodes:", ", ".join(f"{o} {n}" for o, n in mix["top_ops"]))


if __name__ == "__main__":
    sys.exit(main())
