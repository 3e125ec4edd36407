#!/usr/bin/env python3
"""bench.py -- the R-NaD self-play hot path on MI355X: env-steps/s and updates/s at batch 2^20.

    python bench.py --gpus 1 --steps 2000 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

One "step" = one iteration of the reference's training loop (learn/rnad.py:495-526): roll out a batch of episodes with the
learner net (Episodes.generate), sample the buffer, RNaD.__learn (the four nets, V-trace / NeuRD, backward), Adam, EMA target.
Workload = BASELINE.json configs[1]: depth-6 ternary (3x3) tree, C = 1, 66 431 states, GLOBAL batch 2^20 episodes x 12 env steps,
MLP width 256, fp32.  `value` times RNaD's default net-evaluation mode (tabular = True: the nets are evaluated once per (player,
state) observation -- 132 862 rows for 12.6 M slots --, the rollout is bucket-ordered, the per-slot gradients are summed per row in
LDS and one backward over the rows gives the weight gradients; the step is replayed from a captured hipGraph); `other_modes` times
"forward" (backward per slot: bit-identical to dense) and dense (every net on every slot, as the reference does) in the same process.
N > 1 (default): BASELINE.json configs[2] -- ONE batch of 2^22 episodes sharded over the ranks (2^19 per GPU at N = 8; "scaling": "strong")
with one RCCL all-reduce of the 2 loss normalisers (beside the learner kernel) and one of the 43 KB gradient bucket per step; rank 0
also times the same 2^22 batch on its GPU alone (`strong_scaling.base`), the N = 1 point of that curve.  --scaling weak gives every
rank its own 2^batch-log2 episodes instead.  The N > 1 run times the eagerly enqueued step first and then the step replayed from a
hipGraph with the RCCL collectives captured inside it, under a watchdog: if the captured variant does not finish, the eager
measurement is what gets printed.

Prints ONE JSON line on rank 0.  `value` = env steps of all ranks / wall time of the K timed steps (inputs resident in HBM; the
tree is generated and uploaded before the timed region).  `roofline` is for the kernel that takes the largest share of the step,
its launches bracketed with hipEvents on the launching stream in an eager (un-captured) leg of the same steps right after the
timed region (events cannot sit inside a replayed graph); `kernels` lists every bracketed kernel of the step the same way.
`cpu_baseline` times the CPU port (oracle/port.py: C oracle + PyTorch-CPU MLP) of the same step on a bounded sample, rank 0, N = 1.
"""
import argparse
import json
import os
import sys
import tempfile
import time

# host threads of the CPU-baseline leg: fixed BEFORE torch / libgomp start (256 spinning threads on the GPU box's host
# are 8x slower than 16).  The GPU path does not use them.
CPU_THREADS = int(os.environ.get("RNAD_CPU_THREADS", min(os.cpu_count() or 1, 16)))
os.environ.setdefault("OMP_NUM_THREADS", str(CPU_THREADS))

ROOT = os.path.dirname(os.path.realpath(__file__))
sys.path.insert(0, os.path.join(ROOT, "r-nad_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
CLOCK_HZ = 2.4e9       # max shader clock (same guide)
SIMDS = 256 * 4        # 256 CUs x 4 SIMDs
# The rooflines of the integer / gather kernels of this path (keys, rollout, learner), all three always in the line:
#   hbm    algorithmic bytes per launch / launch duration / 8 TB/s
#   issue  VALU issue-port time / (SIMDs x launch duration x 2.4 GHz).  Issue-port time = SQ_INSTS_VALU (dynamic wave-instructions per
#          launch, rocprofv3 counter pass: profiles/r0N_pmc.json) x the kernel's mean cycles per instruction = sum over instruction
#          classes of [share of the class in the kernel's loops (tools/isa_hist.py on `hipcc -S`: profiles/r0N_isa_mix.json)] x [cycles a
#          SIMD's issue port is busy per wave64 instruction of the class (tools/micro/valu_issue.hip on this hardware:
#          profiles/r0N_valu_issue.json: 1.8 - 1.9 for plain fp32 / integer / move, 2.8 - 3.0 for min / max / med3 / compare / select /
#          64-bit / fp64 / packed, 3.1 - 3.4 for v_cvt_f64_f32 / v_mad_u64_u32, 5.3 transcendental)]
#   wait   SQ_WAIT_ANY / SQ_WAVE_CYCLES (share of a resident wave's time parked on s_waitcnt) and SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
#          (issue stalls) from the same counter pass
# (r03 priced the issue roof at 4 cycles per instruction from SQ_ACTIVE_INST_VALU, which counts instructions, not cycles.)
def _profile_file(name):
    """profiles/r06_<name> (tools/round_artifacts.sh r06), else the r05 file of the same name."""
    for tag in ("r06", "r05"):
        path = os.path.join(ROOT, "profiles", f"{tag}_{name}")
        if os.path.exists(path):
            return path
    return os.path.join(ROOT, "profiles", f"r06_{name}")


PMC_FILE = _profile_file("pmc.json")
PMC_C4_FILE = _profile_file("pmc_c4.json")  # the same counter passes over the configs[3] step (tools/pmc_config.sh)
ISA_MIX_FILE = _profile_file("isa_mix.json")
ISSUE_FILE = _profile_file("valu_issue.json")


def _load(path):
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def pmc_counters():
    """(kernels, matches): counter evidence of the step's kernels from the separate rocprofv3 --pmc passes (tools/round_artifacts.sh ->
    tools/pmc_json.py) and whether the file was measured on THIS build (its source hash is this tree's).  Counters of another build
    are still reported -- instruction counts move little between builds -- but flagged."""
    import rnad_hip

    pmc = _load(PMC_FILE)
    return pmc.get("kernels", {}), pmc.get("source_hash") == rnad_hip.source_hash()


def issue_cycles_per_instruction(kernel):
    """Mean issue-port cycles per VALU wave-instruction of `kernel`: its static class mix x the measured class costs; None if either
    file is missing.  Returns (cycles, detail)."""
    mix = _load(ISA_MIX_FILE).get("kernels", {}).get(kernel)
    cost = _load(ISSUE_FILE).get("class_cycles")
    if not mix or not cost:
        return None, None
    default = cost.get("slow32", 2.9)  # (a class the microbenchmark has no opcode of is priced as a slow 32-bit instruction)
    cyc = sum(share * cost.get(cls, default) for cls, share in mix["share"].items())
    return cyc, {"class_share": mix["share"], "class_cycles": {c: cost.get(c) for c in mix["share"]}, "static_valu_in_loops": mix["valu"],
                 "vgprs": mix.get("resources", {}).get("num_vgpr")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch-log2", type=int, default=None,
                    help="log2 of the episode batch: of the whole job under --scaling strong, per GPU under --scaling weak.  Default: 20 on "
                         "one GPU (BASELINE.json configs[1]), 22 on several (configs[2]: one 2^22 batch sharded over the GPUs)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="N > 1: strong (default) = one 2^batch-log2 batch sharded over the GPUs (configs[2]); weak = every GPU gets 2^batch-log2 episodes")
    ap.add_argument("--no-base-leg", action="store_true",
                    help="N > 1, strong scaling: do not time the same global batch on rank 0's GPU alone (the N = 1 point of the curve)")
    ap.add_argument("--graph-timeout", type=float, default=120.0,
                    help="N > 1: seconds the captured-graph leg may take before the eager measurement is printed instead")
    ap.add_argument("--depth", type=int, default=6)
    ap.add_argument("--actions", type=int, default=3)
    ap.add_argument("--transitions", type=int, default=1)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--prune", type=int, nargs=2, default=(0, 0), metavar=("NUM", "DEN"),
                    help="each child's depth drops by 2 more with probability NUM/DEN (reference main.py:37); configs[3] uses a pruned tree")
    ap.add_argument("--threshold", type=float, default=None, help="transition_threshold (default 0 for C=1, 0.5/C otherwise)")
    ap.add_argument("--tree-seed", type=int, default=0)
    ap.add_argument("--net-mode", choices=("default", "dense", "forward", "tabular"), default="default",
                    help="RNaD.tabular for the timed `value`: dense = False, forward = 'forward', tabular = True (RNaD's default); "
                         "the other modes are reported under other_modes either way")
    ap.add_argument("--no-graph", action="store_true", help="RNaD.use_graph = False: enqueue every step eagerly")
    ap.add_argument("--shard-rows", action="store_true",
                    help="N > 1: RNaD.shard_rows -- the table forwards / records / backward on a rank's 1/N of the 2S rows (all-gather of the "
                         "record tables, all-reduce of the 64-bit per-row sums); default: every rank evaluates all rows")
    ap.add_argument("--obs-half", action="store_true", help="fp16 observations (BASELINE configs[4])")
    ap.add_argument("--other-steps", type=int, default=20, help="timed steps of each entry of other_modes and of the eager kernel-timing leg")
    ap.add_argument("--cpu-lanes-log2", type=int, default=None,
                    help="episodes per step of the CPU-baseline sample (C port); default: the benchmarked batch itself up to 2^20 (r06: one full "
                         "2^20-lane step of the port takes ~25 s on 16 host threads -- r05: 65 s, hence its 2^18-lane sample)")
    ap.add_argument("--cpu-torch-lanes-log2", type=int, default=16, help="episodes per step of the CPU-baseline sample (plain PyTorch-CPU leg)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-legs", action="store_true",
                    help="skip the legs the default one-GPU line carries beside `value`: configs[3], dedup off, a sharp policy, 2^19 / 2^22 lanes")
    ap.add_argument("--side-steps", type=int, default=200, help="timed replays of each side leg")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.scaling is None:
        args.scaling = "strong" if world > 1 else "weak"  # (one GPU: the two coincide)
    if args.batch_log2 is None:
        args.batch_log2 = 22 if (world > 1 and args.scaling == "strong") else 20
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # RNAD_BENCH_REHEARSAL=1: the N > 1 control flow on a one-GPU box -- every rank on cuda:0, gloo transport (RCCL refuses two ranks
    # on one device; gloo collectives cannot be captured, so the graph leg reports "could not be captured").  Not a measurement.
    rehearsal = os.environ.get("RNAD_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    import rnad_hip
    from environment.episode import Buffer, Episodes
    from environment.tree import Tree
    from learn.rnad import RNaD

    A, C, depth = args.actions, args.transitions, args.depth
    if args.scaling == "weak":
        local_batch = 1 << args.batch_log2
        global_batch = local_batch * world
    else:
        global_batch = 1 << args.batch_log2
        assert global_batch % world == 0
        local_batch = global_batch // world

    # ---- setup (untimed): tree tables into HBM, nets, optimizer
    t0 = time.perf_counter()
    threshold = args.threshold if args.threshold is not None else (0.0 if C == 1 else 0.5 / C)
    tree = Tree(device=device, max_actions=A, max_transitions=C, depth_bound=depth, transition_threshold=threshold)
    tree.generate_native(seed=args.tree_seed, prune=tuple(args.prune))
    handle = tree.handle()
    setup_tree_s = time.perf_counter() - t0
    os.environ["RNAD_SAVE_DIR"] = tempfile.mkdtemp(prefix="rnad_bench_")
    delta_m = 10_000

    def make_trainer(batch, name, data_parallel=True, tree=tree):
        torch.manual_seed(0)
        t = RNaD(tree=tree, device=device, directory_name=name, batch_size=batch, eta=0.2, b1_adam=0.0,
                 net_params={"type": "MLP", "max_actions": tree.max_actions, "width": args.width})
        t.data_parallel = data_parallel
        t.initialize()
        t.obs_half = args.obs_half
        t.use_graph = not args.no_graph
        t.shard_rows = bool(args.shard_rows and data_parallel)
        if args.net_mode != "default":
            t.tabular = {"dense": False, "forward": "forward", "tabular": True}[args.net_mode]
        with torch.no_grad():
            # the general case of rnad.py:382: two DISTINCT regularisation nets and 0 < alpha < 1 (all four nets matter)
            for p in t.net_reg_.parameters():
                p.mul_(1.001)
        buf = Buffer(t.n_batches_per_buffer)
        count = {"i": 0}

        def step():
            i = count["i"]
            alpha = t.alpha_of(i, delta_m)  # rnad.py:497
            t.alpha_ahead = lambda k: t.alpha_of(i + k, delta_m)  # (as RNaD.run does: the scalars of the coming steps are queued on the device)
            t.train_step(buf, alpha)
            t.total_steps += 1
            count["i"] = i + 1

        return t, buf, step

    rn, buffer, one_step = make_trainer(global_batch, f"bench-r{rank}")

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n):
        fence()
        t_s = time.perf_counter()
        host = 0.0
        for _ in range(n):
            h0 = time.perf_counter()
            one_step()
            host += time.perf_counter() - h0
        fence()
        return time.perf_counter() - t_s, host

    # setup (untimed, before the caller's warmup): priming steps, so that every code object, the caching allocator's pools, RCCL's
    # channels and -- in the default mode -- the captured graph of the step exist whatever --warmup is, and the GPU has left the idle
    # clocks the host-side setup (tree generation, uploads) let it fall to: the first replays after an idle second run ~6 % slow, which
    # a short timed region (--steps 20 is 5 ms) would otherwise be made of.  Reported as `priming_steps`.
    PRIME = 300

    def timed_leg(use_graph):
        rn.use_graph = use_graph
        for _ in range(PRIME):
            one_step()
        fence()
        for _ in range(args.warmup):
            one_step()
        # ---- timed region: EXACTLY --steps steps between two fences
        sec = timed(args.steps)[0]
        if world > 1:
            t = torch.tensor([sec], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sec = float(t.item())
        return sec

    legs = {}
    if world == 1:
        elapsed = timed_leg(not args.no_graph)
    else:
        # the eagerly enqueued step first: RCCL's ordinary path.  The captured step (collectives inside the graph) is measured after
        # everything the JSON line needs exists, under a watchdog (see below).
        elapsed = legs["eager"] = timed_leg(False)
    # host cost of enqueueing one step, measured with the queue kept short (a long un-synchronised run measures back-pressure)
    host_s, host_n = 0.0, min(64, args.steps)
    for i in range(host_n):
        h0 = time.perf_counter()
        one_step()
        host_s += time.perf_counter() - h0
        if i % 4 == 3:
            torch.cuda.synchronize()
    fence()
    graph_state = getattr(rn, "_graph", None)
    replayed = bool(graph_state and graph_state.get("graph") is not None)
    T = rn.last_episodes.t_eff + 1
    args.compact_in_effect = getattr(rn.last_episodes, "_compact", None) is not None
    args.rel_bytes = args.stored_slots = 0
    if args.compact_in_effect:
        args.rel_bytes = rn.last_episodes.buckets.plan.rel_bytes
        args.stored_slots = rnad_hip.stored_state_slots(handle, rn.last_episodes.buckets, T)
    lazy_now = rn._use_lazy_rows(handle, local_batch, T, None, buffer) and mode_now_is_true(rn, T, local_batch)
    args.visited_rows = int(rn.last_rows.count.item()) if (lazy_now and rn.last_rows is not None) else 0
    staged = getattr(rn.last_episodes, "staged_rows", None) if lazy_now else None
    args.policy_rows = int(sum(r.count.item() for r in staged)) if staged else 0  # rows the staged policy head was evaluated on
    args.fold = bool(rn._fold() and mode_now_is_true(rn, T, local_batch))
    # distinct observations (RNaD.dedup_rows): the table launch and the backward run on one representative row per observation
    dd = rn._dedup_now(handle, None, lazy_now, None, rn._fold()) if mode_now_is_true(rn, T, local_batch) else None
    args.unique_rows = int(dd.n_unique) if dd is not None else 0
    default_mode = rn.tabular
    mode_now = rn._tabular_mode(T, local_batch)

    # ---- per-kernel durations: the same steps again, eagerly, every kernel bracketed with hipEvents on the launching stream
    E = max(1, min(args.other_steps, args.steps))
    rn.use_graph = False
    for _ in range(2):
        one_step()
    fence()
    rnad_hip.prof_enable(True)
    EP = max(E, min(200, args.steps))  # steps of the bracketed leg (eager steps are host-paced: short legs see the clocks ramp)
    for _ in range(EP):
        one_step()
    fence()
    # (the default step's table launch also writes the row records: csrc/mlp_rows.hip; the lazy-rows step runs it for the value heads)
    fused_rows = mode_now_is_true(rn, T, local_batch) and rnad_hip.mlp_rows_records_supported(A, args.width, rn._fold(), lazy_now)
    fwd_name = "k_rows_forward_records (table forwards + row records)" + (" + k_mlp_forward (staged actor)" if lazy_now else "") if fused_rows else "k_mlp_forward"
    names = {rnad_hip.PROF_OBSERVE: "k_observe", rnad_hip.PROF_ACT: "rollout (all kernels of Episodes.generate)",
             rnad_hip.PROF_LEARN: "learner (all kernels between the forwards and the backward)", rnad_hip.PROF_MLP: fwd_name,
             rnad_hip.PROF_MLP_BWD: "k_mlp_backward", rnad_hip.PROF_BUCKET_KEYS: "k_bucket_keys (with the sort tile's histogram)",
             rnad_hip.PROF_BUCKET_SORT: "k_bucket_scan+scatter (+hist on the global-table fallback)", rnad_hip.PROF_BUCKET_ROLLOUT: "k_bucket_rollout",
             rnad_hip.PROF_BUCKET_LEARN: "k_bucket_learn", rnad_hip.PROF_BUCKET_FINISH: "k_bucket_finish"}
    prof = {}
    for k, nm in names.items():
        n, ms = rnad_hip.prof_read(k)
        if n:
            prof[k] = dict(name=nm, launches_per_step=n / EP, avg_launch_us=ms * 1e3 / n, us_per_step=ms * 1e3 / EP)
    rnad_hip.prof_enable(False)
    rn.use_graph = not args.no_graph
    # rollout and learner in one launch (RNaD.fuse_rollout_learner, k_bucket_play_learn): booked under the learner's id, nothing under the rollout's
    args.fused_play_learn = (mode_now is True and rnad_hip.PROF_BUCKET_LEARN in prof and rnad_hip.PROF_BUCKET_ROLLOUT not in prof)
    # (the learner on the tree's leaf paths, DESIGN.md section 5.6: rollout and learner are two launches inside the same scope)
    with rnad_hip.workspace_owner(rn._workspace_token()):  # (the trainer's own plan: no second LeafPaths on the default one)
        args.leaf_paths = bool(args.fused_play_learn and rn._fuse_now() and rn._leaf_now(handle, local_batch, T) is not None)
    if args.leaf_paths:
        args.fused_play_learn = False
        prof[rnad_hip.PROF_BUCKET_LEARN]["name"] = "k_bucket_play_count + k_bucket_learn_c<WEIGHTED> (rollout, then the learner on the tree's leaf paths)"
    elif args.fused_play_learn:
        prof[rnad_hip.PROF_BUCKET_LEARN]["name"] = "k_bucket_play_learn (rollout + learner of a work item in one launch)"

    # ---- the same step in the other net-evaluation modes of RNaD (reported separately, NOT `value`), eager
    variants = {}
    for name, mode in (("dense_nets", False), ("forward", "forward"), ("tabular_nets", True)):
        if world > 1:
            break  # the N > 1 lines are for the scaling curve: the other modes are reported at N = 1
        if mode == default_mode or (mode and not rn._fused_mlp()):
            continue
        rn.tabular = mode
        if rn._tabular_mode(T, local_batch) != mode:
            continue
        one_step()
        one_step()
        variants[name] = timed(E)[0]
    rn.tabular = default_mode
    # rollout alone (Episodes.generate as RNaD.train_step calls it), outside the headline timed region
    actor_tables = rn._table_outputs(0.5, fold=args.fold) if mode_now is True else None
    fence()
    t_r = time.perf_counter()
    for i in range(E):
        ep = Episodes(tree, local_batch, seed=1000 + i, lane_offset=rank * local_batch, obs_half=args.obs_half)
        ep.generate(rn.net, trim=False, skip_absorbed=True, store_values=False, tabular=bool(mode_now), bucketed=mode_now is True,
                    logits_table=actor_tables["logit"] if actor_tables else None)
    fence()
    rollout_s = time.perf_counter() - t_r
    # K1, the API's episode-gather kernel (States.observations): not part of the default step any more (observations are
    # materialised on demand); timed on its own over the T steps of the last rollout
    k1 = None
    if rank == 0:
        # every launch writes its own [B, 2, A, A] slice of a [T, B, 2, A, A] buffer (906 MB at T = 12, fp32: beyond the 256 MiB Infinity Cache)
        obs = torch.empty((T, local_batch, 2, A, A), dtype=torch.float16 if args.obs_half else torch.float32, device=device)
        bits = torch.empty((T, local_batch), dtype=torch.uint8, device=device)
        idx_all = ep.indices
        for t in range(T):
            rnad_hip.observe(handle, idx_all[t], t & 1, obs=obs[t], half=args.obs_half, mask_bits=bits[t])
        fence_local = torch.cuda.synchronize
        fence_local()
        rnad_hip.prof_enable([rnad_hip.PROF_OBSERVE])
        for _ in range(3):
            for t in range(T):
                rnad_hip.observe(handle, idx_all[t], t & 1, obs=obs[t], half=args.obs_half, mask_bits=bits[t])
        n_obs, obs_ms = rnad_hip.prof_read(rnad_hip.PROF_OBSERVE)
        rnad_hip.prof_enable(False)
        # what plain streams reach on this GPU over the same buffer (the practical ceiling K1's fraction of the 8 TB/s spec peak is to be
        # read against): torch.fill_ (stores only) and torch.copy_ (a read and a write stream)
        practical = {}
        try:
            flat = obs.view(-1)
            half_n = flat.numel() // 2
            for name, fn, moved in (("fill", lambda: flat.fill_(0.0), flat.numel() * flat.element_size()),
                                    ("copy", lambda: flat[:half_n].copy_(flat[half_n: 2 * half_n]), 2 * half_n * flat.element_size())):
                for _ in range(2):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                practical[name + "_TBps"] = moved / (e0.elapsed_time(e1) * 1e-3 / 10) / 1e12
        except Exception as err:
            practical = {"error": str(err)[:200]}
        k1 = (n_obs, obs_ms, practical)
        del obs, bits
    if world > 1:
        t = torch.tensor([rollout_s, host_s], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rollout_s, host_s = t.tolist()
    # ---- N > 1: what the step's one exposed collective costs on THIS node -- the 43 KB gradient bucket all-reduced back to back on the
    # step's stream, eagerly and replayed from a graph of its own (side_legs' `strong_scaling.predicted` ASSUMES ALLREDUCE_US_ASSUMED for it)
    allreduce_us = None
    if world > 1:
        try:
            n_params = sum(p_.numel() for p_ in rn.net.parameters())
            bucket = torch.zeros((n_params,), dtype=torch.float32, device=device)
            for _ in range(20):
                dist.all_reduce(bucket)
            fence()
            t_a = time.perf_counter()
            for _ in range(200):
                dist.all_reduce(bucket)
            torch.cuda.synchronize()
            allreduce_us = {"bytes": 4 * n_params, "eager_back_to_back_us": (time.perf_counter() - t_a) / 200 * 1e6, "calls": 200,
                            "what": "dist.all_reduce of the flat gradient bucket, 200 calls between two fences, max over ranks below"}
            t = torch.tensor([allreduce_us["eager_back_to_back_us"]], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            allreduce_us["eager_back_to_back_us"] = float(t.item())
            del bucket
        except Exception as err:  # (never the reason a scaling line is lost)
            allreduce_us = {"error": str(err)[:300]}

    # read back NOW: emit() may be called from the watchdog thread while the device queue is stuck
    alive = rn.last_episodes.alive.cpu().numpy()[:T]

    # ---- strong scaling: the N = 1 point of this curve -- the SAME global batch on rank 0's GPU alone (no collectives), same mode,
    # same replayed graph; the other ranks wait at the barrier below
    base = None
    if world > 1 and args.scaling == "strong" and not args.no_base_leg:
        if rank == 0:
            try:
                solo, _, solo_step = make_trainer(global_batch, "bench-solo", data_parallel=False)
                for _ in range(6 + args.warmup):
                    solo_step()
                torch.cuda.synchronize()
                n_base = max(1, min(args.steps, 200))
                t_b = time.perf_counter()
                for _ in range(n_base):
                    solo_step()
                torch.cuda.synchronize()
                g_b = getattr(solo, "_graph", None)
                base = {"n_gpus": 1, "global_batch": global_batch, "steps": n_base, "ms_per_step": (time.perf_counter() - t_b) / n_base * 1e3,
                        "net_mode_in_effect": repr(solo._tabular_mode(T, global_batch)),
                        "step_replayed_from_hipGraph": bool(g_b and g_b.get("graph") is not None)}
                del solo, solo_step
                torch.cuda.empty_cache()
            except Exception as err:  # the reference leg must not cost the run its line
                base = {"error": str(err)[:300]}
        dist.barrier()

    # ---- side legs (one GPU, the default workload only; outside the timed region, each its own trainer, replayed graphs)
    side = {}
    default_line = (world == 1 and (A, C, depth, tuple(args.prune), args.width, args.batch_log2) == (3, 1, 6, (0, 0), 256, 20)
                    and args.net_mode == "default" and not args.no_graph and not args.obs_half and not args.no_side_legs)
    if default_line:
        side = side_legs(make_trainer, tree, args, device)

    def emit(elapsed, replayed, note=None):
        if rank != 0:
            return
        # the reference's loop (episode.py:194) runs until every lane is absorbed and counts all B lanes in each of those steps;
        # on the regular c2 tree that is all T = 2 * depth steps, on pruned trees the trailing all-absorbed steps are not counted
        T_ref = int((alive > 0).sum())
        env_steps = global_batch * T_ref * args.steps
        live_slots = int(alive.sum())
        c2_tree = (A, C, depth, tuple(args.prune), args.width) == (3, 1, 6, (0, 0), 256)
        default_workload = c2_tree and global_batch == 1 << 20 and world == 1
        configs2 = c2_tree and global_batch == 1 << 22 and world > 1 and args.scaling == "strong"
        if default_workload:
            which = ", BASELINE.json configs[1]"
        elif configs2:
            which = f", BASELINE.json configs[2] (one 2^22 batch sharded over {world} GPUs" + ("" if world == 8 else "; configs[2] names 8") + ")"
        elif c2_tree:
            which = ", the BASELINE.json configs[1]/[2] tree at another batch"
        else:
            which = f", prune {args.prune[0]}/{args.prune[1]}, threshold {threshold:g} (a BASELINE.json configs[3]/[4]-style variant)"
        what = {False: "every net evaluated on every (t, b) slot, as the reference does",
                "forward": "forward evaluations once per (player, state) row and gathered per slot; backward per slot; bit-identical to dense",
                True: "nets evaluated once per (player, state) row; bucket-ordered rollout; per-slot gradients summed per row in LDS (64-bit "
                      "fixed point, reproducible); one backward over the rows; gradients equal the dense ones up to fp32 summation order"}
        kernels = kernel_report(prof, A, C, args, local_batch, T, live_slots, tree, mode_now)
        dominant = max((k for k in kernels.values() if k.get("single_kernel")), key=lambda k: k["us_per_step"], default=None)
        out = {
            "metric": "env_steps_per_sec (rollout + R-NaD update, one iteration of learn/rnad.py:495-526 per step)",
            "value": env_steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "priming_steps": PRIME,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32" if not args.obs_half else "f32 (fp16 observations)",
            "data": "synthetic",
            "config": {
                "workload": f"depth-{depth} {A}x{A} matrix tree, C={C}, S={tree.index_tensor.shape[0]}, "
                            + (f"2^{args.batch_log2} episodes per GPU" if world > 1 and args.scaling == "weak" else f"global batch 2^{args.batch_log2} episodes")
                            + f" x T={T_ref} env steps, MLP width {args.width}" + which,
                "global_batch": global_batch, "per_gpu_batch": local_batch, "T": T_ref, "T_buffer": T,
                "valid_env_steps_per_step": live_slots * world,
                "parallelism": "single GPU (no collectives)" if world == 1 else
                               f"dp{world} (episodes sharded, RCCL all-reduce of 43 KB grads"
                               + ("; the normalisers of this tree need no collective: every episode has the same length)"
                                  if tree.handle().uniform_length else " + 2 normalisers)"),
            },
            "updates_per_sec": args.steps / elapsed,
            "host_enqueue_ms_per_step": host_s / host_n * 1e3,
            "net_evaluation": {"mode": f"RNaD.tabular = {default_mode!r}" + (" (default)" if args.net_mode == "default" else ""),
                               "in_effect": repr(mode_now), "what": what[mode_now],
                               "step_replayed_from_hipGraph": replayed, "compact_trajectory": bool(args.compact_in_effect),
                               "rollout_and_learner_in_one_launch": bool(getattr(args, "fused_play_learn", False)),
                               "learner_on_leaf_paths": bool(getattr(args, "leaf_paths", False)),
                               # (RNaD.distinct_trajectories: by itself only after RNaD.DISTINCT_AFTER = 4096 updates -- not within a default run)
                               "learner_on_distinct_trajectories_at_the_end": bool(rn._fuse_now() and rn._distinct_now()),
                               "lazy_rows_visited": args.visited_rows or None, "staged_policy_rows": args.policy_rows or None,
                               "rows_after_dedup": args.unique_rows or None,  # (RNaD.dedup_rows: rows with distinct observation bits)
                               "legal_fold": args.fold,
                               "table_rows": 2 * handle.S,  # (player, state) rows of the tree: what the nets are evaluated on without dedup
                               "slots": T * local_batch},
            "other_modes": {name: {"env_steps_per_sec": global_batch * T_ref * E / sec, "updates_per_sec": E / sec,
                                   "ms_per_step": sec / E * 1e3, "steps": E,
                                   "what": what[{"dense_nets": False, "forward": "forward", "tabular_nets": True}[name]]}
                            for name, sec in variants.items()},
            "rollout_env_steps_per_sec": global_batch * T_ref * E / rollout_s,
            "rollout_ms_per_step": rollout_s / E * 1e3,
            "roofline": roofline_of(dominant),
            "kernels": kernels,
            "k1_observe": k1_report(k1, A, args, local_batch),
            "setup": {"tree_generate_and_upload_s": setup_tree_s, "tree_table_bytes": handle.table_bytes},
        }
        if world > 1:
            out["legs_ms_per_step"] = {k: v / args.steps * 1e3 for k, v in legs.items()}
            if not rehearsal:
                assert dist.get_backend() == "nccl" and dist.get_world_size() == args.gpus, "bench.py --gpus N is one RCCL rank per GPU"
            out["collectives"] = {"backend": dist.get_backend(), "rccl_ranks": dist.get_world_size() if dist.get_backend() == "nccl" else None,
                                  "ranks": dist.get_world_size(), "per_step": "all_reduce(2 x f64 normalisers) + all_reduce(43 KB fp32 gradient bucket)",
                                  "shard_rows": bool(args.shard_rows), "allreduce_us": allreduce_us}
            if args.shard_rows:
                A1, nu = A + 1, rn.last_episodes.buckets.plan.n_upper
                per = (2 * handle.S + world - 1) // world
                out["collectives"]["per_step"] += (" + all_gather(policy rows, fast records, records: %d B received per rank) + all_reduce(int64 per-row sums, %d B)"
                                                   % (4 * per * (world - 1) * (int(rnad_hip.lib().rnad_bucket_policy_row_stride(A))
                                                                               + int(rnad_hip.lib().rnad_bucket_fast_record_stride(A))
                                                                               + int(rnad_hip.lib().rnad_bucket_record_stride(A))),
                                                      8 * (2 * handle.S * A1 + rnad_hip.BUCKET_REPLICAS * 2 * max(nu, 1) * A1)))
            if base is not None:
                out["strong_scaling"] = {"base": base}
                if base.get("ms_per_step"):
                    out["strong_scaling"]["speedup_vs_one_gpu_same_batch"] = base["ms_per_step"] / (elapsed / args.steps * 1e3)
            if note:
                out["note"] = note
        out.update(side)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(tree, args, T)
        print(json.dumps(out), flush=True)

    if world == 1:
        emit(elapsed, replayed)
        return
    if args.no_graph:
        emit(elapsed, False)
    else:
        # ---- N > 1: the same K steps replayed from a hipGraph with the RCCL collectives captured inside.  A rank that gets stuck
        # (capture of collectives is the one thing a one-GPU box cannot rehearse) must not cost the run its line: after
        # --graph-timeout seconds every rank leaves, rank 0 printing the eager measurement first.
        import threading

        def bail():
            emit(elapsed, False, note=f"the captured-graph leg did not finish within {args.graph_timeout:g} s: eager measurement")
            sys.stdout.flush()
            os._exit(0)

        watchdog = threading.Timer(args.graph_timeout, bail)
        watchdog.daemon = True
        watchdog.start()
        graph_sec = legs["graph"] = timed_leg(True)
        g = getattr(rn, "_graph", None)
        flag = torch.tensor([1 if (g and g.get("graph") is not None) else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        watchdog.cancel()
        if int(flag.item()) == 1 and graph_sec < elapsed:
            emit(graph_sec, True)
        else:
            emit(elapsed, False, note="eager steps were not slower than the replayed graph" if int(flag.item()) == 1 else
                 "the step could not be captured on every rank")
    dist.barrier()
    dist.destroy_process_group()


FP32_PEAK_TFLOPS = 157.3  # fp32 MFMA == fp32 vector peak (/opt/skills/guides/MI355X_MICROARCH.md)
ALLREDUCE_US_ASSUMED = 15.0  # one 43 KB all-reduce over xGMI, exposed between backward and optimiser (latency-bound)


def side_legs(make_trainer, tree, args, device):
    """What the default line carries beside `value`, each measured like `value` -- its own trainer, primed, the
    step replayed from its captured graph, `--side-steps` replays between two fences:
      other_configs.c4          BASELINE.json configs[3]: depth-8 5x5 tree, chance branching 4, pruned to ~1 M states, batch 2^20
      dedup_off_ms_per_step     the default workload with RNaD.dedup_rows = False: nets and backward on all 2S rows (what a tree without
                                repeated observations pays)
      trained_policy_ms_per_step  the default workload with a SHARP actor (the policy head's output layer scaled: lanes pile up on few
                                trajectories, as after training) and the learner on the distinct trajectories of a work item, which is what
                                RNaD switches to after DISTINCT_AFTER updates
      strong_scaling.predicted  configs[2] without the 8 GPUs: the one-GPU time of the 2^22 batch over (a rank's 2^19-lane step + one
                                exposed 43 KB all-reduce, assumed ALLREDUCE_US_ASSUMED us)."""
    from environment.tree import Tree

    n = max(1, args.side_steps)

    def run(trainer_step, steps=n, prime=60):
        t, _, step = trainer_step
        for _ in range(prime):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        g = getattr(t, "_graph", None)
        return ms, bool(g and g.get("graph") is not None), t

    out = {}
    try:
        ms, rep, t = run(_with(make_trainer(1 << 20, "side-nodedup"), dedup_rows=False))
        out["dedup_off_ms_per_step"] = {"ms_per_step": ms, "step_replayed_from_hipGraph": rep,
                                        "what": "RNaD.dedup_rows = False: table launch and backward on all 2S rows of the tree"}
        del t
    except Exception as err:
        out["dedup_off_ms_per_step"] = {"error": str(err)[:300]}
    try:
        tr = _with(make_trainer(1 << 20, "side-sharp"), distinct_trajectories=True)
        with torch.no_grad():
            for name, p_ in tr[0].net.named_parameters():
                if name.startswith("policy_fc1"):
                    p_.mul_(40.0)
        tr[0].invalidate_tables()
        ms, rep, t = run(tr)
        ep = t.last_episodes
        per_bucket = torch.bincount(ep.buckets.items[: int(ep.buckets.n_items.item()), 2].long(),
                                    weights=ep.buckets.items[: int(ep.buckets.n_items.item()), 1].double())
        out["trained_policy_ms_per_step"] = {"ms_per_step": ms, "step_replayed_from_hipGraph": rep,
                                             "largest_bucket_share_of_lanes": float(per_bucket.max().item()) / (1 << 20),
                                             "what": "a sharp actor (policy_fc1 x 40: near-deterministic, lanes concentrate as after training) with "
                                                     "RNaD.distinct_trajectories = True (what RNaD switches to after DISTINCT_AFTER = 4096 updates)"}
        del t, tr
    except Exception as err:
        out["trained_policy_ms_per_step"] = {"error": str(err)[:300]}
    torch.cuda.empty_cache()
    # the same sharp actor at 2^21 and 2^22 lanes with the learner RNaD picks there on its own -- the leaf paths of the tree (DESIGN.md
    # section 5.6; r06: crowded buckets are counted by the rollout's work items, so a sharpened policy no longer ends it)
    out["trained_policy_large_batches"] = {}
    for log2 in (21, 22):
        try:
            tr = make_trainer(1 << log2, f"side-sharp{log2}")
            with torch.no_grad():
                for name, p_ in tr[0].net.named_parameters():
                    if name.startswith("policy_fc1"):
                        p_.mul_(40.0)
            tr[0].invalidate_tables()
            ms, rep, t = run(tr, steps=max(1, n // 2), prime=30)
            ep = t.last_episodes
            items = ep.buckets.items[: int(ep.buckets.n_items.item())]
            per_bucket = torch.bincount(items[:, 2].long(), weights=items[:, 1].double())
            out["trained_policy_large_batches"][f"2^{log2}"] = {
                "ms_per_step": ms, "step_replayed_from_hipGraph": rep, "leaf_path_learner": getattr(ep.buckets.plan, "leaf", None) is not None,
                "largest_bucket_share_of_lanes": float(per_bucket.max().item()) / (1 << log2)}
            del t, tr
        except Exception as err:
            out["trained_policy_large_batches"][f"2^{log2}"] = {"error": str(err)[:300]}
        torch.cuda.empty_cache()
    try:
        ms19, rep19, t = run(make_trainer(1 << 19, "side-b19"))
        del t
        ms22, rep22, t = run(make_trainer(1 << 22, "side-b22"), steps=max(1, n // 2), prime=20)
        del t
        out["strong_scaling"] = {"predicted": {
            "one_gpu_2p22_ms_per_step": ms22, "rank_share_2p19_ms_per_step": ms19, "assumed_exposed_allreduce_us": ALLREDUCE_US_ASSUMED,
            "assumed_speedup_8_gpus": ms22 / (ms19 + ALLREDUCE_US_ASSUMED * 1e-3), "replayed": bool(rep19 and rep22),
            "measured_on": "ONE GPU (both legs); nothing here is a multi-GPU measurement",
            "what": "BASELINE.json configs[2] predicted from one GPU: the 2^22 batch on one GPU over (a rank's 2^19 lanes + one exposed 43 KB "
                    "all-reduce); both legs measured here, the collective's latency assumed (no multi-GPU node in the build loop)"}}
    except Exception as err:
        out["strong_scaling"] = {"predicted": {"error": str(err)[:300]}}
    torch.cuda.empty_cache()
    try:
        t0 = time.perf_counter()
        c4 = Tree(device=device, max_actions=5, max_transitions=4, depth_bound=8, transition_threshold=0.1)
        c4.generate_native(seed=0, prune=(7, 8))
        gen_s = time.perf_counter() - t0
        ms, rep, t = run(make_trainer(1 << 20, "side-c4", tree=c4), prime=40)
        ep = t.last_episodes
        alive = ep.alive.cpu().numpy()
        T_ref = int((alive[: ep.t_eff + 1] > 0).sum())
        staged = getattr(ep, "staged_rows", None)
        out["other_configs"] = {"c4": {
            "workload": f"BASELINE.json configs[3]: depth-8 5x5 tree, chance branching 4, threshold 0.1, pruned 7/8, S={c4.index_tensor.shape[0]}, "
                        f"batch 2^20 x T={T_ref}", "ms_per_step": ms, "env_steps_per_sec": (1 << 20) * T_ref / (ms * 1e-3),
            "valid_env_steps_per_step": int(alive[: ep.t_eff + 1].sum()), "updates_per_sec": 1e3 / ms, "step_replayed_from_hipGraph": rep,
            "lazy_rows_visited": int(t.last_rows.count.item()) if t.last_rows is not None else None,
            "staged_policy_rows": int(sum(r.count.item() for r in staged)) if staged else None,
            "table_rows": 2 * c4.handle().S, "tree_generate_and_upload_s": gen_s, "steps": n}}
        try:
            out["other_configs"]["c4"]["kernels"] = c4_kernel_table(t, ep, c4, T_ref)
        except Exception as err:
            out["other_configs"]["c4"]["kernels"] = {"error": str(err)[:300]}
        del t, c4
    except Exception as err:
        out["other_configs"] = {"c4": {"error": str(err)[:300]}}
    torch.cuda.empty_cache()
    return out


def c4_kernel_table(t, ep, tree, T):
    """Per kernel of the configs[3] step: the counter evidence of tools/pmc_config.sh (profiles/r06_pmc_c4.json: FETCH_SIZE / WRITE_SIZE / SQ / TCP passes
    over the eagerly enqueued step, durations of those passes) and, for the two gather kernels, their algorithmic bytes -- the same yardsticks
    as the headline `roofline`, for the configuration where no degeneracy of the tree helps."""
    import rnad_hip as rh

    pmc = _load(PMC_C4_FILE)
    kernels, matches = pmc.get("kernels", {}), pmc.get("source_hash") == rh.source_hash()
    handle, B, A = tree.handle(), ep.batch_size, tree.max_actions
    stored = rh.stored_state_slots(handle, ep.buckets, T)
    rb = ep.buckets.plan.rel_bytes
    visited = int(t.last_rows.count.item()) if t.last_rows is not None else 2 * handle.S
    algo = {"k_bucket_rollout_items": (4 * B + stored * rb + 12 * B, "lane id 4 read; relative state per stored slot, packed actions 8 + reward 4 per lane written"),
            "k_bucket_learn_c": (stored * rb + 12 * B + visited * (4 + 4 * A) * 4, "relative state per stored slot, packed actions 8 + reward 4 per lane, "
                                 "the fast records of the visited rows once each (%d B)" % ((4 + 4 * A) * 4))}
    table = {}
    for name, c in kernels.items():
        us = c.get("duration_us_sq_pass") or c.get("duration_us_fetch_pass")
        if not us:
            continue
        sec = us * 1e-6
        e = {"avg_launch_us_counter_pass": us, "launches_in_pass": c.get("launches"), "traffic": c.get("traffic_bytes_per_launch")}
        if c.get("traffic_bytes_per_launch"):
            e["hbm_frac_on_counter_traffic"] = c["traffic_bytes_per_launch"] / sec / 1e9 / HBM_PEAK_GBS
        if name in algo:
            e.update(algorithmic_bytes_per_launch=algo[name][0], bytes_model=algo[name][1], hbm_frac_on_algorithmic_bytes=algo[name][0] / sec / 1e9 / HBM_PEAK_GBS)
        if c.get("SQ_INSTS_VALU"):
            e["valu_wave_instructions"] = c["SQ_INSTS_VALU"]
            e["issue_frac_at_2p4_cycles"] = c["SQ_INSTS_VALU"] * 2.4 / (SIMDS * sec * CLOCK_HZ)
        if c.get("SQ_WAVE_CYCLES"):
            e["parked_on_waitcnt"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
            e["issue_stalled"] = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        if c.get("TCP_TOTAL_CACHE_ACCESSES_sum") and c.get("duration_us_tcp_pass"):
            e["l1_accesses"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"]
            e["l1_clock_enabled_frac"] = c.get("TCP_GATE_EN1_sum", 0.0) / (256 * c["duration_us_tcp_pass"] * 1e-6 * CLOCK_HZ)
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES_mfma_pass") and c.get("duration_us_mfma_pass"):
            e["matrix_pipe_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES_mfma_pass"] / (SIMDS * c["duration_us_mfma_pass"] * 1e-6 * CLOCK_HZ)
        table[name] = e
    return {"counters_from_this_build": matches, "counters_source": os.path.relpath(PMC_C4_FILE, ROOT),
            "note": "durations are those of the counter passes (eager launches under rocprofv3); issue_frac prices every VALU wave-instruction at 2.4 cycles "
                    "(the mean of the measured class costs, profiles/r0N_valu_issue.json)", "per_kernel": table}


def _with(trainer_step, **attrs):
    for k, v in attrs.items():
        setattr(trainer_step[0], k, v)
    return trainer_step


def mode_now_is_true(rn, T, local_batch):
    return rn._tabular_mode(T, local_batch) is True


def kernel_report(prof, A, C, args, B, T, live_slots, tree, mode_now):
    """Per bracketed kernel: time per step and, where one kernel is bracketed alone, its algorithmic bytes / flops per launch
    (DESIGN.md section 5 states each figure) against the HBM or fp32-MFMA peak."""
    import rnad_hip as rh

    rec = ((4 * A + 3 + 3) & ~3) * 4  # bytes of a row record (rnad_bucket_record_stride)
    S2 = 2 * tree.handle().S
    # input features of the first layer as the kernels of this mode see them: the legal fold (include/rnad_hip.h) leaves A^2 + 1 (+ padding)
    K = ((A * A + 2) & ~1) if getattr(args, "fold", False) else 2 * A * A
    W = args.width
    rem = (K + 1) % 16
    feat = 16 * ((K + 1) // 16 + (1 if rem > 4 else 0)) + (4 if 0 < rem <= 4 else 0)
    slots = B * T
    compact = bool(args.compact_in_effect)
    fast = (4 + 4 * A) * 4  # bytes of a fast row record (rnad_bucket_fast_record_stride)
    if compact:
        rb, stored = args.rel_bytes, args.stored_slots  # relative states: below the cut of the tree only (include/rnad_hip.h "Compact trajectory")
        rollout_model = (4 * B + stored * rb + 8 * B + 4 * B,
                         f"per lane: lane id 4 (read; the decisions above the cut are one word per work item); per slot BELOW the cut (and the "
                         f"final state): relative state {rb} (written; {stored} of the {slots + B} slots of this batch: the steps a lane shares with its "
                         "bucket are not stored); per lane: packed actions 8, reward 4 (written).  Policy rows and transition records are "
                         "gathered from the L2-resident tables")
        learn_model = (stored * rb + B * (8 + 4) + S2 * fast,
                       f"per stored slot: relative state {rb}; per lane: packed actions 8, reward 4; the 2S fast records (64 B at A = 3) once "
                       "each -- gathered per slot below the cut from L2 / MALL after the first touch, read through the scalar cache for the "
                       "steps a workgroup shares; sums stay in LDS")
    else:
        rollout_model = (4 * B + slots * (4 + 1 + 4 * A + 4 + 4) + 4 * B,
                         "lane_ids 4 B/lane + per slot: state 4, legal bits 1, policy 4A, action 4, reward 4 (+ final state 4 B/lane)")
        learn_model = (live_slots * (4 + 4 + 4 * A) + (live_slots // 2) * 4,
                       "per live slot: state 4, action 4, acting policy 4A; reward 4 on column steps; sums stay in LDS")
    fused = bool(getattr(args, "fused_play_learn", False)) and compact
    if fused:
        learn_model = (4 * B + stored * rb + B * (8 + 4) + S2 * fast,
                       f"per lane: lane id 4 (read); per slot below the cut (and the final state): relative state {rb} (written: the trajectory "
                       "the API hands out; read back by the thread that wrote it, from L2); per lane: packed actions 8, reward 4 (written); the 2S "
                       "fast records (64 B at A = 3) once each -- gathered per slot below the cut from L2 / MALL, through the scalar cache for "
                       "the steps a workgroup shares; policy rows and transition records from the L2-resident tables; sums stay in LDS")
    model = {
        # bytes the kernel must move per launch (streams; the L2-resident tables it gathers from are not HBM traffic)
        rh.PROF_BUCKET_ROLLOUT: ("hbm",) + rollout_model,
        rh.PROF_BUCKET_LEARN: ("hbm",) + learn_model,
        rh.PROF_BUCKET_KEYS: ("hbm", 4 * B + 8 * B, "keys 4 B/lane and drawn decisions 8 B/lane written (+ one histogram row per 4096 lanes); the upper "
                              "states' tables are staged in LDS once per workgroup"),
        rh.PROF_OBSERVE: ("hbm", B * (4 + 8 * A * A + 2 * A * A * (2 if args.obs_half else 4) + 4 * A), "SURVEY 8d"),
    }
    out = {}
    default_shape = (A, C, args.depth, tuple(args.prune), B, args.width, args.obs_half) == (3, 1, 6, (0, 0), 1 << 20, 256, False)
    pmc, pmc_matches = pmc_counters() if default_shape else ({}, False)
    rel = "unsigned char" if rh.bucket_plan(tree.handle(), B) is not None and rh.bucket_plan(tree.handle(), B).rel_bytes == 1 else "unsigned short"
    # (the kernel a scope ran depends on the tree: LDS walk / global-table walk; compact / dense trajectory)
    pmc_name = {rh.PROF_BUCKET_KEYS: "k_bucket_keys_lds" if "k_bucket_keys_lds" in pmc else "k_bucket_keys",
                rh.PROF_BUCKET_ROLLOUT: "k_bucket_rollout_items" if compact else "k_bucket_rollout",
                rh.PROF_BUCKET_LEARN: "k_bucket_play_learn" if fused else "k_bucket_learn_c" if compact else "k_bucket_learn",
                rh.PROF_OBSERVE: "k_observe"}
    mix_name = {rh.PROF_BUCKET_KEYS: f"k_bucket_keys_lds<{A}, 1, 4096>", rh.PROF_BUCKET_ROLLOUT: f"k_bucket_rollout_items<{A}, {rel}, 1>",
                rh.PROF_BUCKET_LEARN: f"k_bucket_play_learn<{A}, {rel}, false>" if fused else f"k_bucket_learn_c<{A}, {rel}, false, false>"}
    units = {rh.PROF_BUCKET_KEYS: (B, "lane"), rh.PROF_BUCKET_ROLLOUT: (slots, "slot"), rh.PROF_BUCKET_LEARN: (max(live_slots, 1), "live slot")}
    for k, p in prof.items():
        e = dict(p)
        e["single_kernel"] = (k in (rh.PROF_BUCKET_KEYS, rh.PROF_BUCKET_ROLLOUT, rh.PROF_BUCKET_LEARN, rh.PROF_OBSERVE, rh.PROF_MLP, rh.PROF_MLP_BWD)
                              and not (k == rh.PROF_BUCKET_LEARN and getattr(args, "leaf_paths", False)))
        if k in model:
            bound, nbytes, how = model[k]
            sec = p["avg_launch_us"] * 1e-6
            gbs = nbytes / sec / 1e9
            c = pmc.get(pmc_name.get(k, ""), {})
            e.update(bound="hbm", algorithmic_bytes_per_launch=nbytes, achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS,
                     bytes_model=how, traffic=c.get("traffic_bytes_per_launch"))
            if k in units:
                issue = {"frac": None, "model": "SQ_INSTS_VALU x sum_class(share x cycles) / (1024 SIMDs x duration x 2.4 GHz)"}
                cpi, detail = issue_cycles_per_instruction(mix_name[k]) if compact or k == rh.PROF_BUCKET_KEYS else (None, None)
                n_inst = c.get("SQ_INSTS_VALU")
                if n_inst and cpi:
                    issue.update(frac=n_inst * cpi / (SIMDS * sec * CLOCK_HZ), valu_wave_instructions_per_launch=n_inst,
                                 cycles_per_instruction=cpi, valu_instructions_per_unit={"per": units[k][1], "value": n_inst * 64 / units[k][0]},
                                 lds_wave_instructions_per_launch=c.get("SQ_INSTS_LDS"), **detail)
                wait = None
                if c.get("SQ_WAVE_CYCLES"):
                    wait = {"parked_on_waitcnt": c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAIT_ANY") else None,
                            "issue_stalled": c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAIT_INST_ANY") else None,
                            "what": "SQ_WAIT_ANY / SQ_WAVE_CYCLES and SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES of the counter pass"}
                l1 = None
                if c.get("TCP_TOTAL_CACHE_ACCESSES_sum") and c.get("duration_us_tcp_pass"):
                    l1 = {"cache_accesses_per_launch": c["TCP_TOTAL_CACHE_ACCESSES_sum"],
                          "tcp_clock_enabled_frac": c.get("TCP_GATE_EN1_sum", 0.0) / (256 * c["duration_us_tcp_pass"] * 1e-6 * CLOCK_HZ),
                          "what": "TCP_TOTAL_CACHE_ACCESSES_sum and TCP_GATE_EN1_sum / (256 CUs x launch duration x 2.4 GHz) of their own counter "
                                  "pass: the vector L1s' access count and busy share -- what holds the gather kernels"}
                e.update(issue=issue, wait=wait, l1=l1, counters_from_this_build=pmc_matches if c else None,
                         counters_source="%s (separate rocprofv3 --pmc passes: FETCH_SIZE | WRITE_SIZE | SQ_*; traffic = 2 x "
                                         "FETCH_SIZE + WRITE_SIZE), %s, %s" % tuple(os.path.relpath(f, ROOT) for f in (PMC_FILE, ISA_MIX_FILE, ISSUE_FILE)) if c else None)
        out[p["name"]] = e
    # the fused MLP kernels: flops the matrix cores execute per sample (first layer of a head: 2 K W; relu + second layer run on the
    # VALU; backward: recompute of both heads + dW0 over the augmented input padded to its MFMA tiles)
    uniform = tree.handle().uniform_length
    visited = args.visited_rows  # lazy rows: the value heads and the backward run on the rows the batch visited
    uniq = getattr(args, "unique_rows", 0)  # (0: every row evaluated)
    bwd_samples = (visited or uniq or S2) if mode_now is True else (live_slots if not uniform else slots)
    def mlp_counters(entry, name):
        """Counter evidence of an MLP kernel (profiles/r0N_pmc.json): HBM traffic, the matrix pipe's busy share of the launch's SIMD cycles,
        VALU wave-instructions, wait shares."""
        c = pmc.get(name, {})
        if not c:
            return
        entry["traffic"] = c.get("traffic_bytes_per_launch")
        entry["counters_from_this_build"] = pmc_matches
        dur = c.get("duration_us_mfma_pass")
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES_mfma_pass") and dur:
            entry["matrix_pipe"] = {"busy_cycles_per_launch": c["SQ_VALU_MFMA_BUSY_CYCLES_mfma_pass"],
                                    "busy_frac_of_simd_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES_mfma_pass"] / (SIMDS * dur * 1e-6 * CLOCK_HZ),
                                    "valu_wave_instructions_per_launch": c.get("SQ_INSTS_VALU"),
                                    "what": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x launch duration x 2.4 GHz), its own rocprofv3 --pmc pass; fp32 "
                                            "MFMA and VALU share the ALUs on gfx950, so the VALU instructions of the epilogues add to it"}
        if c.get("SQ_WAVE_CYCLES"):
            entry["wait"] = {"parked_on_waitcnt": c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"],
                             "issue_stalled": c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"],
                             "what": "SQ_WAIT_ANY / SQ_WAVE_CYCLES and SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES of the counter pass"}

    if rh.PROF_MLP_BWD in prof:
        p = prof[rh.PROF_MLP_BWD]
        flops = (2.0 * K * 2 * W + 2.0 * feat * 2 * W) * bwd_samples
        tf = flops / (p["us_per_step"] * 1e-6) / 1e12  # the launches of a step together (backward + its reduction)
        out[p["name"]].update(bound="mfma", achieved=tf, peak=FP32_PEAK_TFLOPS, unit="TFLOP/s", frac=tf / FP32_PEAK_TFLOPS,
                              samples_per_step=bwd_samples,
                              flops_model=f"per sample 2*K*2W (recompute) + 2*feat*2W (dW0 tiles), K = {K}" + (" (legal fold)" if getattr(args, "fold", False) else ""))
        mlp_counters(out[p["name"]], "rnad_mlp::k_mlp_backward_t")
    if rh.PROF_MLP in prof and mode_now is True:
        p = prof[rh.PROF_MLP]
        # learner: both heads, target: value head, on the 2S rows (regularisation tables are cached); lazy rows: the two value heads
        # on the visited rows only
        policy_rows = (getattr(args, "policy_rows", 0) or S2) if visited else S2
        flops = 2.0 * K * W * (3 * (uniq or S2) if not visited else policy_rows + 2 * visited)
        tf = flops / (p["us_per_step"] * 1e-6) / 1e12
        out[p["name"]].update(bound="mfma", achieved=tf, peak=FP32_PEAK_TFLOPS, unit="TFLOP/s", frac=tf / FP32_PEAK_TFLOPS, samples_per_step=uniq or S2,
                              flops_model=f"2*K*W per head and row, K = {K} input features" + (" (legal fold)" if getattr(args, "fold", False) else "")
                                          + ": learner 2 heads + target value head"
                                          + (f"; on the {uniq} distinct observations of the tree's {S2} rows (RNaD.dedup_rows)" if uniq and not visited else "")
                                          + (f"; lazy rows: policy head on {policy_rows} rows (upper states + the groups the batch descends into), "
                                             f"the two value heads on the {visited} visited rows" if visited else ""))
        mlp_counters(out[p["name"]], "k_rows_forward_records" if "k_rows_forward_records" in pmc else "k_mlp_forward")
    return out


def roofline_of(k):
    if k is None:
        return None
    binds = None
    if k.get("bound") == "hbm" and (k.get("issue") or k.get("wait")):
        # the contract's `bound` names the roof the bytes are priced against; what actually holds a gather kernel of this path is below
        iss, wt = (k.get("issue") or {}).get("frac"), (k.get("wait") or {}).get("parked_on_waitcnt")
        binds = ("dependent-gather latency and the vector L1's access rate, not HBM: VALU issue %s of its roof, %s of a resident wave's time "
                 "parked on s_waitcnt" % ("%.2f" % iss if iss else "n/a", "%.2f" % wt if wt else "n/a"))
    bound = k.get("bound")
    if binds is not None:
        bound = "l1/latency"  # what holds the kernel; `frac` stays its algorithmic bytes over the HBM peak, as the contract prices it
    r = {"kernel": k["name"], "bound": bound, "priced_against": k.get("bound"), "binds": binds, "achieved": k.get("achieved"), "peak": k.get("peak"), "unit": k.get("unit"),
         "frac": k.get("frac"), "traffic": k.get("traffic"), "avg_launch_us": k["avg_launch_us"], "launches_per_step": k["launches_per_step"],
         "bytes_per_launch": k.get("algorithmic_bytes_per_launch"), "bytes_model": k.get("bytes_model"),
         "share_of_step_us": k["us_per_step"],
         "measured": "hipEvents around each launch, eager leg of the same steps after the timed region (the timed steps replay a graph)"}
    for extra in ("issue", "wait", "l1", "counters_from_this_build", "counters_source", "flops_model", "samples_per_step", "matrix_pipe"):
        if k.get(extra) is not None:
            r[extra] = k[extra]
    return r


def k1_report(k1, A, args, B):
    if not k1 or not k1[0]:
        return None
    n, ms = k1[0], k1[1]
    practical = k1[2] if len(k1) > 2 else None
    us = ms * 1e3 / n
    algo = B * (4 + 8 * A * A + 2 * A * A * (2 if args.obs_half else 4) + 4 * A)
    out = {"kernel": "k_observe (K1, States.observations: the API's episode-gather kernel; not in the default step any more)",
           "avg_launch_us": us, "launches": n, "algorithmic_bytes_per_launch_survey_8d": algo,
           "algorithmic_GBps": algo / (us * 1e-6) / 1e9,
           # SURVEY 8(d) prices an env step at idx 4 + two gathered 4 A^2-byte rows + the observation + a 4 A-byte mask; the kernel reads ONE
           # 48-byte node row per lane, which comes from L2 (the tables are 3 MB), and writes the mask as one byte: by that model the launch
           # would exceed the HBM peak -- the model counts bytes that never travel, so the fraction that counts is the counters' below
           "frac_of_hbm_peak_by_survey_8d_model": algo / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
    if practical:
        out["plain_streams_on_this_gpu"] = dict(practical, what="torch.fill_ (stores only) and torch.copy_ (read + write) over the same buffer, TB/s: the "
                                                "practical ceilings of a write stream and of a mixed stream -- K1 reads a state id and a node row and writes "
                                                "the observation, i.e. it is a mixed stream")
    traffic = k1_traffic(A, args)
    if traffic:
        out.update(counter_bytes_per_launch=traffic, frac_of_hbm_peak_from_counter_bytes=traffic / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                   note="counter bytes = 2 x FETCH_SIZE + WRITE_SIZE of a separate rocprofv3 --pmc pass (profiles/r0N_pmc.json) over launches that "
                        "each write their own slice of a [T, B, 2, A, A] buffer (906 MB: beyond the Infinity Cache); the SURVEY model over-counts "
                        "(node rows are L2 hits, the mask travels as 1 byte), so the fraction is taken from the counters")
    return out


def k1_traffic(A, args):
    """HBM bytes per K1 launch from the separate rocprofv3 PMC passes over tools/k1_pmc.py (FETCH_SIZE doubled per the gfx950
    correction + WRITE_SIZE), valid for the default workload and for the build the counters were taken on; None otherwise."""
    if A != 3 or args.batch_log2 != 20 or args.gpus != 1 or args.obs_half:
        return None
    return pmc_counters()[0].get("k_observe", {}).get("traffic_bytes_per_launch")


def cpu_baseline(tree, args, T):
    """The same step on the host cores, two ways, on bounded samples of the same workload:
      port       the C oracle (OpenMP) + PyTorch-CPU MLP (oracle/port.py) -- `value`
      torch_cpu  the reference's op sequence in plain PyTorch on the CPU (oracle/torch_port.py, pinned against the reference's own
                 gradients by tests/test_oracle_golden.py) -- what the reference is on this box without a GPU"""
    from oracle.port import CpuTrainer
    from oracle.torch_port import TorchCpuTrainer

    arrs = dict(index=tree.index_tensor.cpu().numpy(), value=tree.value_tensor.cpu().numpy(), chance=tree.chance_tensor.cpu().numpy(),
                expected_value=tree.expected_value_tensor.cpu().numpy(), legal=tree.legal_tensor.cpu().numpy(),
                depth_bound=tree.depth_bound)
    cores = CPU_THREADS
    torch.set_num_threads(cores)
    ct = CpuTrainer(arrs, width=args.width, chunk_rows=1 << 14)  # (r06: the MLP in cache-sized row chunks -- 4.6x faster updates than whole-batch matrices)
    if args.cpu_lanes_log2 is None:
        args.cpu_lanes_log2 = min(args.batch_log2, 20)
    lanes = 1 << args.cpu_lanes_log2
    ct.step(min(lanes, 4096), seed=0)  # warm-up (thread pools, page faults)
    t0 = time.perf_counter()
    n, roll, upd = 0, 0.0, 0.0
    while n < 1 or (time.perf_counter() - t0 < 12 and n < 6):
        Tc, r, u = ct.step(lanes, seed=1 + n)
        roll += r
        upd += u
        n += 1
    dt = time.perf_counter() - t0
    full = (1 << args.batch_log2) * Tc / (lanes * Tc * n / dt)
    out = {
        "value": lanes * Tc * n / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
        # (one step of the full 2^batch-log2 batch at this rate: beyond the minute the default run may spend here, hence the sample)
        "full_batch_step_estimate_s": full,
        "sample": f"{n} full step(s) (rollout + update) of 2^{args.cpu_lanes_log2} episodes x T={Tc} on the same tree; "
                  f"C oracle (OpenMP) + PyTorch-CPU MLP in 16 384-row chunks, {cores} threads",
        "rollout_env_steps_per_sec": lanes * Tc * n / roll, "updates_per_sec_at_sample_batch": n / dt,
        "host": {"nproc": os.cpu_count(), "torch_num_threads": torch.get_num_threads()},
    }
    try:
        tt = TorchCpuTrainer(arrs, width=args.width)
        tl = 1 << args.cpu_torch_lanes_log2
        tt.step(min(tl, 4096), seed=0)
        t0 = time.perf_counter()
        m, roll_t = 0, 0.0
        while m < 1 or (time.perf_counter() - t0 < 10 and m < 6):
            Tt, r, _ = tt.step(tl, seed=1 + m)
            roll_t += r
            m += 1
        dt_t = time.perf_counter() - t0
        out["torch_cpu"] = {"value": tl * Tt * m / dt_t, "unit": "env-steps/s", "cores": torch.get_num_threads(), "kind": "port",
                            "sample": f"{m} full step(s) of 2^{args.cpu_torch_lanes_log2} episodes x T={Tt}: the reference's op sequence "
                                      "(episode.py:194-212, rnad.py:365-425, vtrace.py) in plain PyTorch on the CPU",
                            "rollout_env_steps_per_sec": tl * Tt * m / roll_t, "updates_per_sec_at_sample_batch": m / dt_t}
    except Exception as err:  # the second leg must not cost the run its line
        out["torch_cpu"] = {"error": str(err)[:300]}
    return out


if __name__ == "__main__":
    main()
