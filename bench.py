#!/usr/bin/env python3
"""bench.py -- the R-NaD self-play hot path on MI355X: env-steps/s and updates/s at batch 2^20.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

One "step" = one iteration of the reference's training loop (learn/rnad.py:495-526): roll out a batch of episodes with
the learner net (Episodes.generate), sample the buffer, RNaD.__learn (4 MLP evaluations, fused V-trace/NeuRD kernel,
backward), Adam, EMA target.  `value` times RNaD's default net-evaluation mode (tabular = "forward": the forward evaluations
run once per (player, state) observation and are gathered per slot, the backward runs per slot -- every result bit-identical
to evaluating every net on every slot); `other_modes` times that dense mode and tabular = True in the same process.  Workload = BASELINE.json configs[1]: depth-6 ternary (3x3) tree, C = 1, 66 431 states,
GLOBAL batch 2^20 episodes x 12 env steps, MLP width 256, fp32.  N > 1 shards the episodes over the ranks (strong
scaling, BASELINE north_star) with one RCCL all-reduce of the 2 loss normalisers and one of the 43 KB gradient bucket.

Prints ONE JSON line on rank 0.  `value` = env steps of all ranks / wall time of the K timed steps (inputs resident in HBM;
the tree is generated and uploaded before the timed region).  `roofline` is for K1, the episode-gather kernel
(rnad_observe): algorithmic bytes per launch (160 B per env step at A = 3 fp32) / mean launch duration
measured with hipEvents on the launching stream inside the timed region.  `cpu_baseline` times the CPU port
(oracle/port.py: C oracle + PyTorch-CPU MLP) of the same step on a bounded sample, rank 0, N = 1 only.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch-log2", type=int, default=20, help="log2 of the GLOBAL episode batch")
    ap.add_argument("--depth", type=int, default=6)
    ap.add_argument("--actions", type=int, default=3)
    ap.add_argument("--transitions", type=int, default=1)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--prune", type=int, nargs=2, default=(0, 0), metavar=("NUM", "DEN"),
                    help="each child's depth drops by 2 more with probability NUM/DEN (reference main.py:37); configs[3] uses a pruned tree")
    ap.add_argument("--threshold", type=float, default=None, help="transition_threshold (default 0 for C=1, 0.5/C otherwise)")
    ap.add_argument("--tree-seed", type=int, default=0)
    ap.add_argument("--net-mode", choices=("default", "dense", "forward", "tabular"), default="default",
                    help="RNaD.tabular for the timed `value`: dense = False, forward = 'forward' (RNaD's default), tabular = True; "
                         "the other modes are reported under other_modes either way")
    ap.add_argument("--obs-half", action="store_true", help="fp16 observations (BASELINE configs[4])")
    ap.add_argument("--cpu-lanes-log2", type=int, default=15, help="episodes in the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    import rnad_hip
    from environment.episode import Buffer
    from environment.tree import Tree
    from learn.rnad import RNaD

    A, C, depth = args.actions, args.transitions, args.depth
    global_batch = 1 << args.batch_log2
    assert global_batch % world == 0
    local_batch = global_batch // world

    # ---- setup (untimed): tree tables into HBM, nets, optimizer
    t0 = time.perf_counter()
    threshold = args.threshold if args.threshold is not None else (0.0 if C == 1 else 0.5 / C)
    tree = Tree(device=device, max_actions=A, max_transitions=C, depth_bound=depth, transition_threshold=threshold)
    tree.generate_native(seed=args.tree_seed, prune=tuple(args.prune))
    tree.handle()
    setup_tree_s = time.perf_counter() - t0
    os.environ["RNAD_SAVE_DIR"] = tempfile.mkdtemp(prefix="rnad_bench_")
    torch.manual_seed(0)
    rn = RNaD(tree=tree, device=device, directory_name=f"bench-r{rank}", batch_size=global_batch, eta=0.2, b1_adam=0.0,
              net_params={"type": "MLP", "max_actions": A, "width": args.width})
    rn.initialize()
    rn.obs_half = args.obs_half
    if args.net_mode != "default":
        rn.tabular = {"dense": False, "forward": "forward", "tabular": True}[args.net_mode]
    with torch.no_grad():
        # the general case of rnad.py:382: two DISTINCT regularisation nets and 0 < alpha < 1 (four net evaluations per update).
        # During m == 0 the two coincide and for alpha == 1 one of them has weight 0; RNaD then evaluates one net less --
        # that is not what is timed here.
        for p in rn.net_reg_.parameters():
            p.mul_(1.001)
    buffer = Buffer(rn.n_batches_per_buffer)
    delta_m = 10_000

    def one_step(i):
        alpha = 1 if i > delta_m / 2 else i * 2 / delta_m
        rn.train_step(buffer, alpha)
        rn.total_steps += 1

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # setup (untimed, before the caller's warmup): two priming steps, so that every code object, the caching allocator's pools
    # and RCCL's channels exist whatever --warmup is (the first launches of a kernel load its code object: tens of ms)
    for i in range(2):
        one_step(i)
    fence()
    for i in range(args.warmup):
        one_step(i)
    fence()
    # timed region: hipEvent brackets around K1 only (every bracket costs a few us of dispatch latency)
    rnad_hip.prof_enable([rnad_hip.PROF_OBSERVE])
    t_start = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t_start
    n_obs, obs_ms = rnad_hip.prof_read(rnad_hip.PROF_OBSERVE)
    T = rn.last_episodes.t_eff + 1
    # the other kernels: the same steps again with every kernel bracketed, outside the headline timing
    rnad_hip.prof_enable(True)
    for i in range(args.steps):
        one_step(args.warmup + args.steps + i)
    fence()
    n_act, act_ms = rnad_hip.prof_read(rnad_hip.PROF_ACT)
    n_learn, learn_ms = rnad_hip.prof_read(rnad_hip.PROF_LEARN)
    n_mlp, mlp_ms = rnad_hip.prof_read(rnad_hip.PROF_MLP)
    n_bwd, bwd_ms = rnad_hip.prof_read(rnad_hip.PROF_MLP_BWD)
    rnad_hip.prof_enable(False)
    # the same step in the other net-evaluation modes of RNaD (reported separately, NOT `value`):
    #   dense_nets   RNaD.tabular = False: every net on every (t, b) slot, as the reference does
    #   forward      RNaD.tabular = "forward" (the default): forward evaluations once per (player, state), backward per slot;
    #                every result bit-identical to dense_nets
    #   tabular_nets RNaD.tabular = True: per-slot gradients summed per (player, state) row, one backward over the 2S rows;
    #                same rollouts and losses, weight gradients equal up to fp32 summation order
    default_mode = rn.tabular
    base = args.warmup + 2 * args.steps
    variants = {}
    for name, mode in (("dense_nets", False), ("forward", "forward"), ("tabular_nets", True)):
        if mode == default_mode or (mode and 8 * tree.handle().S > T * local_batch):
            continue
        rn.tabular = mode
        one_step(base)
        one_step(base)
        fence()
        t_s = time.perf_counter()
        for i in range(args.steps):
            one_step(base + 1 + i)
        fence()
        variants[name] = time.perf_counter() - t_s
        base += args.steps + 1
    rn.tabular = default_mode
    # rollout alone (Episodes.generate, reference episode.py:175-230), outside the headline timed region
    from environment.episode import Episodes
    fence()
    t_r = time.perf_counter()
    for i in range(args.steps):
        Episodes(tree, local_batch, seed=1000 + i, lane_offset=rank * local_batch, obs_half=args.obs_half).generate(
            rn.net, trim=False, skip_absorbed=True, store_values=False,
            tabular=bool(rn.tabular) and 8 * tree.handle().S <= T * local_batch)  # as RNaD.train_step calls it
    fence()
    rollout_s = time.perf_counter() - t_r
    if world > 1:
        names = sorted(variants)
        t = torch.tensor([elapsed, rollout_s] + [variants[k] for k in names], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, rollout_s, *rest = t.tolist()
        variants = dict(zip(names, rest))

    if rank == 0:
        # the reference's loop (episode.py:194) runs until every lane is absorbed and counts all B lanes in each of those steps;
        # on the regular c2 tree that is all T = 2 * depth steps, on pruned trees the trailing all-absorbed steps are not counted
        alive = rn.last_episodes.alive.cpu().numpy()[:T]
        T_ref = int((alive > 0).sum())
        env_steps = global_batch * T_ref * args.steps
        default_workload = (A, C, depth, tuple(args.prune), args.batch_log2, args.width) == (3, 1, 6, (0, 0), 20, 256)
        obs_elem = 2 if args.obs_half else 4
        k1_bytes_per_step = 4 + 8 * A * A + 2 * A * A * obs_elem + 4 * A  # idx + ev row + legal row + obs + mask
        k1_bytes_per_launch = k1_bytes_per_step * local_batch
        k1_avg_s = obs_ms / 1e3 / max(n_obs, 1)
        achieved = k1_bytes_per_launch / k1_avg_s / 1e9 if n_obs else 0.0
        learn_bytes = (69 + 16) * local_batch * T if A == 3 else None
        # fused MLP: flops the matrix cores execute per sample (first layer, both heads; relu + second layer run on the VALU)
        K = 2 * A * A
        mlp_flops_per_sample = 2.0 * K * 2 * args.width
        # backward per sample: recompute of the first layer + dW0 over the augmented input padded to its MFMA feature tiles
        # (16-wide tiles, plus a 4-wide one when at most 4 features are left over; csrc/mlp_bwd.hip)
        rem = (K + 1) % 16
        feat = 16 * ((K + 1) // 16 + (1 if rem > 4 else 0)) + (4 if 0 < rem <= 4 else 0)
        bwd_flops_per_sample = mlp_flops_per_sample + 2.0 * feat * 2 * args.width
        n_live = int(alive.sum()) if rn.skip_absorbed and not tree.handle().uniform_length else local_batch * T
        bwd_tflops = bwd_flops_per_sample * n_live / (bwd_ms / max(n_bwd, 1) / 1e3) / 1e12 if n_bwd else None
        what = {False: "every net evaluated on every (t, b) slot, as the reference does",
                "forward": "forward evaluations once per (player, state) row and gathered per slot; backward per slot; bit-identical to dense",
                True: "forward evaluations AND gradient sums per (player, state) row; gradients equal up to fp32 summation order"}
        out = {
            "metric": "env_steps_per_sec (rollout + R-NaD update, one iteration of learn/rnad.py:495-526 per step)",
            "value": env_steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32" if not args.obs_half else "f32 (fp16 observations)",
            "data": "synthetic",
            "config": {
                "workload": f"depth-{depth} {A}x{A} matrix tree, C={C}, S={tree.index_tensor.shape[0]}, global batch 2^{args.batch_log2}"
                            f" episodes x T={T_ref} env steps, MLP width {args.width}"
                            + (", BASELINE.json configs[1]" if default_workload else
                               f", prune {args.prune[0]}/{args.prune[1]}, threshold {threshold:g} (a BASELINE.json configs[3]/[4]-style variant)"),
                "global_batch": global_batch, "per_gpu_batch": local_batch, "T": T_ref, "T_buffer": T,
                "valid_env_steps_per_step": int(alive.sum()) * world,
                "parallelism": f"dp{world} (episodes sharded, RCCL all-reduce of 2 normalisers + 43 KB grads)",
            },
            "updates_per_sec": args.steps / elapsed,
            "net_evaluation": {"mode": f"RNaD.tabular = {default_mode!r}" + (" (default)" if args.net_mode == "default" else ""),
                               "what": what[default_mode],
                               "distinct_observations": 2 * tree.handle().S, "slots": T * local_batch},
            "other_modes": {name: {"env_steps_per_sec": env_steps / sec, "updates_per_sec": args.steps / sec,
                                   "ms_per_step": sec / args.steps * 1e3,
                                   "what": what[{"dense_nets": False, "forward": "forward", "tabular_nets": True}[name]]}
                            for name, sec in variants.items()},
            "rollout_env_steps_per_sec": env_steps / rollout_s,
            "rollout_ms_per_step": rollout_s / args.steps * 1e3,
            "roofline": {
                "kernel": "k_observe (K1 episode gather, rnad_observe)",
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": k1_traffic(A, args),
                "bytes_per_launch": k1_bytes_per_launch, "avg_launch_us": k1_avg_s * 1e6, "launches": n_obs,
            },
            "other_kernels": {
                "k_act": {"launches": n_act, "avg_launch_us": act_ms * 1e3 / max(n_act, 1)},
                "k_mlp_forward": {"launches": n_mlp, "total_ms_per_step": mlp_ms / args.steps},
                "k_mlp_backward": {"launches": n_bwd, "total_ms_per_step": bwd_ms / args.steps, "bound": "mfma", "achieved": bwd_tflops,
                                   "peak": 157.3, "unit": "TFLOP/s executed (fp32 MFMA, v_mfma_f32_32x32x2_f32)",
                                   "frac": bwd_tflops / 157.3 if bwd_tflops else None, "note": "dominant kernel by time"},
                "k_learn_fused": {"launches": n_learn, "avg_launch_us": learn_ms * 1e3 / max(n_learn, 1),
                                  "achieved_GBps": (learn_bytes / (learn_ms / 1e3 / max(n_learn, 1)) / 1e9) if learn_bytes and n_learn else None},
            },
            "setup": {"tree_generate_and_upload_s": setup_tree_s, "tree_table_bytes": tree.handle().table_bytes},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(tree, args, T)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def k1_traffic(A, args):
    """HBM bytes per K1 launch from the separate rocprofv3 PMC passes (profiles/r01_k1_pmc.json: FETCH_SIZE doubled per the
    gfx950 correction + WRITE_SIZE), valid for the default workload only; None otherwise."""
    path = os.path.join(ROOT, "profiles", "r01_k1_pmc.json")
    if A != 3 or args.batch_log2 != 20 or args.gpus != 1 or args.obs_half or not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)["traffic_bytes_per_launch"]


def cpu_baseline(tree, args, T):
    """The same step on the host cores: C oracle + PyTorch-CPU MLP (oracle/port.py), bounded sample."""
    from oracle.port import CpuTrainer

    arrs = dict(index=tree.index_tensor.cpu().numpy(), value=tree.value_tensor.cpu().numpy(), chance=tree.chance_tensor.cpu().numpy(),
                expected_value=tree.expected_value_tensor.cpu().numpy(), legal=tree.legal_tensor.cpu().numpy(),
                depth_bound=tree.depth_bound)
    cores = CPU_THREADS
    torch.set_num_threads(cores)
    ct = CpuTrainer(arrs, width=args.width)
    lanes = 1 << args.cpu_lanes_log2
    ct.step(min(lanes, 4096), seed=0)  # warm-up (thread pools, page faults)
    t0 = time.perf_counter()
    n, roll, upd = 0, 0.0, 0.0
    while n < 2 or (time.perf_counter() - t0 < 12 and n < 6):
        Tc, r, u = ct.step(lanes, seed=1 + n)
        roll += r
        upd += u
        n += 1
    dt = time.perf_counter() - t0
    return {
        "value": lanes * Tc * n / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
        "sample": f"{n} full steps (rollout + update) of 2^{args.cpu_lanes_log2} episodes x T={Tc} on the same tree; "
                  f"C oracle (OpenMP) + PyTorch-CPU MLP, {cores} threads",
        "rollout_env_steps_per_sec": lanes * Tc * n / roll, "updates_per_sec_at_sample_batch": n / dt,
    }


if __name__ == "__main__":
    main()
