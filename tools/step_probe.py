#!/usr/bin/env python3
"""RNaD.train_step alone (one mode, no CPU baseline, no other legs) -- the command rocprofv3 trace / counter passes run, and a
host-enqueue timer: prints GPU-inclusive ms per step and the host time spent enqueueing a step.

    python tools/step_probe.py --steps 50 [--mode tabular|forward|dense] [--batch-log2 20] [--actions 3 --transitions 1 --depth 6]
"""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "r-nad_amd"))
import torch  # noqa: E402

from environment.episode import Buffer  # noqa: E402
from environment.tree import Tree  # noqa: E402
from learn.rnad import RNaD  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--mode", choices=("tabular", "forward", "dense"), default="tabular")
    ap.add_argument("--batch-log2", type=int, default=20)
    ap.add_argument("--actions", type=int, default=3)
    ap.add_argument("--transitions", type=int, default=1)
    ap.add_argument("--depth", type=int, default=6)
    ap.add_argument("--prune", type=int, nargs=2, default=(0, 0))
    ap.add_argument("--threshold", type=float, default=None)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--freeze", action="store_true", help="restore the nets' weights before every step (ablated kernels write garbage gradients)")
    ap.add_argument("--no-graph", action="store_true", help="enqueue every step eagerly (one dispatch per kernel launch for the counter passes)")
    ap.add_argument("--sharp", type=float, default=0.0, help="scale the learner's policy output layer by this factor first (a near-deterministic actor, as after training)")
    ap.add_argument("--distinct", action="store_true", help="RNaD.distinct_trajectories = True")
    ap.add_argument("--no-dedup", action="store_true", help="RNaD.dedup_rows = False: nets and backward on all 2S rows (kernel quality of the MLP launches)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    A, C = args.actions, args.transitions
    thr = args.threshold if args.threshold is not None else (0.0 if C == 1 else 0.5 / C)
    tree = Tree(device=dev, max_actions=A, max_transitions=C, depth_bound=args.depth, transition_threshold=thr)
    tree.generate_native(seed=0, prune=tuple(args.prune))
    os.environ["RNAD_SAVE_DIR"] = tempfile.mkdtemp(prefix="rnad_probe_")
    torch.manual_seed(0)
    rn = RNaD(tree=tree, device=dev, directory_name="probe", batch_size=1 << args.batch_log2, eta=0.2, b1_adam=0.0,
              net_params={"type": "MLP", "max_actions": A, "width": args.width})
    rn.initialize()
    rn.tabular = {"tabular": True, "forward": "forward", "dense": False}[args.mode]
    rn.use_graph = not args.no_graph
    with torch.no_grad():
        for p in rn.net_reg_.parameters():
            p.mul_(1.001)
        if args.sharp:
            for name, p in rn.net.named_parameters():
                if name.startswith("policy_fc1"):
                    p.mul_(args.sharp)
    if args.distinct:
        rn.distinct_trajectories = True
    if args.no_dedup:
        rn.dedup_rows = False
    buf = Buffer(1)
    for i in range(5):
        rn.train_step(buf, 0.3)
        rn.total_steps += 1
    torch.cuda.synchronize()
    frozen = None
    if args.freeze:
        rn.fused_optimizer = False  # (its kernel would also keep the packed weight images current with the garbage)
        frozen = [(p, p.detach().clone()) for m in (rn.net, rn.net_target) for p in m.parameters()]
    host = 0.0
    t0 = time.perf_counter()
    for i in range(args.steps):
        if frozen is not None:
            with torch.no_grad():
                for p, keep in frozen:
                    p.copy_(keep)
        h0 = time.perf_counter()
        rn.train_step(buf, 0.3)
        rn.total_steps += 1
        host += time.perf_counter() - h0
        if i % 8 == 7:
            torch.cuda.synchronize()  # keep the queue short: `host` then measures enqueue cost, not back-pressure
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # host-only enqueue cost: the same steps with the GPU kept idle-waiting is not possible; report both numbers
    share = None
    bk = getattr(rn.last_episodes, "buckets", None)
    if bk is not None:
        n = int(bk.n_items.item())
        it = bk.items[:n]
        per = torch.zeros((bk.plan.n_buckets,), dtype=torch.int64, device=it.device).index_add_(0, it[:, 2].long(), it[:, 1].long())
        share = float(per.max().item()) * max(bk.plan.n_groups, 1) / rn.batch_size
    print(f"largest_bucket_share={share}")
    print(f"mode={args.mode} S={tree.handle().S} B=2^{args.batch_log2} ms_per_step={dt / args.steps * 1e3:.4f} "
          f"host_enqueue_ms_per_step={host / args.steps * 1e3:.4f}")


if __name__ == "__main__":
    main()
