#!/usr/bin/env python3
"""Rollouts alone (Episodes.generate as RNaD.train_step calls it) on a tree of the given shape -- the command the rocprofv3
counter passes of the rollout kernels run (tools/pmc_run.sh).

    python tools/rollout_probe.py --actions 5 --transitions 4 --depth 8 --prune 7 8 --threshold 0.1 --reps 3
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "r-nad_amd"))
import torch  # noqa: E402

from environment.episode import Episodes  # noqa: E402
from environment.tree import Tree  # noqa: E402
from nn.net import MLP  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--actions", type=int, default=3)
    ap.add_argument("--transitions", type=int, default=1)
    ap.add_argument("--depth", type=int, default=6)
    ap.add_argument("--prune", type=int, nargs=2, default=(0, 0))
    ap.add_argument("--threshold", type=float, default=None)
    ap.add_argument("--batch-log2", type=int, default=20)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--dense", action="store_true", help="evaluate the actor on every lane at every step (tabular=False)")
    ap.add_argument("--obs-half", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    A, C = args.actions, args.transitions
    thr = args.threshold if args.threshold is not None else (0.0 if C == 1 else 0.5 / C)
    tree = Tree(device=dev, max_actions=A, max_transitions=C, depth_bound=args.depth, transition_threshold=thr)
    tree.generate_native(seed=0, prune=tuple(args.prune))
    h = tree.handle()
    net = MLP(A, args.width, device=dev)
    B = 1 << args.batch_log2
    tabular = (not args.dense) and 8 * h.S <= 2 * h.max_depth * B
    for i in range(2):
        Episodes(tree, B, seed=i, obs_half=args.obs_half).generate(net, trim=False, skip_absorbed=True, store_values=False, tabular=tabular)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.reps):
        ep = Episodes(tree, B, seed=10 + i, obs_half=args.obs_half)
        ep.generate(net, trim=False, skip_absorbed=True, store_values=False, tabular=tabular)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    alive = ep.alive.cpu().numpy()
    print(f"S={h.S} A={A} C={C} depth={h.max_depth} table_bytes={h.table_bytes} B=2^{args.batch_log2} tabular={tabular} "
          f"rollout_ms={dt * 1e3:.3f} alive={alive.tolist()}")


if __name__ == "__main__":
    main()
