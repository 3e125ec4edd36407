#!/usr/bin/env python3
"""Long run of graph-replayed training steps: the parameters must stay finite and the run must keep pace (a captured-graph problem
that only shows after thousands of replays -- like the memset node this package no longer uses -- shows up here).

    python tools/soak.py --steps 50000 [--actions 3 --transitions 1 --depth 6 --batch-log2 20] [--lazy]
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
    ap.add_argument("--steps", type=int, default=50000)
    ap.add_argument("--batch-log2", type=int, default=20)
    ap.add_argument("--actions", type=int, default=3)
    ap.add_argument("--transitions", type=int, default=1)
    ap.add_argument("--depth", type=int, default=6)
    ap.add_argument("--prune", type=int, nargs=2, default=(0, 0))
    ap.add_argument("--lazy", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    A, C = args.actions, args.transitions
    tree = Tree(device=dev, max_actions=A, max_transitions=C, depth_bound=args.depth, transition_threshold=0.0 if C == 1 else 0.5 / C)
    tree.generate_native(seed=0, prune=tuple(args.prune))
    os.environ["RNAD_SAVE_DIR"] = tempfile.mkdtemp(prefix="rnad_soak_")
    torch.manual_seed(0)
    rn = RNaD(tree=tree, device=dev, directory_name="soak", batch_size=1 << args.batch_log2, eta=0.2, b1_adam=0.0, lr=1e-4,
              net_params={"type": "MLP", "max_actions": A, "width": 256})
    rn.initialize()
    if args.lazy:
        rn.lazy_rows = True
    buf = Buffer(1)
    delta_m = 2000
    t0 = time.perf_counter()
    for i in range(args.steps):
        n = i % delta_m
        if n == 0 and i:  # the rotation of rnad.py:528-531
            rn.net_reg_.load_state_dict(rn.net_reg.state_dict())
            rn.net_reg.load_state_dict(rn.net_target.state_dict())
        rn.train_step(buf, 1 if n > delta_m / 2 else n * 2 / delta_m)
        rn.total_steps += 1
        if i % 5000 == 4999:
            torch.cuda.synchronize()
            ok = all(bool(torch.isfinite(p).all()) for net in (rn.net, rn.net_target) for p in net.parameters())
            print(f"step {i + 1}: finite={ok} {1e3 * (time.perf_counter() - t0) / (i + 1):.4f} ms/step "
                  f"graph={rn._graph is not None and rn._graph.get('graph') is not None} "
                  f"leaf={getattr(rn.last_episodes.buckets.plan, 'leaf', None) is not None } "
                  f"largest_bucket_share={rn.__dict__.get('_leaf_share')}", flush=True)
            if not ok:
                raise SystemExit("parameters are not finite")
    print("soak ok")


if __name__ == "__main__":
    main()
