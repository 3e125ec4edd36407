#!/usr/bin/env python3
"""How much of the compact rollout (k_bucket_rollout_items) hides behind the learner (k_bucket_learn_c + k_bucket_finish) when both are
resident on the chip at once?  The rollout of one batch on a second stream beside the learner of another batch, against the same two
calls back to back on one stream: the upper bound of what a fused rollout + learner launch could gain.

    python tools/micro/overlap_probe.py [--reps 200]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "r-nad_amd"))
import torch  # noqa: E402

import rnad_hip  # noqa: E402
from environment.tree import Tree  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--batch-log2", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    A = 3
    tree = Tree(device=dev, max_actions=A, max_transitions=1, depth_bound=6, transition_threshold=0.0)
    tree.generate_native(seed=0)
    h = tree.handle()
    B, T = 1 << args.batch_log2, 2 * h.max_depth
    hp = rnad_hip.make_learn_params(0.3, 0.2)
    tabs = [torch.randn((2 * h.S, A), device=dev) for _ in range(3)]
    v, vt = torch.randn((2 * h.S, 1), device=dev), torch.randn((2 * h.S, 1), device=dev)
    records, fast = rnad_hip.bucket_records(h, tabs[0], v, vt, tabs[1], tabs[2], hp, fast=True)
    trajs = [rnad_hip.Trajectory(h, B, T, dev, with_observations=False, with_values=False, compact=True) for _ in range(2)]
    bks = [rnad_hip.rollout_bucketed_compact(h, tr, records, seed=7 + i) for i, tr in enumerate(trajs)]
    torch.cuda.synchronize()

    def play(i):
        rnad_hip.bucket_play(h, trajs[i], bks[i], records._policy_rows, seed=7 + i, table_is_policy=True, column=0)

    def learn(i):
        rnad_hip.learn_bucketed_compact(h, bks[i], trajs[i], T, records, fast, bks[i].norm, hp)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.reps * 1e6

    def only_play():
        for _ in range(args.reps):
            play(1)

    def only_learn():
        for _ in range(args.reps):
            learn(0)

    def back_to_back():
        for _ in range(args.reps):
            play(1)
            learn(0)

    full = rnad_hip.Trajectory(h, B, T, dev, with_observations=False, with_values=False, compact=True)

    def two_calls():  # the production sequence: keys, scan, scatter, rollout | learner (adds the alive counts up), finish
        for _ in range(args.reps):
            bk = rnad_hip.rollout_bucketed_compact(h, full, records, seed=3, defer_alive=True)
            rnad_hip.learn_bucketed_compact(h, bk, full, T, records, fast, bk.norm, hp)

    def one_call():  # keys, scan, scatter, rollout + learner, alive, finish
        for _ in range(args.reps):
            rnad_hip.rollout_learn_bucketed_compact(h, full, records, fast, hp, seed=3)

    side = torch.cuda.Stream()

    def side_by_side():
        side.wait_stream(torch.cuda.current_stream())
        for _ in range(args.reps):
            with torch.cuda.stream(side):
                play(1)
            learn(0)
        torch.cuda.current_stream().wait_stream(side)

    # graphs keep the host out of the picture
    def graphed(fn):
        g = torch.cuda.CUDAGraph()
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            fn()
        return g.replay

    res = {}
    for name, fn in (("rollout", only_play), ("learner+finish", only_learn), ("back_to_back", back_to_back), ("side_by_side", side_by_side),
                     ("two_calls", two_calls), ("one_call", one_call)):
        try:
            res[name] = timed(graphed(fn))
            res[name + " (eager)"] = timed(fn)
        except Exception as e:  # noqa: BLE001
            res[name] = f"failed: {e}"
    print({k: (round(x, 2) if isinstance(x, float) else x) for k, x in res.items()}, "us per repetition")


if __name__ == "__main__":
    main()
