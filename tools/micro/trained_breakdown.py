#!/usr/bin/env python3
"""Where the step's time goes once the policy has sharpened: N graph-replayed updates on configs[1], then 100 eagerly enqueued steps with
every bracketed kernel timed by hipEvents (rnad_hip.prof_*), with and without the learner on distinct trajectories.

    python tools/micro/trained_breakdown.py [--updates 20000]
"""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "r-nad_amd"))
import torch  # noqa: E402

import rnad_hip  # noqa: E402
from environment.episode import Buffer  # noqa: E402
from environment.tree import Tree  # noqa: E402
from learn.rnad import RNaD  # noqa: E402

NAMES = {rnad_hip.PROF_MLP: "table forwards + records", rnad_hip.PROF_BUCKET_KEYS: "keys (+ copies)", rnad_hip.PROF_BUCKET_SORT: "scan + scatter",
         rnad_hip.PROF_BUCKET_ROLLOUT: "rollout", rnad_hip.PROF_BUCKET_LEARN: "rollout + learner (one launch) / learner",
         rnad_hip.PROF_BUCKET_FINISH: "finish", rnad_hip.PROF_MLP_BWD: "backward + reduction"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--updates", type=int, default=20000)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    tree = Tree(device=dev, max_actions=3, max_transitions=1, depth_bound=6, transition_threshold=0.0)
    tree.generate_native(seed=0)
    os.environ["RNAD_SAVE_DIR"] = tempfile.mkdtemp(prefix="rnad_tb_")
    torch.manual_seed(0)
    rn = RNaD(tree=tree, device=dev, directory_name="tb", batch_size=1 << 20, eta=0.2, b1_adam=0.0,
              net_params={"type": "MLP", "max_actions": 3, "width": 256})
    rn.initialize()
    buf = Buffer(1)
    for i in range(args.updates):
        rn.train_step(buf, 0.3)
        rn.total_steps += 1
    torch.cuda.synchronize()
    for distinct in (True, False):
        rn.distinct_trajectories = distinct
        rn.use_graph = True
        for _ in range(8):
            rn.train_step(buf, 0.3)
            rn.total_steps += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(500):
            rn.train_step(buf, 0.3)
            rn.total_steps += 1
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 500 * 1e3
        rn.use_graph = False
        for _ in range(3):
            rn.train_step(buf, 0.3)
        torch.cuda.synchronize()
        rnad_hip.prof_enable(True)
        for _ in range(100):
            rn.train_step(buf, 0.3)
        torch.cuda.synchronize()
        parts = {}
        for k, nm in NAMES.items():
            n, t = rnad_hip.prof_read(k)
            if n:
                parts[nm] = round(t * 1e3 / 100, 1)
        rnad_hip.prof_enable(False)
        print(f"after {args.updates} updates, learner on distinct trajectories = {distinct}: {ms:.4f} ms per replayed step; us per eager step: {parts}", flush=True)


if __name__ == "__main__":
    main()
