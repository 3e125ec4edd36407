#!/usr/bin/env python3
"""What the learner's work items look like on a tree (default: configs[3]) -- lanes per item, steps shared with the bucket, live slots
above / below the cut, rows of the LDS table an item touches.  One eager step, then statistics of the batch it played."""
import argparse
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "r-nad_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import rnad_hip  # noqa: E402
from environment.episode import Buffer  # noqa: E402
from environment.tree import Tree  # noqa: E402
from learn.rnad import RNaD  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch-log2", type=int, default=20)
    ap.add_argument("--actions", type=int, default=5)
    ap.add_argument("--transitions", type=int, default=4)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--prune", type=int, nargs=2, default=(7, 8))
    ap.add_argument("--threshold", type=float, default=0.1)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    tree = Tree(device=dev, max_actions=args.actions, max_transitions=args.transitions, depth_bound=args.depth, transition_threshold=args.threshold)
    tree.generate_native(seed=0, prune=tuple(args.prune))
    os.environ["RNAD_SAVE_DIR"] = tempfile.mkdtemp(prefix="rnad_probe_")
    torch.manual_seed(0)
    B = 1 << args.batch_log2
    rn = RNaD(tree=tree, device=dev, directory_name="probe", batch_size=B, eta=0.2, b1_adam=0.0,
              net_params={"type": "MLP", "max_actions": args.actions, "width": 256})
    rn.initialize()
    rn.use_graph = False
    buf = Buffer(1)
    for _ in range(2):
        rn.train_step(buf, 0.3)
    torch.cuda.synchronize()
    ep = rn.last_episodes
    h = tree.handle()
    bk = ep.buckets
    plan = bk.plan
    n = int(bk.n_items.item())
    items = bk.items[:n].cpu().numpy().astype(np.int64)
    shared = rnad_hip.bucket_shared_steps(h, B).numpy().astype(np.int64)
    T = ep.t_eff + 1
    idx = ep.indices.cpu().numpy()[:T]  # [T, B] bucket order
    live = idx != 0
    length = live.sum(0)
    cnt = items[:, 1]
    print(f"S={h.S} rows={plan.rows} buckets={plan.n_buckets} groups={plan.n_groups} upper={plan.n_upper} items={n} T={T} lds={plan.lds}")
    print("lanes per item: mean %.1f, quantiles 10/50/90/99 %s; items with <=16/32/64/128 lanes: %s" % (
        cnt.mean(), np.percentile(cnt, [10, 50, 90, 99]).tolist(), [(cnt <= k).mean().round(3) for k in (16, 32, 64, 128)]))
    print("lanes in items with <=64 lanes: %.3f of all lanes" % (cnt[cnt <= 64].sum() / B))
    sh_item = shared[items[:, 2]]
    print("shared steps per item (lane-weighted) mean %.2f; histogram %s" % ((sh_item * cnt).sum() / B, np.bincount(sh_item, weights=cnt).astype(int).tolist()))
    term = items[:, 2] >= plan.n_groups
    print("lanes in terminal buckets: %.3f; items of terminal buckets: %d" % (cnt[term].sum() / B, term.sum()))
    # live slots above / below the cut
    col_shared = np.repeat(sh_item, cnt)
    col_begin = np.repeat(items[:, 0], cnt) + np.concatenate([np.arange(c) for c in cnt])
    sh_col = np.zeros(B, np.int64)
    sh_col[col_begin] = col_shared
    t = np.arange(T)[:, None]
    above = (live & (t < sh_col[None, :])).sum()
    below = (live & (t >= sh_col[None, :])).sum()
    print(f"live slots {live.sum()}: above the cut (phase 2) {above}, below (phase 1) {below}; mean length {length.mean():.2f}")
    # phase-1 loop iterations a wave executes: per item, per wave of 64 columns, max over lanes of (length - shared)+, against the sum of live ones
    it_exec, it_live, it_T = 0, 0, 0
    for b, c, bu, _ in items:
        ln = length[b:b + c] - shared[bu]
        ln = np.maximum(ln, 0)
        for w in range(0, c, 64):
            seg = ln[w:w + 64]
            it_exec += int(seg.max()) * 64
            it_live += int(seg.sum())
            it_T += (T - min(T, shared[bu])) * 64
    print(f"phase-1 lane-iterations: T-bounded loop {it_T}, wave-max-bounded {it_exec}, live {it_live}")
    # rows of the table that an item actually touches
    lo = None
    print("items per bucket: mean %.2f" % (n / max(len(np.unique(items[:, 2])), 1)))


if __name__ == "__main__":
    main()
