#!/usr/bin/env python3
"""How many (player, state) rows would a staged actor have to evaluate if it staged k more drawn levels below the cut?  Plays one
batch on the configs[3] tree with the default step's rollout and counts, from the played states: the rows the batch visited, the rows
of the non-empty groups (what the staged actor evaluates today), and the rows of the subtrees entered after 1 / 2 / 3 more transitions.

    python tools/stage_probe.py [--actions 5 --transitions 4 --depth 8 --prune 7 8 --threshold 0.1 --batch-log2 20]
"""
import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "r-nad_amd"))
import torch  # noqa: E402

import rnad_hip  # noqa: E402
from environment.episode import Episodes  # noqa: E402
from environment.tree import Tree  # noqa: E402
from nn.net import MLP  # noqa: E402


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
    A, C, B = args.actions, args.transitions, 1 << args.batch_log2
    tree = Tree(device=dev, max_actions=A, max_transitions=C, depth_bound=args.depth, transition_threshold=args.threshold)
    tree.generate_native(seed=0, prune=tuple(args.prune))
    h = tree.handle()
    S = h.S
    torch.manual_seed(0)
    net = MLP(A, 256, device=dev)
    ep = Episodes(tree, B, seed=1)
    ep.generate(net, bucketed=True)
    idx = ep.indices.cpu().numpy().astype(np.int64)  # [T, B]
    T = idx.shape[0]
    states = idx[::2]  # the state of every transition (both env steps of a transition share it)
    bucket_of, n_groups = rnad_hip.bucket_map(h, B)
    bucket_of = bucket_of.numpy().astype(np.int64)
    # tree structure on the host: parent, depth, subtree sizes (ids are DFS pre-order)
    index = tree.index_tensor.cpu().numpy().reshape(S, -1).astype(np.int64)
    parent = np.zeros(S, np.int64)
    src = np.repeat(np.arange(S), index.shape[1])
    flat = index.reshape(-1)
    ok = flat != 0
    parent[flat[ok]] = src[ok]
    depth = np.zeros(S, np.int64)
    order = np.arange(2, S)  # parents have smaller ids
    for s in order:
        depth[s] = depth[parent[s]] + 1
    size = np.ones(S, np.int64)
    size[0] = 0
    for s in order[::-1]:
        size[parent[s]] += size[s]
    visited = np.unique(states[states != 0])
    print(f"S = {S}, B = {B}, T = {T}; visited states {visited.size} ({2 * visited.size} rows)")
    in_group = (bucket_of >= 0) & (bucket_of < n_groups)
    # per lane: index of the first transition whose state lies in a group
    g = in_group[states] & (states != 0)
    first = np.where(g.any(0), g.argmax(0), -1)
    lanes = np.arange(B)
    have = first >= 0
    groups_hit = np.unique(bucket_of[states[first[have], lanes[have]]])
    rows_groups = int(np.isin(bucket_of, groups_hit).sum())
    n_upper = int(((bucket_of >= n_groups)).sum())
    print(f"groups {n_groups}, non-empty {groups_hit.size}; states of non-empty groups {rows_groups} ({2 * rows_groups} rows staged today, + {2 * n_upper} upper rows)")
    for k in (1, 2, 3):
        # after k more drawn transitions below the cut: the subtrees entered at that level; above them the k levels of states passed through
        passed = []
        for j in range(k):
            t = first + j
            okj = have & (t < states.shape[0])
            st = states[np.clip(t, 0, states.shape[0] - 1), lanes]
            passed.append(np.unique(st[okj & (st != 0)]))
        t = first + k
        okk = have & (t < states.shape[0])
        st = states[np.clip(t, 0, states.shape[0] - 1), lanes]
        roots = np.unique(st[okk & (st != 0)])
        n_passed = np.unique(np.concatenate(passed)).size
        # level j < k needs the policies of ALL states a lane could be in at that level given what is known: the states passed through are
        # known only after the draw at level j - 1, so each level is a stage of its own
        total = n_passed + int(size[roots].sum())
        print(f"k = {k}: stages on {[p.size for p in passed]} states, then {roots.size} subtrees with {int(size[roots].sum())} states -> {2 * total} rows in all")


if __name__ == "__main__":
    main()
