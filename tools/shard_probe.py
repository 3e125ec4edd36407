#!/usr/bin/env python3
"""What a rank's row work costs when the 2S rows of the tree are sharded over N ranks (RNaD.shard_rows): the fused table launch
(rnad_mlp_rows_records) and the backward (+ its reduction) on a contiguous 1/N of the rows, timed on ONE GPU with hipEvents, next to the
bytes the three extra collectives move.  Input of the strong-scaling prediction in DESIGN.md section 7.

    python tools/shard_probe.py [--actions 3 --transitions 1 --depth 6 --width 256]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "r-nad_amd"))
import torch  # noqa: E402

import rnad_hip  # noqa: E402
from environment.tree import Tree  # noqa: E402
from nn.net import MLP  # noqa: E402


def timed(fn, reps=50):
    """us per call, replayed from a hipGraph of `reps` calls (an eagerly enqueued call is host-paced at these sizes)."""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4):
        graph.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (4 * reps) * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--actions", type=int, default=3)
    ap.add_argument("--transitions", type=int, default=1)
    ap.add_argument("--depth", type=int, default=6)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--batch-log2", type=int, default=22)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    A, C, W = args.actions, args.transitions, args.width
    tree = Tree(device=dev, max_actions=A, max_transitions=C, depth_bound=args.depth, transition_threshold=0.0 if C == 1 else 0.5 / C)
    tree.generate_native(seed=0)
    h = tree.handle()
    N = 2 * h.S
    torch.manual_seed(0)
    nets = [MLP(A, W, device=dev) for _ in range(4)]
    fold = h.legal_foldable
    table = h.observations_table()
    packs = rnad_hip.mlp_pack_many([n._weights() for n in nets], A, fold=fold)
    regs = rnad_hip.mlp_forward_multi(packs[2:], W, table, A, [(True, False), (True, False)], fold=h if fold else False)
    hp = rnad_hip.make_learn_params(alpha=0.3, eta=0.2)
    dl = torch.randn((N, A), device=dev) * 1e-3
    dv = torch.randn((N, 1), device=dev) * 1e-3
    weights = nets[0]._weights()
    out = {"rows": N, "A": A, "width": W, "ranks": {}}
    plan = rnad_hip.bucket_plan(h, (1 << args.batch_log2) // 8)
    A1 = A + 1
    acc_bytes = 8 * (N * A1 + rnad_hip.BUCKET_REPLICAS * 2 * max(plan.n_upper, 1) * A1) if plan is not None else None
    strides = [int(rnad_hip.lib().rnad_bucket_policy_row_stride(A)), int(rnad_hip.lib().rnad_bucket_fast_record_stride(A)),
               int(rnad_hip.lib().rnad_bucket_record_stride(A))]
    for n in (1, 2, 4, 8):
        per = (N + n - 1) // n
        rows = None if n == 1 else rnad_hip.RowList(torch.arange(0, per, dtype=torch.int32), N, dev)
        fwd = timed(lambda: rnad_hip.mlp_rows_records(h, packs[0], packs[1], W, table, regs[0][0], regs[1][0], hp, fold=h if fold else False,
                                                      rows=rows, alloc_rows=per * n if n > 1 else None))
        bwd = timed(lambda: rnad_hip.mlp_backward(packs[0], weights, table, A, dl, dv, live=rows, fold=h if fold else False))
        out["ranks"][n] = {"rows_per_rank": per, "forward_records_us": round(fwd, 1), "backward_reduce_us": round(bwd, 1),
                           "all_gather_bytes_received": None if n == 1 else [4 * s * per * (n - 1) for s in strides],
                           "accumulator_all_reduce_bytes": None if n == 1 else acc_bytes}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
