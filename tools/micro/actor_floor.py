#!/usr/bin/env python3
"""What a launch of the staged actor (rnad_mlp_forward_actor on a row list) costs as a function of the number of rows, configs[3] shape
(A = 5, width 256, legal fold):  python tools/micro/actor_floor.py     (RNAD_HIP_SO selects a library variant)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "r-nad_amd"))
import torch  # noqa: E402

import rnad_hip  # noqa: E402
from environment.tree import Tree  # noqa: E402
from nn.net import MLP  # noqa: E402

dev = torch.device("cuda:0")
tree = Tree(device=dev, max_actions=5, max_transitions=4, depth_bound=7, transition_threshold=0.1)
tree.generate_native(seed=0, prune=(7, 8))
h = tree.handle()
torch.manual_seed(0)
net = MLP(5, 256, device=dev)
fold = h.legal_foldable
packed = rnad_hip.mlp_pack_many([net._weights()], 5, fold=fold)[0]
table = h.observations_table()
N = table.shape[0]
logit = torch.empty((N, 5), device=dev)
pol = torch.empty((N, int(rnad_hip.lib().rnad_bucket_policy_row_stride(5))), device=dev)
print("rows in the table", N, "fold", fold)
for n in (64, 1024, 5000, 20000, 60000, min(N, 100000), min(N, 180000)):
    rows = rnad_hip.RowList(torch.randperm(N)[:n].sort().values.to(torch.int32), N, dev)
    for _ in range(5):
        rnad_hip.mlp_forward_actor(h, packed, 256, table, logit, pol, rows=rows, fold=h if fold else False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            rnad_hip.mlp_forward_actor(h, packed, 256, table, logit, pol, rows=rows, fold=h if fold else False)
    g.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"{n:7d} rows: {e0.elapsed_time(e1) * 1e3 / 100:7.2f} us per launch")
