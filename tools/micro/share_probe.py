#!/usr/bin/env python3
"""How crowded do the buckets of the bench's trainer get?  Largest bucket over the even share every 512 updates, bench.py's setup
(configs[1], 2^20 lanes, lr 5e-5, alpha ramp of a 10 000-step outer iteration).   python tools/micro/share_probe.py"""
import os, sys, tempfile
sys.path.insert(0, "/root/repo/r-nad_amd")
import torch
from environment.episode import Buffer
from environment.tree import Tree
from learn.rnad import RNaD
dev = torch.device("cuda:0")
tree = Tree(device=dev, max_actions=3, max_transitions=1, depth_bound=6); tree.generate_native(seed=0)
os.environ["RNAD_SAVE_DIR"] = tempfile.mkdtemp()
torch.manual_seed(0)
rn = RNaD(tree=tree, device=dev, directory_name="share", batch_size=1 << 20, eta=0.2, b1_adam=0.0, net_params={"type": "MLP", "max_actions": 3, "width": 256})
rn.initialize()
with torch.no_grad():
    for p in rn.net_reg_.parameters(): p.mul_(1.001)
buf = Buffer(1)
for i in range(6000):
    a = rn.alpha_of(i, 10000)
    rn.alpha_ahead = lambda k, i=i: rn.alpha_of(i + k, 10000)
    rn.train_step(buf, a); rn.total_steps += 1
    if i % 512 == 511:
        bk = rn.last_episodes.buckets; n = int(bk.n_items.item()); it = bk.items[:n]
        per = torch.zeros((bk.plan.n_buckets,), dtype=torch.int64, device=dev).index_add_(0, it[:, 2].long(), it[:, 1].long())
        print(i + 1, float(per.max()) * bk.plan.n_groups / rn.batch_size, rn.__dict__.get("_distinct_crowded"), rn._distinct_now(), flush=True)
