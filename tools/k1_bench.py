#!/usr/bin/env python3
"""Time K1 (rnad_observe) alone on the c2 tree, B = 2^20, per env step of a real rollout (hipEvents via torch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "r-nad_amd"))
import torch
import rnad_hip
from environment.episode import Episodes
from environment.tree import Tree
from nn.net import MLP
dev = torch.device("cuda:0"); torch.manual_seed(0); B = 1 << 20
tree = Tree(device=dev, max_actions=3, max_transitions=1, depth_bound=6); tree.generate_native(seed=0)
ep = Episodes(tree, B, seed=1); ep.generate(MLP(3, 256, device=dev)); T = ep.t_eff + 1
obs = torch.empty((T, B, 2, 3, 3), device=dev); bits = torch.empty((T, B), dtype=torch.uint8, device=dev)
def run():
    for t in range(T):
        rnad_hip.observe(tree.handle(), ep.indices[t], t & 1, obs=obs[t], mask_bits=bits[t])
run(); torch.cuda.synchronize()
reps = 20
per_t = []
for t in range(T):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        rnad_hip.observe(tree.handle(), ep.indices[t], t & 1, obs=obs[t], mask_bits=bits[t])
    b.record(); torch.cuda.synchronize(); per_t.append(a.elapsed_time(b) / reps * 1e3)
print("us per launch by step:", " ".join(f"{x:.1f}" for x in per_t), " mean %.2f us -> %.0f GB/s algorithmic" % (sum(per_t)/T, 160*B/(sum(per_t)/T)/1e3))
