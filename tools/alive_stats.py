#!/usr/bin/env python3
"""Alive-lane fraction per rollout step on the bench tree (how much of [T, B] is padding)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "r-nad_amd"))
import torch  # noqa: E402

from environment import episode  # noqa: E402
from environment.tree import Tree  # noqa: E402
from nn.net import MLP  # noqa: E402

dev = torch.device("cuda:0")
A, C, depth, B = 3, 1, 6, 2**20
tree = Tree(device=dev, max_actions=A, max_transitions=C, depth_bound=depth, transition_threshold=0.0)
tree.generate_native(seed=0)
net = MLP(A, 256, device=dev)
ep = episode.Episodes(tree, B, seed=1)
ep.generate(net, trim=False)
print("max_depth", tree.handle().max_depth, "alive", [int(a) / B for a in ep.alive.cpu()])
print("indices != 0:", [(int((ep.indices[t] != 0).sum())) for t in range(ep.indices.shape[0])])
print("indices[:, :4]", ep.indices[:, :4].cpu().tolist())
print("traj idx shape", ep._traj.indices.shape)
