"""Helpers for the -m gpu tests: put golden fixtures on the device behind the product's own classes."""
import numpy as np
import torch

from _util import load, load_tree, mlp_weights
from environment.episode import Episodes
from environment.tree import Tree
from nn.net import MLP
from oracle import oracle

DEV = torch.device("cuda:0")


def gpu(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV).contiguous()


def tree_from_arrays(arrs, depth_bound=1, device=DEV):
    """A product Tree whose seven tensors are the given reference-layout arrays."""
    A, C = arrs["index"].shape[-1], arrs["index"].shape[1]
    t = Tree(device=device, max_actions=A, max_transitions=C, depth_bound=depth_bound)
    t.index_tensor = torch.as_tensor(arrs["index"]).to(device)
    t.value_tensor = torch.as_tensor(arrs["value"]).to(device)
    t.chance_tensor = torch.as_tensor(arrs["chance"]).to(device)
    t.expected_value_tensor = torch.as_tensor(arrs["expected_value"]).to(device)
    t.legal_tensor = torch.as_tensor(arrs["legal"]).to(device)
    t.root_value_tensor = torch.as_tensor(arrs["root_value"]).to(device)
    t.solution_tensor = torch.as_tensor(arrs["solution"]).to(device)
    t._handle = None
    return t


def golden_tree(name):
    g = load_tree(name)
    return tree_from_arrays(g, depth_bound=g["meta"]["kw"]["depth_bound"]), g


def mlp_from(d, A, prefix="w_", device=DEV):
    w = mlp_weights(d, prefix)
    net = MLP(A, w[0].shape[0], device=device)
    sd = dict(zip(oracle.MLP_KEYS, [torch.as_tensor(x) for x in w]))
    net.load_state_dict(sd)
    return net


def mask_bits_of(masks):
    """f32 [..., A] 0/1 -> u8 [...] bit i = legal i."""
    A = masks.shape[-1]
    return (masks.astype(np.int64) * (1 << np.arange(A))).sum(-1).astype(np.uint8)


def episodes_from_golden(tree, ro):
    """An Episodes object holding the reference's recorded trajectory (compact primaries on the GPU)."""
    T = int(ro["t_eff"]) + 1
    B = ro["indices"].shape[1]
    ep = Episodes(tree, B, seed=0)
    ep.t_eff = T - 1
    ep.indices = gpu(ro["indices"], torch.int32)
    ep.observations = gpu(ro["observations"])
    ep.mask_bits = gpu(mask_bits_of(ro["masks"]))
    ep.policy = gpu(ro["policy"])
    ep.action_idx = gpu(ro["actions"].argmax(-1), torch.int32)
    ep.rewards = gpu(ro["rewards"])
    ep.values = gpu(ro["values"])
    alive = np.zeros(T + 1, np.int32)
    alive[:T] = (ro["indices"] != 0).sum(1)
    ep.alive = gpu(alive)
    ep.finished = True
    return ep


class ReplayNet(torch.nn.Module):
    """A net honouring the reference contract (nn/net.py:37-51) that replays recorded outputs step by step."""

    def __init__(self, logits, policy, values, actions=None, fast=False):
        super().__init__()
        self.logits, self.policy, self.values, self.acts = logits, policy, values, actions
        self.t = 0
        self.device = DEV
        if fast:
            self.forward_logits = self._forward_logits

    def _forward_logits(self, obs):
        t = min(self.t, self.logits.shape[0] - 1)
        self.t += 1
        return self.logits[t], self.values[t].view(-1, 1)

    def forward(self, obs):
        t = min(self.t, self.logits.shape[0] - 1)
        self.t += 1
        return self.logits[t], self.policy[t], self.values[t].view(-1, 1), self.acts[t]


def cpu(t):
    return t.detach().cpu().numpy()
