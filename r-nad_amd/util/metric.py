"""NashConv evaluation -- drop-in for reference util/metric.py, level-batched on the GPU.

`NashConvData(tree)` keeps the reference's attributes (`joint_policy, row_best, col_best, reach_probability, depth`)
and methods (`get_nashconv_from_net`, `get_nashconv`, `mean_nashconv_by_depth`).  The reference moves the tree to the
CPU and recurses in Python with ~20 tiny tensor ops per state (metric.py:84-175); here the whole tree is inferenced
with two `forward_policy` calls per slice and the best-response values come from rnad_nashconv's two level sweeps.
"""
from typing import Dict

import torch

import rnad_hip


class NashConvData:
    def __init__(self, tree):
        self.size = tree.value_tensor.shape[0]
        dev = tree.device
        self.joint_policy = torch.zeros((self.size, 2 * tree.max_actions), device=dev, dtype=torch.float)
        self.row_best = torch.zeros((self.size,), device=dev, dtype=torch.float)
        self.col_best = torch.zeros((self.size,), device=dev, dtype=torch.float)
        self.reach_probability = torch.zeros((self.size,), device=dev, dtype=torch.float)
        self.depth = torch.zeros((self.size,), device=dev, dtype=torch.int)

    def to(self, device):
        for key, value in self.__dict__.items():
            if torch.is_tensor(value):
                self.__dict__[key] = value.to(device)

    # ---------------------------------------------------------------- metric.py:51-90
    def get_nashconv_from_net(self, tree, net, inference_batch_size: int = 10**5) -> None:
        """Inference every state for both players, then evaluate.  Observations of all states come from K1 with
        idx = 0..S-1 (rnad_observe_all) instead of cat / negate / swapaxes per slice (metric.py:66-81)."""
        net.eval()
        A = tree.max_actions
        handle = tree.handle()
        obs_row, obs_col = rnad_hip.observe_all(handle, self.joint_policy.device)
        with torch.no_grad():
            for lo in range(0, self.size, inference_batch_size):
                hi = min(lo + inference_batch_size, self.size)
                self.joint_policy[lo:hi, :A] = net.forward_policy(obs_row[lo:hi])  # row player
                self.joint_policy[lo:hi, A:] = net.forward_policy(obs_col[lo:hi])  # column player
        self.get_nashconv(tree, self.joint_policy)
        net.train()

    # ---------------------------------------------------------------- metric.py:93-175
    def get_nashconv(self, tree, joint_policy: torch.Tensor, state_index: int = 1, reach_probablity: float = 1, depth: int = 0) -> None:
        """Best-response values of both players against `joint_policy` for the sub-tree below `state_index`.

        Like the reference, the state the call starts from uses `joint_policy[state_index]` while every descendant uses
        `self.joint_policy` (the recursion at metric.py:148-151 passes `self.joint_policy`); `get_nashconv_from_net`
        makes the two coincide.  `depth` is accepted for signature compatibility (the reference never reads it)."""
        dev = self.joint_policy.device
        root_policy = joint_policy[state_index].detach().to(device=dev, dtype=torch.float).contiguous()
        rnad_hip.nashconv(tree.handle(), self.joint_policy.contiguous(), root_policy, int(state_index), float(reach_probablity),
                          self.row_best, self.col_best, self.reach_probability, self.depth)

    # ---------------------------------------------------------------- metric.py:178-190
    def mean_nashconv_by_depth(self) -> Dict[int, float]:
        max_depth = int(self.depth[1].item())
        nashconv = self.row_best + self.col_best
        means: Dict[int, float] = {}
        for depth in range(1, max_depth + 1):
            idx = self.depth == depth
            means[depth] = torch.mean(nashconv[idx]).item()
        return means


def kld(p: torch.Tensor, q: torch.Tensor, valid: torch.Tensor, legal_actions: torch.Tensor, valid_count: int = None):
    """Masked KL(p || q) averaged over valid steps -- logging only (reference util/metric.py:193-211)."""
    if valid_count is None:
        valid_count = valid.sum().item()
    mask = (valid.unsqueeze(-1) * legal_actions).to(torch.bool)
    return torch.where(mask, p * (torch.log(p) - torch.log(q)), 0).sum().item() / valid_count
