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


def bucket_keys(tree, B, idx, last):
    """Bucket of every lane from its recorded states (host restatement of k_bucket_keys): the first state on its path that lies in
    a group, or -- if it leaves the tree before reaching one -- the terminal bucket of its last (upper) state.
    idx [T, B] states at the start of each step, last [B] the states after the last step."""
    import rnad_hip

    bucket_of, n_groups = rnad_hip.bucket_map(tree.handle(), B)
    bucket_of = bucket_of.numpy().astype(np.int64)
    states = np.concatenate([idx, last[None]], 0)
    key = bucket_of[states[0]]
    done = key < n_groups
    for t in range(1, states.shape[0]):
        st = states[t]
        upd = (~done) & (st != 0)
        key = np.where(upd, bucket_of[st], key)
        done |= key < n_groups
    return key, n_groups


def bucket_order(tree, B, idx):
    """(perm, items): the stable sort of the lanes by bucket and the learner's work list (<= 256 lanes per item), as the sort
    passes of csrc/bucket.hip build them -- for batches that were NOT played by rnad_rollout_bucketed (the reference's recordings)."""
    key, _ = bucket_keys(tree, B, idx, np.zeros(B, idx.dtype))
    perm = np.argsort(key, kind="stable")
    items = []
    sk = key[perm]
    start = 0
    while start < B:
        n = int((sk == sk[start]).sum())
        chunks = (n + 255) // 256
        for c in range(chunks):
            items.append((start + c * 256, min(256, n - c * 256), int(sk[start]), int(chunks == 1)))
        start += n
    return perm, items


def buckets_of(tree, B, perm, items, norm=None):
    import rnad_hip

    plan = rnad_hip.bucket_plan(tree.handle(), B)
    assert plan is not None
    b = rnad_hip.Buckets(plan, DEV)
    b.lane_ids.copy_(gpu(perm, torch.int32))
    b.items[: len(items)] = torch.as_tensor(items, dtype=torch.int32, device=DEV)
    b.n_items.fill_(len(items))
    if norm is not None:
        b.norm.copy_(gpu(np.asarray(norm, np.float64)))
    return b


def compact_episodes_from_recorded(tree, indices, actions, rewards):
    """An Episodes object holding a RECORDED trajectory (indices [T, B], action ids [T, B], rewards [T, B], all lane-ordered, e.g. the
    reference's) the way rnad_rollout_bucketed_compact would have left it: lanes stably sorted by bucket, 64 bytes per lane --
    states, the actions packed 3 bits per step, the episode's one reward (rewards *= (indices == 0), episode.py:120-121) --
    plus the learner's work list.  The records table (the actor) is attached by the caller: ep._compact = (traj, records)."""
    import rnad_hip

    T, B = indices.shape
    assert T <= rnad_hip.COMPACT_MAX_STEPS
    idx = np.asarray(indices).astype(np.int64)
    perm, items = bucket_order(tree, B, idx)
    idx, act, rew = idx[:, perm], np.asarray(actions).astype(np.int64)[:, perm], np.asarray(rewards, np.float32)[:, perm]
    live = idx != 0
    acts = np.zeros(B, np.uint64)
    for t in range(T):
        acts |= (np.where(live[t], act[t], 0).astype(np.uint64) << np.uint64(3 * t))
    # the reward of the step that leaves the tree: the last live step of the lane (a column step); every other reward is zero
    t_last = live.shape[0] - 1 - np.argmax(live[::-1], axis=0)
    final = np.where(live.any(0), rew[t_last, np.arange(B)], np.float32(0)).astype(np.float32)
    others = rew.copy()
    others[t_last, np.arange(B)] = 0
    assert (others == 0).all(), "more than one non-zero reward in an episode"
    assert (t_last[live.any(0)] % 2 == 1).all(), "episodes end on a column step"
    traj = rnad_hip.Trajectory(tree.handle(), B, T, DEV, compact=True)
    traj.acts.copy_(gpu(acts.view(np.int64)))
    traj.final_reward.copy_(gpu(final))
    alive = np.zeros(T + 1, np.int32)
    alive[:T] = live.sum(1)
    traj.alive.copy_(gpu(alive))
    ep = Episodes(tree, B, seed=0)
    ep.t_eff, ep.finished = T - 1, True
    ep._traj = traj
    ep.alive = traj.alive
    for name in ep._DENSE:
        setattr(ep, name, None)
    norm = [alive[0:T:2].sum(), alive[1:T:2].sum()]
    ep.buckets = buckets_of(tree, B, perm, items, norm)
    ep.lane_ids = ep.buckets.lane_ids
    # the states: one byte (two on wide cuts) per slot below the cut, nothing above it (rnad_bucket_pack_states checks that every column
    # is a lane of its work item's bucket); Episodes.indices rebuilds the recorded tensor from them
    full = gpu(np.concatenate([idx, np.zeros((1, B), np.int64)], 0), torch.int32)
    rnad_hip.bucket_pack_states(tree.handle(), ep.buckets, traj, full)
    ep._compact = (traj, None)
    assert torch.equal(ep.indices, full[:T]), "pack -> indices must be the identity"
    return ep, perm
