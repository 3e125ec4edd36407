"""Replay buffer path (reference environment/episode.py:243-333): sample / collate / Buffer on host tensors (no kernels involved)."""
import random

import numpy as np
import torch

from environment.episode import Buffer, Episodes
from environment.tree import Tree


def _fake(tree, B, T, seed):
    g = torch.Generator().manual_seed(seed)
    ep = Episodes.__new__(Episodes)
    ep.tree, ep.batch_size, ep.t_eff, ep._lazy = tree, B, T - 1, {}
    ep.seed, ep.lane_offset, ep.obs_half, ep.finished, ep.actor_logits = 0, 0, False, True, None
    A = tree.max_actions
    ep.indices = torch.randint(0, 5, (T, B), generator=g, dtype=torch.int32)
    ep.observations = torch.randn((T, B, 2, A, A), generator=g)
    ep.mask_bits = torch.randint(1, 8, (T, B), generator=g, dtype=torch.uint8)
    ep.policy = torch.rand((T, B, A), generator=g)
    ep.action_idx = torch.randint(0, A, (T, B), generator=g, dtype=torch.int32)
    ep.rewards = torch.randn((T, B), generator=g)
    ep.values = torch.randn((T, B), generator=g)
    alive = torch.zeros(T + 1, dtype=torch.int32)
    alive[:T] = (ep.indices != 0).sum(1).to(torch.int32)
    ep.alive = alive
    return ep


def test_sample_subset_and_full_batch():
    np.random.seed(0)
    tree = Tree(max_actions=3, depth_bound=2)
    ep = _fake(tree, 10, 4, seed=1)
    assert ep.sample(10) is ep and ep.sample(99) is ep  # whole batch: no copy (a lane permutation changes no loss term)
    random.seed(5)
    full = ep.sample(10, shuffle=True)  # the reference's behaviour: a random permutation of the lanes
    perm = [int(np.flatnonzero((ep.rewards[0] == r).numpy())[0]) for r in full.rewards[0]]
    assert sorted(perm) == list(range(10)) and perm != list(range(10))
    for key in Episodes._PRIMARY:
        assert torch.equal(getattr(full, key), getattr(ep, key)[:, perm])
    sub = ep.sample(4)
    assert sub.batch_size == 4 and sub.t_eff == ep.t_eff and sub.finished and sub.indices.shape == (4, 4)
    cols = [int(np.flatnonzero((ep.rewards[0] == r).numpy())[0]) for r in sub.rewards[0]]
    assert len(set(cols)) == 4
    assert torch.equal(sub.alive[:4], (sub.indices != 0).sum(1).to(torch.int32))
    np.testing.assert_allclose(sub.valid_counts.numpy(), [(sub.indices[0::2] != 0).sum(), (sub.indices[1::2] != 0).sum()])


def test_collate_pads_time_with_invalid_steps_and_concatenates_lanes():
    np.random.seed(0)
    tree = Tree(max_actions=3, depth_bound=2)
    a, b = _fake(tree, 3, 6, seed=2), _fake(tree, 5, 4, seed=3)
    c = Episodes.collate([a, b])
    assert c.batch_size == 8 and c.t_eff == 5 and c.finished
    for key in Episodes._PRIMARY:
        x = getattr(c, key)
        assert x.shape[:2] == (6, 8)
        assert torch.equal(x[:, :3], getattr(a, key)) and torch.equal(x[:4, 3:], getattr(b, key))
        assert (x[4:, 3:] == 0).all()  # zero padding == invalid steps (index 0), episode.py:262-263
    assert torch.equal(c.alive[:6], (c.indices != 0).sum(1).to(torch.int32))
    assert Episodes.collate([a]) is a


def test_buffer_sample_draws_multinomial_bucket_sizes():
    np.random.seed(3)
    random.seed(3)
    tree = Tree(max_actions=3, depth_bound=2)
    buf = Buffer(2)
    for s in range(3):
        buf.append(_fake(tree, 6, 4, seed=10 + s))
    assert len(buf.episodes_buffer) == 2  # oldest batch evicted
    out = buf.sample(6)
    assert out.batch_size == 6 and out.indices.shape == (4, 6)
    buf.clear()
    assert len(buf.episodes_buffer) == 0
