"""rnad_hip.leaf_columns / leaf_items on the CPU (no GPU needed): the columns built from a tree's tensors are the trajectories episodes
take -- checked against the C oracle's rollouts (oracle_rollout: reference environment/episode.py:175-230) on the four fixture trees of
the reference: every lane's states, actions and reward are those of the column of the transition it left the tree by."""
import numpy as np
import pytest
import torch

from _util import TREES, load_tree
from oracle import oracle


def _max_depth(index):
    """Longest chain of transitions from state 1 (what rnad_tree_info(tree, 3) reports)."""
    S = index.shape[0]
    depth = np.full(S, -1)
    depth[1] = 0
    frontier = [1]
    while frontier:
        nxt = []
        for s in frontier:
            for child in np.unique(index[s]):
                if child != 0 and depth[child] < 0:
                    depth[child] = depth[s] + 1
                    nxt.append(int(child))
        frontier = nxt
    return int(depth.max()) + 1, depth


@pytest.mark.parametrize("name", TREES)
def test_columns_are_the_trajectories_of_the_oracles_rollouts(name):
    import rnad_hip

    g = load_tree(name)
    index, chance, value = g["index"], g["chance"], g["value"]
    S, C, A, _ = index.shape
    max_depth, depth = _max_depth(index)
    # buckets: by the depth-1 ancestor (any map that is constant on subtrees and -1 on unreachable states will do for the builder)
    bucket_of = np.where(depth >= 0, 0, -1)
    cols = rnad_hip.leaf_columns(torch.as_tensor(index), torch.as_tensor(chance), torch.as_tensor(value), torch.as_tensor(bucket_of), max_depth)
    n = cols["n_cols"]
    reach = depth >= 0
    reach[0] = False
    want = int(((index == 0) & (chance > 0))[reach].sum())
    assert n == want, "one column per terminal transition of a reachable state"
    col_of = cols["col_of"].numpy()
    assert (np.sort(col_of[col_of >= 0]) == np.arange(n)).all()
    # episodes of the oracle (uniform-ish random nets): the lane's last live transition -> its column
    rng = np.random.default_rng(0)
    W = 16
    weights = [rng.normal(size=s).astype(np.float32) * 0.3 for s in ((W, 2 * A * A), (W,), (1, W), (1,), (W, 2 * A * A), (W,), (A, W), (A,))]
    B, T_cap = 512, 2 * max_depth
    tree = dict(index=index, value=value, chance=chance, expected_value=g["expected_value"], legal=g["legal"])
    ro = oracle.rollout(tree, weights, B, T_cap, seed=3)
    T = ro["T"]
    idx = np.zeros((T_cap + 1, B), np.int64)
    idx[:T] = ro["indices"]
    live = idx[:T_cap] != 0
    assert live[0].all() and not idx[T_cap].any()
    t_last = live.sum(0) - 1
    lanes = np.arange(B)
    s_last = idx[t_last, lanes]
    a0, a1 = ro["actions"][t_last - 1, lanes], ro["actions"][t_last, lanes]
    rew = ro["rewards"][t_last, lanes]
    term = (index[s_last][lanes, :, a0, a1] == 0) & (chance[s_last][lanes, :, a0, a1] > 0)
    val = value[s_last][lanes, :, a0, a1]
    match = term & (val.view(np.uint32) == rew.view(np.uint32)[:, None])
    assert match.any(1).all()
    c = match.argmax(1)
    col = col_of[((s_last * A + a0) * A + a1) * C + c]
    assert (col >= 0).all()
    np.testing.assert_array_equal(cols["indices"].numpy()[:, col], idx)
    acts = np.zeros(B, np.int64)
    for t in range(T):
        acts |= np.where(live[t], ro["actions"][t], 0).astype(np.int64) << (3 * t)
    np.testing.assert_array_equal(cols["acts"].numpy()[col], acts)
    assert (cols["final_reward"].numpy()[col].view(np.uint32) == rew.view(np.uint32)).all()


def test_items_tile_the_columns_in_equal_shares():
    import rnad_hip

    bucket = torch.tensor([0] * 729 + [1] * 10 + [3] * 257)
    items = rnad_hip.leaf_items(bucket, 256)
    assert items[:3] == [(0, 243, 0, 0), (243, 243, 0, 0), (486, 243, 0, 0)]
    assert items[3] == (729, 10, 1, 1)
    assert items[4:] == [(739, 129, 3, 0), (868, 128, 3, 0)]
    assert sum(i[1] for i in items) == bucket.numel() and max(i[1] for i in items) <= 256
