"""The learner on the tree's leaf paths (include/rnad_hip.h rnad_leaf_paths_t, rnad_hip.LeafPaths, RNaD.leaf_paths): the transition a lane
leaves the tree by fixes its whole trajectory (reference tree.py:311-330: DFS pre-order ids, one parent entry per state), so the on-policy
update of learn/rnad.py:365-425 is  sum over the terminal transitions of  (lanes that took it) x (that trajectory's addends)  -- 64-bit
integer sums: the per-lane learner's bits."""
import os
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")


def _setup(name, B, seed=6):
    import rnad_hip
    from test_hip_bucket import TREES, _four_nets, _native_tree, _tables

    tree = _native_tree(**TREES[name])
    h = tree.handle()
    A = tree.max_actions
    nets = _four_nets(A, 64, seed=seed)
    logit, v, vt, lr, lr_ = _tables(tree, nets, A)
    hp = rnad_hip.make_learn_params(alpha=0.3, eta=0.2, w_v=0.7, w_n=1.3)
    rec, fast = rnad_hip.bucket_records(h, logit, v, vt, lr, lr_, hp, fast=True)
    return tree, h, rec, fast, hp


@pytest.mark.parametrize("name", ("ternary4", "binary", "a5c4", "pruned"))
def test_columns_are_the_trajectories_lanes_play(name):
    """Every lane of a played batch against the column of the transition it left the tree by: the same states, actions and reward."""
    import rnad_hip

    B = 8192
    tree, h, rec, fast, hp = _setup(name, B)
    leaf = rnad_hip.LeafPaths(h, B, tree.index_tensor, tree.chance_tensor, tree.value_tensor)
    T = 2 * h.max_depth
    traj = rnad_hip.Trajectory(h, B, T, DEV, with_observations=False, with_values=False, compact=True)
    rnad_hip.rollout_bucketed_compact(h, traj, rec, seed=5)
    idx = traj.indices.long()  # [T + 1, B]
    acts = traj.acts
    A, Cc = h.A, h.C
    alive = idx[:T] != 0
    assert bool((idx[T] == 0).all()), "every lane has left the tree by the end of the window"
    t_last = (alive.long().sum(0) - 1)  # the column step of the last live state
    cols = torch.arange(B, device=DEV)
    s_last = idx[t_last, cols]
    a0 = (acts >> (3 * (t_last - 1))) & 7
    a1 = (acts >> (3 * t_last)) & 7
    # the outcome drawn: the one terminal transition of (s, a0, a1) with that reward -- or the only one
    index = tree.index_tensor.to(DEV).long()
    term = (index[s_last, :, a0, a1] == 0) & (tree.chance_tensor.to(DEV)[s_last, :, a0, a1] > 0)  # [B, C]
    val = tree.value_tensor.to(DEV)[s_last, :, a0, a1]
    match = term & (val.view(torch.int32) == traj.final_reward.view(torch.int32).unsqueeze(1))
    assert bool(match.any(1).all())
    c = match.float().argmax(1)
    col = leaf.col_of[((s_last * A + a0) * A + a1) * Cc + c].long()
    assert bool((col >= 0).all())
    assert torch.equal(leaf.indices.long()[:, col], idx), "the column's states are the lane's"
    live_bits = (1 << (3 * (t_last + 1))) - 1
    assert torch.equal(leaf.acts[col] & live_bits, acts & live_bits), "and its actions"
    assert torch.equal(leaf.final_reward[col].view(torch.int32), traj.final_reward.view(torch.int32)), "and its reward"
    # the items tile the columns, bucket by bucket
    items = leaf.items.cpu().numpy()
    assert items[:, 1].sum() == leaf.n_cols and (items[1:, 0] == items[:-1, 0] + items[:-1, 1]).all() and items[:, 1].max() <= 256


@pytest.mark.parametrize("name,B", (("ternary4", 8192), ("ternary4", 3000), ("binary", 4096), ("a5c4", 8192), ("pruned", 8192)))
def test_leaf_learner_adds_up_the_per_lane_learners_bits(name, B):
    """rnad_rollout_learn_bucketed_compact with and without `leaf`: the same trajectory, counts, normalisers and per-row gradient tables,
    bit for bit; the counters are zero again afterwards; also with the finish left to the caller."""
    import rnad_hip

    tree, h, rec, fast, hp = _setup(name, B)
    T = 2 * h.max_depth
    leaf = rnad_hip.LeafPaths(h, B, tree.index_tensor, tree.chance_tensor, tree.value_tensor)

    def traj():
        return rnad_hip.Trajectory(h, B, T, DEV, with_observations=False, with_values=False, compact=True)

    one = traj()
    bk1, dl1, dv1 = rnad_hip.rollout_learn_bucketed_compact(h, one, rec, fast, hp, seed=21, lane0=77)
    for rep in range(2):  # (twice: the accumulators must be clean after a step)
        two = traj()
        bk2, dl2, dv2 = rnad_hip.rollout_learn_bucketed_compact(h, two, rec, fast, hp, seed=21, lane0=77, leaf=leaf)
        assert torch.equal(bk1.lane_ids, bk2.lane_ids) and torch.equal(one.alive, two.alive) and torch.equal(bk1.norm, bk2.norm)
        assert torch.equal(one.acts, two.acts) and torch.equal(one.indices, two.indices)
        assert torch.equal(one.final_reward.view(torch.int32), two.final_reward.view(torch.int32))
        assert torch.equal(dl1, dl2) and torch.equal(dv1, dv2), f"pass {rep}"
    assert torch.isfinite(dl1).all() and float(dl1.abs().sum()) > 0
    late = traj()
    bk3, none_l, none_v = rnad_hip.rollout_learn_bucketed_compact(h, late, rec, fast, hp, seed=21, lane0=77, norm_is_global=False, leaf=leaf)
    assert none_l is None and none_v is None
    dl3, dv3 = torch.empty_like(dl1), torch.empty_like(dv1)
    rnad_hip.bucket_finish(h, bk3, bk3.norm, hp, dl3, dv3)
    assert torch.equal(dl3, dl1) and torch.equal(dv3, dv1)


def _train(tree, leaf, graph, steps=8, B=1 << 14):
    from environment.episode import Buffer
    from learn.rnad import RNaD

    os.environ["RNAD_SAVE_DIR"] = tempfile.mkdtemp(prefix="rnad_leaf_")
    torch.manual_seed(3)
    rn = RNaD(tree=tree, device=DEV, directory_name="leaf", batch_size=B, eta=0.2, b1_adam=0.0, lr=1e-3,
              net_params={"type": "MLP", "max_actions": tree.max_actions, "width": 64})
    rn.initialize()
    rn.tabular_gate, rn.leaf_paths, rn.use_graph = 0, leaf, graph
    rn._seed_base, rn._seed_count = 99, 0
    buf = Buffer(1)
    for _ in range(steps):
        rn.train_step(buf, alpha=0.3)
        rn.total_steps += 1
    torch.cuda.synchronize()
    return rn


@pytest.mark.parametrize("graph", (False, True))
def test_training_on_leaf_paths_is_training_per_lane(graph):
    from test_hip_bucket import TREES, _native_tree

    tree = _native_tree(**TREES["ternary4"])
    per_lane = _train(tree, False, graph)
    on_paths = _train(tree, True, graph)
    auto = _train(tree, None, graph)
    h = tree.handle()
    assert on_paths._leaf_now(h, 1 << 14, 2 * h.max_depth) is not None and per_lane._leaf_now(h, 1 << 14, 2 * h.max_depth) is None
    assert auto._leaf_now(h, 1 << 14, 2 * h.max_depth) is not None, "6 561 leaf paths for 16 384 lanes (two lanes per path at least): automatic"
    assert auto._leaf_now(h, 1 << 13, 2 * h.max_depth) is None
    if graph:
        assert on_paths._graph["graph"] is not None and not on_paths._graph["failed"]
    for (k, a), b, c in zip(on_paths.net.named_parameters(), per_lane.net.parameters(), auto.net.parameters()):
        assert torch.equal(a, b) and torch.equal(a, c), k
    assert torch.equal(on_paths.last_episodes.indices, per_lane.last_episodes.indices)
    # a tree with ragged episode lengths keeps the per-lane learner
    ragged = _native_tree(**TREES["pruned"])
    probe = _train(ragged, None, False, steps=1, B=4096)
    assert probe._leaf_now(ragged.handle(), 4096, 2 * ragged.handle().max_depth) is None


def test_a_crowded_bucket_is_counted_by_the_rollout(monkeypatch):
    """With a sharp actor lanes pile up in one bucket.  r05: every leaf work item of that bucket counted all of its lanes, and RNaD._leaf_watch
    switched the leaf learner off for good.  r06: the rollout's work items count such a bucket (rnad_leaf_paths_t.col_count, forced here for
    EVERY bucket and for none) -- the leaf learner stays on, and training is the per-lane learner's bit for bit either way."""
    from environment.episode import Buffer
    from learn.rnad import RNaD
    from test_hip_bucket import TREES, _native_tree

    monkeypatch.setattr(RNaD, "LEAF_CHECK_EVERY", 4)
    tree = _native_tree(**TREES["ternary4"])
    h = tree.handle()
    B = 1 << 14

    def run(sharp):
        os.environ["RNAD_SAVE_DIR"] = tempfile.mkdtemp(prefix="rnad_leafwatch_")
        torch.manual_seed(3)
        rn = RNaD(tree=tree, device=DEV, directory_name="w", batch_size=B, eta=0.2, b1_adam=0.0, lr=1e-3,
                  net_params={"type": "MLP", "max_actions": tree.max_actions, "width": 64})
        rn.initialize()
        rn.tabular_gate = 0
        rn._seed_base, rn._seed_count = 5, 0
        if sharp:
            with torch.no_grad():
                for name, p in rn.net.named_parameters():
                    if name.startswith("policy_fc1"):
                        p.mul_(60.0)
        buf = Buffer(1)
        used = []
        for _ in range(12):
            rn.train_step(buf, alpha=0.3)
            used.append(getattr(rn.last_episodes.buckets.plan, "leaf", None) is not None and rn._leaf_now(h, B, 2 * h.max_depth) is not None)
            rn.total_steps += 1
        torch.cuda.synchronize()
        return rn, used

    flat, used_flat = run(False)
    assert all(used_flat) and flat._leaf_share < RNaD.DISTINCT_CROWDED
    monkeypatch.setenv("RNAD_LEAF_CROWDED_LANES", "1")  # every bucket counted by the rollout's work items
    sharp, used = run(True)
    assert all(used) and sharp._leaf_share > RNaD.DISTINCT_CROWDED, (used, sharp._leaf_share)
    assert int(sharp.last_episodes.buckets.plan.leaf.col_count.abs().sum().item()) == 0, "the learner leaves the counters zero"
    monkeypatch.setenv("RNAD_LEAF_CROWDED_LANES", "0")  # ... by the learner's (r05)
    sharp0, used0 = run(True)
    assert all(used0)
    monkeypatch.delenv("RNAD_LEAF_CROWDED_LANES")
    for (k, a), b in zip(sharp.net.named_parameters(), sharp0.net.parameters()):
        assert torch.equal(a, b), k
    # ... the watch still turns the learner on the distinct trajectories of a work item on long before DISTINCT_AFTER updates (it is what
    # runs whenever the leaf learner does not)
    assert sharp._distinct_crowded and sharp._distinct_now() and sharp.total_steps < RNaD.DISTINCT_AFTER
    assert not flat.__dict__.get("_distinct_crowded", False) and not flat._distinct_now()
    assert all(torch.isfinite(p).all() for p in sharp.net.parameters())
    # the same run with the leaf learner forced off from the start: the same parameters bit for bit (both learners add up the same sums)
    os.environ["RNAD_SAVE_DIR"] = tempfile.mkdtemp(prefix="rnad_leafwatch_")
    torch.manual_seed(3)
    ref = RNaD(tree=tree, device=DEV, directory_name="w", batch_size=B, eta=0.2, b1_adam=0.0, lr=1e-3,
               net_params={"type": "MLP", "max_actions": tree.max_actions, "width": 64})
    ref.initialize()
    ref.tabular_gate, ref.leaf_paths = 0, False
    ref._seed_base, ref._seed_count = 5, 0
    with torch.no_grad():
        for name, p in ref.net.named_parameters():
            if name.startswith("policy_fc1"):
                p.mul_(60.0)
    buf = Buffer(1)
    for _ in range(12):
        ref.train_step(buf, alpha=0.3)
        ref.total_steps += 1
    torch.cuda.synchronize()
    for (k, a), b in zip(sharp.net.named_parameters(), ref.net.parameters()):
        assert torch.equal(a, b), k
