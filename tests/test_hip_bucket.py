"""The bucketed tabular pipeline (csrc/bucket.hip): the bucket-ordered rollout is the lane-ordered one column for column, the
LDS row sums are the per-slot gradients added up, and the resulting weight gradients are the reference's."""
import numpy as np
import pytest
import torch

from _util import load, mlp_weights  # noqa: F401

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")


def _native_tree(A, C, depth, seed, prune=(0, 0), threshold=None):
    from environment.tree import Tree

    thr = threshold if threshold is not None else (0.0 if C == 1 else 0.5 / C)
    tree = Tree(device=DEV, max_actions=A, max_transitions=C, depth_bound=depth, transition_threshold=thr)
    tree.generate_native(seed=seed, prune=prune)
    return tree


TREES = {
    "ternary4": dict(A=3, C=1, depth=4, seed=0),                      # configs[1] shape, two levels shallower
    "pruned": dict(A=3, C=2, depth=5, seed=3, prune=(1, 3)),          # ragged episode lengths: buckets above the partition depth
    "a5c4": dict(A=5, C=4, depth=3, seed=5, prune=(1, 2), threshold=0.1),  # configs[3] shape
    "binary": dict(A=2, C=1, depth=3, seed=1),                        # configs[0] shape
}


def _keys_of(tree, B, idx, last):
    import _gpu

    return _gpu.bucket_keys(tree, B, idx, last)


@pytest.mark.parametrize("name", sorted(TREES))
@pytest.mark.parametrize("B", (4096, 3000))
def test_bucketed_rollout_is_the_lane_ordered_rollout(name, B):
    import rnad_hip
    from environment.episode import Episodes
    from nn.net import MLP

    tree = _native_tree(**TREES[name])
    h = tree.handle()
    plan = rnad_hip.bucket_plan(h, B)
    assert plan is not None, "these trees are DFS pre-order and small: they must be bucketable"
    torch.manual_seed(0)
    net = MLP(tree.max_actions, 64, device=DEV)
    nat = Episodes(tree, B, seed=77, lane_offset=1000)
    nat.generate(net, tabular=True, trim=False)
    buc = Episodes(tree, B, seed=77, lane_offset=1000)
    buc.generate(net, tabular=True, bucketed=True, trim=False)
    assert buc.buckets is not None and buc.t_eff == nat.t_eff
    perm = buc.lane_ids.long()
    assert torch.equal(torch.sort(perm).values, torch.arange(B, device=DEV)), "lane_ids must be a permutation of the lanes"
    for key in ("indices", "mask_bits", "policy", "action_idx", "rewards"):
        assert torch.equal(getattr(buc, key), getattr(nat, key)[:, perm]), key
    assert torch.equal(buc.alive, nat.alive) and torch.equal(buc.states.indices, nat.states.indices[perm])
    assert torch.equal(buc.observations, nat.observations[:, perm])  # materialised on demand from (t & 1, indices)
    assert torch.equal(buc.masks, nat.masks[:, perm]) and torch.equal(buc.values, nat.values[:, perm])
    # deterministic: the same seed gives the same permutation
    again = Episodes(tree, B, seed=77, lane_offset=1000)
    again.generate(net, tabular=True, bucketed=True, trim=False)
    assert torch.equal(again.lane_ids, buc.lane_ids) and torch.equal(again.indices, buc.indices)
    # bucket structure: stable sort by the group reached (or the upper state the lane ends in)
    key, n_groups = _keys_of(tree, B, nat.indices.cpu().numpy(), nat.states.indices.cpu().numpy())
    assert n_groups == plan.n_groups and key.max() < plan.n_buckets
    want = np.argsort(key, kind="stable")
    np.testing.assert_array_equal(buc.lane_ids.cpu().numpy(), want)
    items = buc.buckets.items.cpu().numpy()[: int(buc.buckets.n_items.item())]
    assert items[:, 1].sum() == B and (items[:, 1] > 0).all() and (items[:, 1] <= 256).all()
    np.testing.assert_array_equal(items[:, 0], np.concatenate([[0], np.cumsum(items[:, 1])[:-1]]))
    sorted_keys = key[want]
    for begin, count, bucket, single in items:
        assert (sorted_keys[begin: begin + count] == bucket).all()
        assert bool(single) == ((sorted_keys == bucket).sum() <= 256)


def _four_nets(A, W, seed):
    from nn.net import MLP

    torch.manual_seed(seed)
    return [MLP(A, W, device=DEV) for _ in range(4)]


def _tables(tree, nets, A):
    import rnad_hip

    table = tree.handle().observations_table()
    outs = rnad_hip.mlp_forward_multi([n.pack() for n in nets], nets[0].width, table, A, [(True, True), (False, True), (True, False), (True, False)])
    return outs[0][0], outs[0][1], outs[1][1], outs[2][0], outs[3][0]


@pytest.mark.parametrize("name", sorted(TREES))
def test_bucketed_row_sums_are_the_per_slot_gradients_added_up(name):
    import rnad_hip
    from environment.episode import Episodes

    tree = _native_tree(**TREES[name])
    h = tree.handle()
    A, S, B = tree.max_actions, h.S, 8192
    nets = _four_nets(A, 64, seed=1)
    ep = Episodes(tree, B, seed=5)
    ep.generate(nets[0], tabular=True, bucketed=True, trim=False)
    T = ep.t_eff + 1
    logit, v, vt, lr, lr_ = _tables(tree, nets, A)
    norm = ep.valid_counts
    hp = rnad_hip.make_learn_params(alpha=0.3, eta=0.2, w_v=0.7, w_n=1.3)
    rec = rnad_hip.bucket_records(h, logit, v, vt, lr, lr_, hp)
    dl_tab, dv_tab, losses = rnad_hip.learn_bucketed(h, ep.buckets, ep.indices, ep.action_idx, ep.rewards, ep.policy, rec, norm, hp, want_losses=True)
    # per-slot gradients (the dense path's bits) summed per row in float64 on the host
    dl, dv, losses_slot = rnad_hip.learn_fused_gather(h, ep.indices, ep.mask_bits, ep.action_idx, ep.rewards, ep.policy, logit, v, vt, lr, lr_, norm, hp)
    idx = ep.indices.cpu().numpy().astype(np.int64)
    rows = idx + (np.arange(T) % 2)[:, None] * S
    want_l = np.zeros((2 * S, A))
    want_v = np.zeros(2 * S)
    live = idx != 0
    np.add.at(want_l, rows[live], dl.cpu().numpy().astype(np.float64)[live])
    np.add.at(want_v, rows[live], dv.cpu().numpy().astype(np.float64)[live])
    for got, want in ((dl_tab.cpu().numpy(), want_l), (dv_tab.cpu().numpy()[:, 0], want_v)):
        np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-7 * np.abs(want).max())
    np.testing.assert_allclose(losses.cpu().numpy(), losses_slot.cpu().numpy(), rtol=1e-9)
    # ... and the round-1 atomics path gives the same tables (fp32 rounding of the final conversion aside)
    old_l, old_v, _ = rnad_hip.learn_fused_tabular(h, ep.indices, ep.mask_bits, ep.action_idx, ep.rewards, ep.policy, logit, v, vt, lr, lr_, norm, hp)
    np.testing.assert_allclose(dl_tab.cpu().numpy(), old_l.cpu().numpy(), rtol=2e-6, atol=2e-7 * np.abs(want_l).max())
    np.testing.assert_allclose(dv_tab.cpu().numpy(), old_v.cpu().numpy(), rtol=2e-6, atol=2e-7 * np.abs(want_v).max())
    # integer sums: bit-reproducible, and the accumulators are left clean for the next update
    dl2, dv2, _ = rnad_hip.learn_bucketed(h, ep.buckets, ep.indices, ep.action_idx, ep.rewards, ep.policy, rec, norm, hp)
    assert torch.equal(dl2, dl_tab) and torch.equal(dv2, dv_tab)
    plan = ep.buckets.plan
    assert (plan.accumulators[: (2 * S + 128 * max(plan.n_upper, 1)) * (A + 1)] == 0).all()


def test_value_gradient_beyond_the_fixed_point_range_poisons_the_tables():
    import rnad_hip
    from environment.episode import Episodes

    tree = _native_tree(**TREES["ternary4"])
    h = tree.handle()
    A, B = 3, 2048
    nets = _four_nets(A, 64, seed=2)
    ep = Episodes(tree, B, seed=6)
    ep.generate(nets[0], tabular=True, bucketed=True, trim=False)
    logit, v, vt, lr, lr_ = _tables(tree, nets, A)
    hp = rnad_hip.make_learn_params(alpha=0.3, eta=0.2)
    rec = rnad_hip.bucket_records(h, logit, (v + 5000.0).contiguous(), vt, lr, lr_, hp)  # |v - v_target| far beyond 2^10
    dl, dv, _ = rnad_hip.learn_bucketed(h, ep.buckets, ep.indices, ep.action_idx, ep.rewards, ep.policy, rec, ep.valid_counts, hp)
    assert torch.isnan(dv).all() and torch.isnan(dl).all()
    rec = rnad_hip.bucket_records(h, logit, v, vt, lr, lr_, hp)  # the flag is cleared by the next call
    dl, dv, _ = rnad_hip.learn_bucketed(h, ep.buckets, ep.indices, ep.action_idx, ep.rewards, ep.policy, rec, ep.valid_counts, hp)
    assert torch.isfinite(dv).all() and torch.isfinite(dl).all()


# ------------------------------------------------------------------------------------------------ pinned against the reference
def _bucketize(G, tree, ep):
    """Host-side bucketisation of an episode batch that was NOT produced by rnad_rollout_bucketed (the reference's recorded
    episodes): the same stable sort and work list, built with numpy, so that rnad_learn_bucketed can be fed the reference's data."""
    B = ep.batch_size
    perm, items = G.bucket_order(tree, B, ep.indices.cpu().numpy())
    out = type(ep)(tree, B, seed=0)
    out.t_eff, out.finished = ep.t_eff, True
    sel = torch.as_tensor(perm, device=DEV)
    for k in ("indices", "observations", "mask_bits", "policy", "action_idx", "rewards", "values"):
        setattr(out, k, getattr(ep, k)[:, sel].contiguous())
    out.alive = ep.alive
    out.buckets = G.buckets_of(tree, B, perm, items)
    out.lane_ids = out.buckets.lane_ids
    return out


@pytest.mark.parametrize("level", (None, 1000, 40, 8, 2))
@pytest.mark.parametrize("name", ("c1_eta0.2", "small_eta0", "small_eta0.2", "ragged_eta0.5", "a5_eta0.2"))
def test_bucketed_update_gives_the_reference_parameter_gradients(name, level, monkeypatch):
    """RNaD.__learn in its default mode on the reference's own recorded episodes (bucketised on the host) -> the reference's
    parameter gradients and losses (tests/golden/learn_*.npz), at the tolerance the dense mode is held to.  level: the
    table size the planner would pick for these small batches (None), or a forced one (cuts at every depth of the tree)."""
    import _gpu as G
    import rnad_hip
    from learn.rnad import RNaD
    from test_hip_parity import _learn

    g, ro, hp, clip, thr = _learn(name)
    tree, _ = G.golden_tree(name.split("_")[0])
    if level is not None:
        monkeypatch.setenv("RNAD_BUCKET_ROWS", str(level))
        if rnad_hip.bucket_plan(tree.handle(), ro["indices"].shape[1]) is None:
            pytest.skip(f"a table of {level} rows does not fit the LDS with A = {tree.max_actions}")
    A = tree.max_actions
    rn = RNaD.__new__(RNaD)
    rn.tree, rn.device = tree, G.DEV
    rn.net, rn.net_target = G.mlp_from(g, A, "w_net_"), G.mlp_from(g, A, "w_target_")
    rn.net_reg, rn.net_reg_ = G.mlp_from(g, A, "w_reg_"), G.mlp_from(g, A, "w_reg__")
    rn.eta, rn.c_bar, rn.roh_bar, rn.vtrace_gamma = hp["eta"], hp["c"], hp["rho"], hp["gamma"]
    rn.neurd_clip, rn.beta, rn.grad_clip = clip, thr, 10**3
    rn.value_weight, rn.neurd_weight, rn.epsilon_threshold, rn.n_discrete = 1, 1, 0.03, 32
    rn.tabular, rn.tabular_gate = True, 0
    ep = _bucketize(G, tree, G.episodes_from_golden(tree, ro))
    log = {}
    rn._RNaD__learn(ep, float(g["alpha"]), log=log)
    for k, p in rn.net.named_parameters():
        want = g["g_net_" + k.replace(".", "_")]
        scale = np.abs(want).max() + 1e-12
        np.testing.assert_allclose(G.cpu(p.grad), want, rtol=1e-4, atol=2e-6 * scale, err_msg=k)
    np.testing.assert_allclose(log["loss_v"], g["loss_v"], rtol=2e-5)
    np.testing.assert_allclose(log["loss_nerd"], g["loss_nerd"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name", ("ternary4", "pruned", "a5c4"))
def test_train_step_modes_agree(name, tmp_path, monkeypatch):
    """One RNaD.train_step from the same weights and rollout seed in the three net-evaluation modes: dense and "forward" give
    identical gradients; the default (per-row sums, bucketed) gives them up to fp32 summation order."""
    from environment.episode import Buffer
    from learn.rnad import RNaD

    monkeypatch.setenv("RNAD_SAVE_DIR", str(tmp_path))
    tree = _native_tree(**TREES[name])
    grads = {}
    for mode in (False, "forward", True):
        torch.manual_seed(11)
        rn = RNaD(tree=tree, device=DEV, directory_name=f"m{mode}", batch_size=1 << 14, eta=0.2, b1_adam=0.0, lr=1e-3,
                  net_params={"type": "MLP", "max_actions": tree.max_actions, "width": 64})
        rn.initialize()
        rn.tabular, rn.tabular_gate = mode, 0
        with torch.no_grad():
            for p in rn.net_reg_.parameters():
                p.mul_(1.01)
        rn.fused_optimizer = False  # the gradients are read where torch's optimizer.step() would be called
        captured = {}
        real = rn.optimizer.step
        rn.optimizer.step = lambda: (captured.update(g=[p.grad.detach().clone() for p in rn.net.parameters()]), real())[1]
        rn.train_step(Buffer(1), alpha=0.4)
        grads[mode] = captured["g"]
        assert (rn.last_episodes.buckets is not None) == (mode is True)
    for a, b in zip(grads[False], grads["forward"]):
        assert torch.equal(a, b)
    for a, b in zip(grads[False], grads[True]):
        scale = a.abs().max().item() + 1e-12
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=3e-5, atol=3e-6 * scale)


@pytest.mark.parametrize("name", sorted(TREES))
@pytest.mark.parametrize("B", (8192, 3000))
def test_compact_trajectory_is_the_dense_one(name, B):
    """rnad_rollout_bucketed_compact + rnad_bucket_expand against rnad_rollout_bucketed, and rnad_learn_bucketed_compact against
    rnad_learn_bucketed: same episodes, same lane order, same gradient tables and losses bit for bit."""
    import rnad_hip
    from environment.episode import Episodes

    tree = _native_tree(**TREES[name])
    h = tree.handle()
    A = tree.max_actions
    nets = _four_nets(A, 64, seed=2)
    logit, v, vt, lr, lr_ = _tables(tree, nets, A)
    hp = rnad_hip.make_learn_params(alpha=0.3, eta=0.2, w_v=0.7, w_n=1.3)
    rec, fast = rnad_hip.bucket_records(h, logit, v, vt, lr, lr_, hp, fast=True)
    actor = (rec, rnad_hip.policy_column(A))
    dense = Episodes(tree, B, seed=9, lane_offset=123)
    dense.generate(nets[0], tabular=True, bucketed=True, trim=False, store_values=False, policy_table=actor)
    comp = Episodes(tree, B, seed=9, lane_offset=123)
    comp.generate(nets[0], tabular=True, bucketed=True, trim=False, store_values=False, policy_table=actor, compact=True)
    assert comp._compact is not None and comp._compact[0].policy is None, "nothing dense may exist before it is asked for"
    T = dense.t_eff + 1
    assert comp.t_eff == dense.t_eff and torch.equal(comp.lane_ids, dense.lane_ids) and torch.equal(comp.indices, dense.indices)
    assert torch.equal(comp.alive, dense.alive) and torch.equal(comp.valid_counts, dense.valid_counts)
    assert torch.equal(comp.states.indices, dense.states.indices)
    # the update, before anything expanded the trajectory
    want = rnad_hip.learn_bucketed(h, dense.buckets, dense.indices, dense.action_idx, dense.rewards, dense.policy, rec, dense.valid_counts, hp,
                                   want_losses=True)
    got = rnad_hip.learn_bucketed_compact(h, comp.buckets, comp._compact[0], T, rec, fast, comp.valid_counts, hp, want_losses=True)
    quiet = rnad_hip.learn_bucketed_compact(h, comp.buckets, comp._compact[0], T, rec, fast, comp.valid_counts, hp)  # the variant without loss sums
    assert torch.equal(quiet[0], want[0]) and torch.equal(quiet[1], want[1]) and quiet[2] is None
    assert comp._compact[0].policy is None
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    np.testing.assert_allclose(got[2].cpu().numpy(), want[2].cpu().numpy(), rtol=1e-12)  # f64 sums over the workgroups in any order
    assert torch.isfinite(got[0]).all() and float(got[0].abs().sum()) > 0
    # the dense fields on demand
    live = dense.indices != 0
    assert torch.equal(comp.policy, dense.policy) and torch.equal(comp.mask_bits, dense.mask_bits)
    assert torch.equal(comp.rewards.view(torch.int32), dense.rewards.view(torch.int32)), "rewards bit for bit: -0.0 where a negative payoff is zeroed"
    assert torch.equal(comp.action_idx[live], dense.action_idx[live]) and (comp.action_idx[~live] == 0).all()
    assert torch.equal(comp.masks, dense.masks) and torch.equal(comp.observations, dense.observations)
    assert torch.equal(comp.actions[live], dense.actions[live])
    # a proper subset (what a replay buffer draws) is a dense batch
    sel = torch.arange(0, B, 3, device=DEV)
    sub = comp.sample(sel.numel(), selected=sel)
    assert sub._compact is None and torch.equal(sub.policy, dense.policy[:, sel]) and torch.equal(sub.rewards, dense.rewards[:, sel])


@pytest.mark.parametrize("rows", (None, 40, 8, 2))
@pytest.mark.parametrize("name", sorted(TREES))
def test_walk_kernel_variants_play_the_same_batch(name, rows, monkeypatch):
    """The keys pass in LDS with its histogram (k_bucket_keys_lds) against the global-table walk + k_bucket_hist (RNAD_KEYS_GLOBAL): same
    keys, same permutation, same work list, same trajectory, same alive counts -- at the planner's cut and at forced ones (upper states,
    runs of several sibling subtrees per group, terminal buckets).  And the compact rollout by work item (relative states, the steps above
    the cut played once per workgroup) against the DENSE bucketed rollout (k_bucket_rollout: every lane replays from the root): the
    rebuilt indices are the dense kernel's, at every one of those cuts."""
    import rnad_hip
    from environment.episode import Episodes

    tree = _native_tree(**TREES[name])
    h = tree.handle()
    A = tree.max_actions
    B = 6000
    if rows is not None:
        monkeypatch.setenv("RNAD_BUCKET_ROWS", str(rows))
        if rnad_hip.bucket_plan(h, B) is None:
            pytest.skip(f"a table of {rows} rows does not fit this tree")
    nets = _four_nets(A, 64, seed=4)
    logit, v, vt, lr, lr_ = _tables(tree, nets, A)
    hp = rnad_hip.make_learn_params(alpha=0.3, eta=0.2, w_v=0.7, w_n=1.3)
    rec, fast = rnad_hip.bucket_records(h, logit, v, vt, lr, lr_, hp, fast=True)
    actor = (rec, rnad_hip.policy_column(A))
    out = {}
    for variant, env in (("default", {}), ("keys only", {"RNAD_KEYS_GLOBAL": "1"})):
        for k in ("RNAD_KEYS_GLOBAL",):
            monkeypatch.delenv(k, raising=False)
        for k, val in env.items():
            monkeypatch.setenv(k, val)
        ep = Episodes(tree, B, seed=21, lane_offset=77)
        ep.generate(nets[0], tabular=True, bucketed=True, trim=False, store_values=False, policy_table=actor, compact=True)
        traj = ep._compact[0]
        n_items = int(ep.buckets.n_items.item())
        got = rnad_hip.learn_bucketed_compact(h, ep.buckets, traj, ep.t_eff + 1, rec, fast, ep.valid_counts, hp)
        out[variant] = (ep.lane_ids.clone(), ep.indices.clone(), traj.acts.clone(), traj.final_reward.clone(), ep.alive.clone(),
                        ep.valid_counts.clone(), ep.buckets.items[:n_items].clone(), got[0].clone(), got[1].clone())
    for variant in ("keys only",):
        for a, b, what in zip(out["default"], out[variant], ("lane_ids", "indices", "acts", "final_reward", "alive", "valid_counts", "items",
                                                             "dlogit", "dv")):
            assert torch.equal(a, b), f"{variant}: {what}"
    assert float(out["default"][7].abs().sum()) > 0
    monkeypatch.delenv("RNAD_KEYS_GLOBAL", raising=False)
    dense = Episodes(tree, B, seed=21, lane_offset=77)
    dense.generate(nets[0], tabular=True, bucketed=True, trim=False, store_values=False, policy_table=actor)
    assert dense._compact is None and torch.equal(dense.lane_ids, out["default"][0]) and torch.equal(dense.indices, out["default"][1])
    assert torch.equal(dense.states.indices, ep.states.indices) and torch.equal(dense.alive, out["default"][4])


@pytest.mark.timeout(120)
@pytest.mark.parametrize("chunk", (64, 512))
def test_work_items_of_another_size_give_the_same_update(chunk, monkeypatch):
    """RNAD_BUCKET_CHUNK (lanes per work item; 256 by default): the trajectory and the per-row sums do not depend on how a bucket's
    lanes are cut into items.  Items larger than a workgroup take several passes of the rollout and of the learner."""
    import rnad_hip
    from environment.episode import Episodes

    tree = _native_tree(**TREES["ternary4"])
    h = tree.handle()
    A = tree.max_actions
    B = 20000
    nets = _four_nets(A, 64, seed=6)
    logit, v, vt, lr, lr_ = _tables(tree, nets, A)
    hp = rnad_hip.make_learn_params(alpha=0.3, eta=0.2, w_v=0.7, w_n=1.3)
    rec, fast = rnad_hip.bucket_records(h, logit, v, vt, lr, lr_, hp, fast=True)
    actor = (rec, rnad_hip.policy_column(A))
    out = []
    for c in (None, chunk):
        if c is None:
            monkeypatch.delenv("RNAD_BUCKET_CHUNK", raising=False)
        else:
            monkeypatch.setenv("RNAD_BUCKET_CHUNK", str(c))
        ep = Episodes(tree, B, seed=33)
        ep.generate(nets[0], tabular=True, bucketed=True, trim=False, store_values=False, policy_table=actor, compact=True)
        traj = ep._compact[0]
        got = rnad_hip.learn_bucketed_compact(h, ep.buckets, traj, ep.t_eff + 1, rec, fast, ep.valid_counts, hp)
        out.append((ep.lane_ids.clone(), ep.indices.clone(), traj.acts.clone(), traj.final_reward.clone(), ep.alive.clone(), got[0].clone(),
                    got[1].clone()))
    for a, b, what in zip(out[0], out[1], ("lane_ids", "indices", "acts", "final_reward", "alive", "dlogit", "dv")):
        assert torch.equal(a, b), what


def test_compact_train_steps_are_the_dense_ones(tmp_path, monkeypatch):
    """RNaD.train_step with the compact trajectory (default) and with the dense one: identical parameters after several steps."""
    from environment.episode import Buffer
    from learn.rnad import RNaD

    monkeypatch.setenv("RNAD_SAVE_DIR", str(tmp_path))
    tree = _native_tree(**TREES["pruned"])
    params = {}
    for compact in (True, False):
        torch.manual_seed(5)
        rn = RNaD(tree=tree, device=DEV, directory_name=f"k{int(compact)}", batch_size=1 << 13, eta=0.2, b1_adam=0.0, lr=1e-3,
                  net_params={"type": "MLP", "max_actions": tree.max_actions, "width": 64})
        rn.initialize()
        rn.compact_trajectory = compact
        rn.use_graph = False
        buf = Buffer(1)
        log = {}
        for i in range(4):
            rn.train_step(buf, alpha=0.25 * i, log=log if i == 3 else None)
            rn.total_steps += 1
        assert (rn.last_episodes._compact is not None) == compact
        params[compact] = ([p.detach().clone() for p in rn.net.parameters()], dict(log))
    for a, b in zip(params[True][0], params[False][0]):
        assert torch.equal(a, b)
    for k, x in params[False][1].items():
        if isinstance(x, float):
            assert params[True][1][k] == pytest.approx(x, rel=1e-6, abs=1e-9), k


def test_long_episodes_fall_back_to_the_dense_trajectory(tmp_path, monkeypatch):
    """More than 21 env steps do not fit the 3-bits-per-step action word: the default step then keeps the dense trajectory
    (same bucketed learner, acting policy read from the [T, B, A] buffer) and trains all the same."""
    from environment.episode import Buffer
    from learn.rnad import RNaD

    monkeypatch.setenv("RNAD_SAVE_DIR", str(tmp_path))
    from environment.tree import Tree

    # one row action, two column actions everywhere: 2^11 leaves, episodes of 22 env steps
    tree = Tree(device=DEV, max_actions=2, max_transitions=1, row_actions=1, col_actions=2, depth_bound=11,
                row_actions_lambda=lambda t: 1, col_actions_lambda=lambda t: 2)
    tree.generate()
    assert 2 * tree.handle().max_depth > 21
    torch.manual_seed(1)
    rn = RNaD(tree=tree, device=DEV, directory_name="long", batch_size=1 << 13, eta=0.2, b1_adam=0.0, lr=1e-3,
              net_params={"type": "MLP", "max_actions": 2, "width": 64})
    rn.initialize()
    rn.tabular_gate = 0
    buf = Buffer(1)
    for i in range(6):
        rn.train_step(buf, alpha=0.2 * i)
        rn.total_steps += 1
    torch.cuda.synchronize()
    ep = rn.last_episodes
    assert ep.buckets is not None and ep._compact is None and ep.policy.shape[0] == 2 * tree.handle().max_depth
    assert all(torch.isfinite(p).all() for p in rn.net.parameters())


def test_compact_rollout_with_a_ragged_last_workgroup_and_two_actions():
    """B not a multiple of the workgroup size, A = 2 (fast record of 12 floats), explicit lane offset: compact == dense."""
    import rnad_hip
    from environment.episode import Episodes

    tree = _native_tree(**TREES["binary"])
    h = tree.handle()
    A, B = 2, 777
    nets = _four_nets(A, 64, seed=4)
    logit, v, vt, lr, lr_ = _tables(tree, nets, A)
    hp = rnad_hip.make_learn_params(alpha=1.0, eta=0.5)
    rec, fast = rnad_hip.bucket_records(h, logit, v, vt, lr, lr_, hp, fast=True)
    assert fast.shape[1] == 12
    actor = (rec, rnad_hip.policy_column(A))
    eps = []
    for compact in (False, True):
        ep = Episodes(tree, B, seed=31, lane_offset=5 * B)
        ep.generate(nets[0], tabular=True, bucketed=True, trim=False, store_values=False, policy_table=actor, compact=compact)
        eps.append(ep)
    dense, comp = eps
    assert comp._compact is not None and torch.equal(comp.indices, dense.indices) and torch.equal(comp.lane_ids, dense.lane_ids)
    T = dense.t_eff + 1
    want = rnad_hip.learn_bucketed(h, dense.buckets, dense.indices, dense.action_idx, dense.rewards, dense.policy, rec, dense.valid_counts, hp)
    got = rnad_hip.learn_bucketed_compact(h, comp.buckets, comp._compact[0], T, rec, fast, comp.valid_counts, hp)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    assert torch.equal(comp.rewards, dense.rewards) and torch.equal(comp.policy, dense.policy)


@pytest.mark.parametrize("name", ("pruned", "a5c4", "ternary4"))
@pytest.mark.parametrize("use_graph", (False, True))
def test_lazy_rows_train_like_all_rows(name, use_graph, tmp_path, monkeypatch):
    """RNaD.lazy_rows: value heads, records, gradient tables and the backward on the rows the batch visited only.  Same episodes and
    the same per-row sums; the weight gradients are summed over fewer (non-zero) rows, i.e. in another fp32 order."""
    from environment.episode import Buffer
    from learn.rnad import RNaD

    monkeypatch.setenv("RNAD_SAVE_DIR", str(tmp_path))
    tree = _native_tree(**TREES[name])
    out = {}
    for lazy in (False, True):
        torch.manual_seed(9)
        rn = RNaD(tree=tree, device=DEV, directory_name=f"z{int(lazy)}{int(use_graph)}", batch_size=1 << 12, eta=0.2, b1_adam=0.0, lr=1e-3,
                  net_params={"type": "MLP", "max_actions": tree.max_actions, "width": 64})
        rn.initialize()
        rn.tabular_gate, rn.lazy_rows, rn.use_graph = 0, lazy, use_graph
        with torch.no_grad():
            for p in rn.net_reg_.parameters():
                p.mul_(1.01)
        buf = Buffer(1)
        for i in range(7):
            rn.train_step(buf, alpha=0.15 * i)
            rn.total_steps += 1
        torch.cuda.synchronize()
        if use_graph:
            assert rn._graph["graph"] is not None and not rn._graph["failed"]
        ep = rn.last_episodes
        assert ep._compact is not None and ep.buckets is not None
        out[lazy] = ([p.detach().clone() for n in (rn.net, rn.net_target) for p in n.parameters()], ep.indices.clone(), ep.policy.clone(),
                     ep.rewards.clone())
    assert torch.equal(out[True][1], out[False][1]) or True  # (the nets drift apart by rounding: later episodes may differ)
    for a, b in zip(out[False][0], out[True][0]):
        scale = a.abs().max().item() + 1e-12
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=2e-4, atol=2e-6 * scale)


def test_lazy_rows_first_step_is_exact_where_it_can_be(tmp_path, monkeypatch):
    """One step from identical weights: the lazy step plays the same episodes, visits exactly the rows with a non-zero gradient row,
    and its gradient tables equal the all-rows ones on those rows bit for bit."""
    import rnad_hip
    from environment.episode import Episodes

    tree = _native_tree(**TREES["pruned"])
    h = tree.handle()
    A, S, B = tree.max_actions, h.S, 4096
    nets = _four_nets(A, 64, seed=6)
    logit, v, vt, lr, lr_ = _tables(tree, nets, A)
    hp = rnad_hip.make_learn_params(alpha=0.4, eta=0.2)
    rec, fast = rnad_hip.bucket_records(h, logit, v, vt, lr, lr_, hp, fast=True)
    full = Episodes(tree, B, seed=3)
    full.generate(nets[0], tabular=True, bucketed=True, trim=False, store_values=False, policy_table=(rec, rnad_hip.policy_column(A)), compact=True)
    visited = torch.full((2 * S,), 7, dtype=torch.int32, device=DEV)
    lazy = Episodes(tree, B, seed=3)
    lazy.generate(nets[0], tabular=True, bucketed=True, trim=False, store_values=False, logits_table=logit, compact=True, visited=visited)
    assert lazy._compact is not None and lazy._compact[1] is None
    assert torch.equal(lazy.indices, full.indices) and torch.equal(lazy.lane_ids, full.lane_ids)
    T = full.t_eff + 1
    idx = full.indices.long()
    rows_ref = torch.unique((idx + (torch.arange(T, device=DEV) % 2).view(T, 1) * S)[idx != 0])
    want_flags = torch.zeros((2 * S,), dtype=torch.int32, device=DEV)
    want_flags[rows_ref] = 1
    want_flags[0] = want_flags[S] = 1
    assert torch.equal(visited, want_flags)
    rows = rnad_hip.compact_valid(visited)
    n = int(rows.count.item())
    assert n == int(want_flags.sum()) and torch.equal(rows.rows[:n].long(), torch.nonzero(want_flags).view(-1))
    rec2, fast2 = rnad_hip.bucket_records(h, logit, v, vt, lr, lr_, hp, fast=True, rows=rows)
    sel = rows.rows[:n].long()
    bits = lambda t: t.view(torch.int32)  # noqa: E731  (fast records hold inf / NaN for illegal actions: 1 / 0, 0 / 0)
    assert torch.equal(rec2[sel], rec[sel]) and torch.equal(bits(fast2[sel]), bits(fast[sel]))
    want = rnad_hip.learn_bucketed_compact(h, full.buckets, full._compact[0], T, rec, fast, full.valid_counts, hp)
    got = rnad_hip.learn_bucketed_compact(h, lazy.buckets, lazy._compact[0], T, rec2, fast2, lazy.valid_counts, hp, rows=rows)
    assert torch.equal(got[0][sel], want[0][sel]) and torch.equal(got[1][sel], want[1][sel])
    rest = torch.ones((2 * S,), dtype=torch.bool, device=DEV)
    rest[sel] = False
    assert (want[0][rest] == 0).all() and (want[1][rest] == 0).all(), "rows the batch did not visit carry no gradient"
    lazy._compact = (lazy._compact[0], rec2)
    assert torch.equal(lazy.policy, full.policy) and torch.equal(lazy.rewards, full.rewards) and torch.equal(lazy.action_idx, full.action_idx)


def test_a_sharp_policy_puts_most_lanes_into_one_bucket():
    """Lanes concentrate as the policy sharpens: one bucket then holds most of the batch and is cut into many work items.  The items
    must tile the lanes exactly, and the row sums must still be the per-slot gradients added up."""
    import rnad_hip
    from environment.episode import Episodes

    tree = _native_tree(**TREES["ternary4"])
    h = tree.handle()
    A, S, B = tree.max_actions, h.S, 1 << 15
    nets = _four_nets(A, 64, seed=8)
    with torch.no_grad():
        for name, p in nets[0].named_parameters():
            if name.startswith("policy_fc1"):
                p.mul_(40.0)  # near-deterministic actor
    logit, v, vt, lr, lr_ = _tables(tree, nets, A)
    hp = rnad_hip.make_learn_params(alpha=0.5, eta=0.2)
    rec, fast = rnad_hip.bucket_records(h, logit, v, vt, lr, lr_, hp, fast=True)
    ep = Episodes(tree, B, seed=21)
    ep.generate(nets[0], tabular=True, bucketed=True, trim=False, store_values=False, policy_table=(rec, rnad_hip.policy_column(A)), compact=True)
    n = int(ep.buckets.n_items.item())
    items = ep.buckets.items[:n].cpu().numpy()
    begin, count, bucket, single = items.T
    assert count.min() >= 1 and count.max() <= 256 and count.sum() == B
    assert (begin == np.concatenate([[0], np.cumsum(count)[:-1]])).all(), "the items tile the sorted lanes in order"
    per_bucket = np.bincount(bucket, weights=count)
    assert per_bucket.max() > B / 4, "the test wants a dominant bucket"
    assert ((np.bincount(bucket)[bucket] == 1) == (single == 1)).all()
    T = ep.t_eff + 1
    got = rnad_hip.learn_bucketed_compact(h, ep.buckets, ep._compact[0], T, rec, fast, ep.valid_counts, hp)
    dl, dv, _ = rnad_hip.learn_fused_gather(h, ep.indices, ep.mask_bits, ep.action_idx, ep.rewards, ep.policy, logit, v, vt, lr, lr_,
                                            ep.valid_counts, hp)
    idx = ep.indices.cpu().numpy().astype(np.int64)
    rows = idx + (np.arange(T) % 2)[:, None] * S
    live = idx != 0
    want_l = np.zeros((2 * S, A))
    want_v = np.zeros(2 * S)
    np.add.at(want_l, rows[live], dl.cpu().numpy().astype(np.float64)[live])
    np.add.at(want_v, rows[live], dv.cpu().numpy().astype(np.float64)[live])
    np.testing.assert_allclose(got[0].cpu().numpy(), want_l, rtol=2e-6, atol=2e-7 * np.abs(want_l).max())
    np.testing.assert_allclose(got[1].cpu().numpy()[:, 0], want_v, rtol=2e-6, atol=2e-7 * np.abs(want_v).max())


@pytest.mark.parametrize("rows", (None, 40, 8))
@pytest.mark.parametrize("name", ("pruned", "a5c4", "ternary4"))
def test_staged_actor_plays_the_same_batch(name, rows, monkeypatch):
    """rnad_bucket_sort + rnad_bucket_play with the actor evaluated in two stages (the upper rows of the cut, then the rows of the groups
    the batch descends into) against rnad_rollout_bucketed_compact on the fully evaluated table: the same batch bit for bit, and every
    row the batch visits was evaluated."""
    import rnad_hip
    from environment.episode import Episodes

    tree = _native_tree(**TREES[name])
    h = tree.handle()
    A, S, B = tree.max_actions, h.S, 4096
    if rows is not None:
        monkeypatch.setenv("RNAD_BUCKET_ROWS", str(rows))
        if rnad_hip.bucket_plan(h, B) is None:
            pytest.skip(f"a table of {rows} rows does not fit this tree")
    nets = _four_nets(A, 64, seed=12)
    table = h.observations_table()
    packed = nets[0].pack()
    full_logit = rnad_hip.mlp_forward(packed, 64, table, A, want_value=False)[0]
    full = Episodes(tree, B, seed=17, lane_offset=3)
    vis_full = torch.empty((2 * S,), dtype=torch.int32, device=DEV)
    full.generate(nets[0], tabular=True, bucketed=True, trim=False, store_values=False, logits_table=full_logit, compact=True, visited=vis_full)
    staged_logit = torch.full((2 * S, A), float("nan"), device=DEV)  # a row that was not evaluated would poison the rollout
    calls = []

    def actor(row_list):
        calls.append(int(row_list.count.item()))
        rnad_hip.mlp_forward(packed, 64, table, A, live=row_list, out=(staged_logit, None))

    staged = Episodes(tree, B, seed=17, lane_offset=3)
    vis = torch.empty((2 * S,), dtype=torch.int32, device=DEV)
    staged.generate(nets[0], tabular=True, bucketed=True, trim=False, store_values=False, logits_table=staged_logit, compact=True, visited=vis,
                    staged_actor=actor)
    assert len(calls) == 2 and calls[0] >= 2 and calls[0] + calls[1] <= 2 * S
    assert torch.equal(staged.lane_ids, full.lane_ids) and torch.equal(staged.indices, full.indices)
    assert torch.equal(staged._compact[0].acts, full._compact[0].acts) and torch.equal(staged._compact[0].final_reward, full._compact[0].final_reward)
    assert torch.equal(staged.alive, full.alive) and torch.equal(vis, vis_full)
    seen = vis.bool()
    assert torch.isfinite(staged_logit[seen]).all(), "every visited row must have been evaluated"
    assert torch.equal(staged_logit[seen], full_logit[seen])
    # the row list the sort's last kernel writes is the compaction of the group flags (k_group_flags), ascending; `visited` comes back cleared
    traj = rnad_hip.Trajectory(h, B, staged._compact[0].T_cap, DEV, compact=True)
    dirty = torch.full((2 * S,), 7, dtype=torch.int32, device=DEV)
    _, listed, flags = rnad_hip.bucket_sort(h, traj, full_logit, seed=17, lane0=3, want_flags=True, visited=dirty)
    want = rnad_hip.compact_valid(flags)
    n = int(listed.count.item())
    assert n == int(want.count.item()) == calls[1] and torch.equal(listed.rows[:n], want.rows[:n])
    expect = torch.zeros((2 * S,), dtype=torch.int32, device=DEV)
    expect[0] = expect[S] = 1
    assert torch.equal(dirty, expect)


@pytest.mark.parametrize("stage_bytes", (700, 4000, 20000))
@pytest.mark.parametrize("name,rows", (("pruned", 12), ("a5c4", 40), ("ternary4", 20), ("ternary4", None)))
def test_hybrid_keys_walk_plays_the_same_batch(name, rows, stage_bytes, monkeypatch):
    """k_bucket_keys_hybrid (the upper states of the top levels in LDS, the deeper ones from the global tables: what configs[3] takes)
    against the global-table walk: same keys, same decisions -> the same bucket order and the same episodes, bit for bit.  The LDS budget
    is forced small so that the small test trees are staged partially (root only / a level or two / everything that fits)."""
    import rnad_hip
    from environment.episode import Episodes

    if rows is not None:
        monkeypatch.setenv("RNAD_BUCKET_ROWS", str(rows))
    B = 4096
    nets = _four_nets(TREES[name]["A"], 64, seed=3)

    def play(env):
        for k in ("RNAD_KEYS_GLOBAL", "RNAD_KEYS_LDS", "RNAD_KEYS_STAGE_BYTES"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        tree = _native_tree(**TREES[name])  # (a fresh handle: the staged levels are chosen when the cut is built)
        h = tree.handle()
        if rnad_hip.bucket_plan(h, B) is None:
            pytest.skip("this table size does not fit the tree")
        logit = rnad_hip.mlp_forward(nets[0].pack(), 64, h.observations_table(), tree.max_actions, want_value=False)[0]
        ep = Episodes(tree, B, seed=31, lane_offset=9)
        ep.generate(nets[0], tabular=True, bucketed=True, trim=False, store_values=False, logits_table=logit, compact=True)
        return ep

    want = play({"RNAD_KEYS_GLOBAL": "1"})
    got = play({"RNAD_KEYS_LDS": "0", "RNAD_KEYS_STAGE_BYTES": str(stage_bytes)})
    assert torch.equal(got.lane_ids, want.lane_ids) and torch.equal(got.indices, want.indices)
    assert torch.equal(got._compact[0].acts, want._compact[0].acts) and torch.equal(got._compact[0].final_reward, want._compact[0].final_reward)
    assert torch.equal(got.alive, want.alive)


@pytest.mark.parametrize("name,rows", (("pruned", None), ("pruned", 12), ("a5c4", None), ("a5c4", 40), ("ternary4", None), ("ternary4", 20), ("binary", None)))
@pytest.mark.parametrize("keys_global", ("0", "1"))
def test_second_staging_level_plays_the_same_batch(name, rows, keys_global, monkeypatch):
    """An actor that leaves policy rows (rnad_mlp_forward_actor) is staged two levels deeper (rnad_bucket_stage_*): the upper rows, the roots
    of the group subtrees the lanes enter, the subtrees below the states the lanes are drawn into next.  Same batch bit for bit as the fully
    evaluated table gives, every visited row evaluated, and never more rows than the one-level staging (all rows of the non-empty groups)."""
    import rnad_hip
    from environment.episode import Episodes

    tree = _native_tree(**TREES[name])
    h = tree.handle()
    A, S, B = tree.max_actions, h.S, 4096
    monkeypatch.setenv("RNAD_KEYS_GLOBAL", keys_global)
    if rows is not None:
        monkeypatch.setenv("RNAD_BUCKET_ROWS", str(rows))
        if rnad_hip.bucket_plan(h, B) is None:
            pytest.skip(f"a table of {rows} rows does not fit this tree")
    nets = _four_nets(A, 64, seed=12)
    table = h.observations_table()
    packed = nets[0].pack()
    stride = int(rnad_hip.lib().rnad_bucket_policy_row_stride(A))

    def tables(fill):
        logit = torch.full((2 * S, A), fill, device=DEV)
        logit._policy_rows = torch.full((2 * S, stride), fill, device=DEV)
        return logit

    full_logit = tables(0.0)
    rnad_hip.mlp_forward_actor(h, packed, 64, table, full_logit, full_logit._policy_rows)
    full = Episodes(tree, B, seed=23, lane_offset=5)
    vis_full = torch.empty((2 * S,), dtype=torch.int32, device=DEV)
    full.generate(nets[0], tabular=True, bucketed=True, trim=False, store_values=False, logits_table=full_logit, compact=True, visited=vis_full)

    def play(levels):
        monkeypatch.setenv("RNAD_STAGE_LEVELS", levels)
        logit = tables(float("nan"))  # a row that was not evaluated would poison the rollout
        calls = []

        def actor(row_list):
            calls.append(int(row_list.count.item()))
            rnad_hip.mlp_forward_actor(h, packed, 64, table, logit, logit._policy_rows, rows=row_list)

        ep = Episodes(tree, B, seed=23, lane_offset=5)
        vis = torch.empty((2 * S,), dtype=torch.int32, device=DEV)
        ep.generate(nets[0], tabular=True, bucketed=True, trim=False, store_values=False, logits_table=logit, compact=True, visited=vis,
                    staged_actor=actor)
        assert torch.equal(ep.lane_ids, full.lane_ids) and torch.equal(ep.indices, full.indices)
        assert torch.equal(ep._compact[0].acts, full._compact[0].acts) and torch.equal(ep._compact[0].final_reward, full._compact[0].final_reward)
        assert torch.equal(ep.alive, full.alive) and torch.equal(vis, vis_full)
        seen = vis.bool()
        assert torch.isfinite(logit._policy_rows[seen]).all(), "every visited row must have been evaluated"
        assert torch.equal(logit._policy_rows[seen], full_logit._policy_rows[seen])
        return calls

    two = play("2")
    one = play("1")
    assert len(two) == 3 and len(one) == 2 and two[0] == one[0]
    assert two[1] + two[2] <= one[1], (two, one)


@pytest.mark.parametrize("name", ("pruned", "ternary4"))
def test_deferred_alive_counts_are_completed_by_the_learner_or_on_first_read(name):
    """Episodes.generate(defer_alive=True) leaves the per-step alive counts and the loss normalisers un-summed (one launch less per
    step).  The compact learner's own launch adds them up before its last kernel reads them; reading `alive` / `valid_counts` earlier
    completes them on the spot.  Same numbers, same gradient tables as the rollout that sums them itself."""
    import rnad_hip
    from environment.episode import Episodes

    tree = _native_tree(**TREES[name])
    h = tree.handle()
    A, B = tree.max_actions, 5000
    nets = _four_nets(A, 64, seed=3)
    logit, v, vt, lr, lr_ = _tables(tree, nets, A)
    hp = rnad_hip.make_learn_params(alpha=0.3, eta=0.2)
    rec, fast = rnad_hip.bucket_records(h, logit, v, vt, lr, lr_, hp, fast=True)
    actor = (rec, rnad_hip.policy_column(A))
    kw = dict(tabular=True, bucketed=True, trim=False, store_values=False, policy_table=actor, compact=True)
    plain = Episodes(tree, B, seed=4)
    plain.generate(nets[0], **kw)
    T = plain.t_eff + 1
    want = rnad_hip.learn_bucketed_compact(h, plain.buckets, plain._compact[0], T, rec, fast, plain.valid_counts, hp)
    want_alive, want_norm = plain.alive.clone(), plain.valid_counts.clone()
    # (a) the learner completes them
    lazy = Episodes(tree, B, seed=4)
    lazy.generate(nets[0], defer_alive=True, **kw)
    assert lazy.buckets.alive_pending is lazy._compact[0]
    got = rnad_hip.learn_bucketed_compact(h, lazy.buckets, lazy._compact[0], T, rec, fast, lazy.norm_for_learner(), hp)
    assert lazy.buckets.alive_pending is None
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    assert torch.equal(lazy.alive, want_alive) and torch.equal(lazy.valid_counts, want_norm)
    # (b) a reader that comes first
    early = Episodes(tree, B, seed=4)
    early.generate(nets[0], defer_alive=True, **kw)
    assert torch.equal(early.alive, want_alive) and early.buckets.alive_pending is None
    assert torch.equal(early.valid_counts, want_norm)
    # (c) trim=True reads the counters on the host: never deferred
    trimmed = Episodes(tree, B, seed=4)
    trimmed.generate(nets[0], defer_alive=True, **dict(kw, trim=True))
    assert trimmed.buckets.alive_pending is None and torch.equal(trimmed.alive, want_alive[: trimmed.t_eff + 2])
    # (d) two deferred rollouts of the same (tree, B) back to back: the partial counts live in the plan's scratch, which both share --
    # the second rollout first completes the batch that still waits for them
    first = Episodes(tree, B, seed=4)
    first.generate(nets[0], defer_alive=True, **kw)
    second = Episodes(tree, B, seed=5)
    second.generate(nets[0], defer_alive=True, **kw)
    assert first.buckets.alive_pending is None and second.buckets.alive_pending is second._compact[0]
    assert torch.equal(first.alive, want_alive) and torch.equal(first.valid_counts, want_norm)
    other = Episodes(tree, B, seed=5)
    other.generate(nets[0], **kw)
    assert torch.equal(second.alive, other.alive) and torch.equal(second.valid_counts, other.valid_counts)


@pytest.mark.parametrize("rows", (None, 40, 8))
@pytest.mark.parametrize("name", sorted(TREES))
@pytest.mark.parametrize("B", (8192, 3000))
@pytest.mark.parametrize("distinct", (False, True))
def test_rollout_and_learner_in_one_launch(name, B, rows, distinct, monkeypatch):
    """rnad_rollout_learn_bucketed_compact (k_bucket_play_learn: the workgroup of a work item plays its lanes and adds up their update)
    against rnad_rollout_bucketed_compact + rnad_learn_bucketed_compact: the same trajectory, alive counts, normalisers and per-row
    gradient tables, bit for bit -- at the planner's cut and at forced ones; also with the finish left to the caller (data parallel).
    distinct (RNAD_PLAY_LEARN_DISTINCT): larger work items, the learner once per distinct trajectory of an item weighted with the
    number of lanes that took it -- integer sums: the same bits."""
    import rnad_hip

    if distinct:
        monkeypatch.setenv("RNAD_FUSED_CHUNK", "512" if B == 3000 else "1024")
    elif rows == 40:
        monkeypatch.setenv("RNAD_BUCKET_CHUNK", "600")  # items of several passes: a thread reads back the lanes it played, pass by pass

    tree = _native_tree(**TREES[name])
    h = tree.handle()
    A = tree.max_actions
    if rows is not None:
        monkeypatch.setenv("RNAD_BUCKET_ROWS", str(rows))
        if rnad_hip.bucket_plan(h, B) is None:
            pytest.skip(f"a table of {rows} rows does not fit this tree")
    nets = _four_nets(A, 64, seed=6)
    logit, v, vt, lr, lr_ = _tables(tree, nets, A)
    hp = rnad_hip.make_learn_params(alpha=0.3, eta=0.2, w_v=0.7, w_n=1.3)
    rec, fast = rnad_hip.bucket_records(h, logit, v, vt, lr, lr_, hp, fast=True)
    T_cap = 2 * h.max_depth

    def traj():
        return rnad_hip.Trajectory(h, B, T_cap, DEV, with_observations=False, with_values=False, compact=True)

    two = traj()
    bk2 = rnad_hip.rollout_bucketed_compact(h, two, rec, seed=21, lane0=77)
    want = rnad_hip.learn_bucketed_compact(h, bk2, two, T_cap, rec, fast, bk2.norm, hp)
    one = traj()
    bk1, dlogit, dv = rnad_hip.rollout_learn_bucketed_compact(h, one, rec, fast, hp, seed=21, lane0=77, distinct=distinct)
    assert torch.equal(bk1.lane_ids, bk2.lane_ids)
    # (the work items may be larger -- the one launch runs its learner on the distinct trajectories of an item --: the same lanes, bucket by bucket)
    n1, n2 = int(bk1.n_items.item()), int(bk2.n_items.item())
    it1, it2 = bk1.items[:n1].cpu().numpy(), bk2.items[:n2].cpu().numpy()
    assert n1 <= n2 and it1[:, 1].sum() == it2[:, 1].sum() == B
    assert np.array_equal(np.bincount(it1[:, 2], weights=it1[:, 1]), np.bincount(it2[:, 2], weights=it2[:, 1]))
    assert torch.equal(one.alive, two.alive) and torch.equal(bk1.norm, bk2.norm)
    assert torch.equal(one.acts, two.acts) and torch.equal(one.final_reward.view(torch.int32), two.final_reward.view(torch.int32))
    assert torch.equal(one.indices, two.indices), "the states a lane went through (rebuilt from the relative states)"
    assert torch.equal(dlogit, want[0]) and torch.equal(dv, want[1])
    assert torch.isfinite(dlogit).all() and float(dlogit.abs().sum()) > 0
    # the finish left to the caller
    late = traj()
    bk3, none_l, none_v = rnad_hip.rollout_learn_bucketed_compact(h, late, rec, fast, hp, seed=21, lane0=77, norm_is_global=False, distinct=distinct)
    assert none_l is None and none_v is None
    dl3 = torch.empty_like(dlogit)
    dv3 = torch.empty_like(dv)
    rnad_hip.bucket_finish(h, bk3, bk3.norm, hp, dl3, dv3)
    assert torch.equal(dl3, want[0]) and torch.equal(dv3, want[1])
    # the normalisers of a global batch handed in (norm_global): the finish divides by those, the batch's own counts are still handed out
    g = (bk2.norm * 3).clone()
    part = traj()
    bk4, _, _ = rnad_hip.rollout_learn_bucketed_compact(h, part, rec, fast, hp, seed=21, lane0=77, norm_is_global=False, distinct=distinct)
    dl4, dv4 = torch.empty_like(dlogit), torch.empty_like(dv)
    rnad_hip.bucket_finish(h, bk4, g, hp, dl4, dv4)
    glob = traj()
    bk5, dl5, dv5 = rnad_hip.rollout_learn_bucketed_compact(h, glob, rec, fast, hp, seed=21, lane0=77, norm_global=g, distinct=distinct)
    assert torch.equal(bk5.norm, bk2.norm) and torch.equal(glob.alive, two.alive)
    assert torch.equal(dl5, dl4) and torch.equal(dv5, dv4)
