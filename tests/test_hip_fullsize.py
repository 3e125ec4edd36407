"""BASELINE.json-sized runs on the GPU, checked through size-independent properties and against the oracle on sub-sampled
lanes (episodes are independent, so any subset of lanes must reproduce exactly)."""
import numpy as np
import pytest
import torch

from _util import assert_bits_equal, load

pytestmark = pytest.mark.gpu


def _arrays(tree):
    return dict(index=tree.index_tensor.cpu().numpy(), value=tree.value_tensor.cpu().numpy(), chance=tree.chance_tensor.cpu().numpy(),
                expected_value=tree.expected_value_tensor.cpu().numpy(), legal=tree.legal_tensor.cpu().numpy())


def _check_lanes_against_oracle(ep, arrs, lanes, seed, C, half=False, lane_ids=None):
    """Given the policy bits the GPU produced and the seeded uniforms (include/rnad_rng.h), every recorded step of `lanes` must be what the oracle computes.
    lane_ids (bucket-ordered batches): `lanes` are COLUMNS of the buffers, column j holds lane lane_ids[j] -- the id its draws are keyed by."""
    from oracle import oracle

    T = ep.t_eff + 1
    lanes_t = torch.as_tensor(lanes, device=ep.indices.device)
    columns = lanes
    if lane_ids is not None:
        lanes = lane_ids[lanes_t].cpu().numpy()
    idx = ep.indices[:, lanes_t].cpu().numpy().astype(np.int64)
    act = ep.action_idx[:, lanes_t].cpu().numpy().astype(np.int64)
    pol = ep.policy[:, lanes_t].cpu().numpy()
    obs = ep.observations[:, lanes_t].cpu().numpy()
    rew = ep.rewards[:, lanes_t].cpu().numpy()
    nxt_last = ep.states.indices[lanes_t].cpu().numpy()
    A = pol.shape[-1]
    n = len(columns)
    for t in range(T):
        want_obs, want_mask = oracle.observe(arrs["expected_value"], arrs["legal"], idx[t], np.full(n, t & 1))
        assert_bits_equal(obs[t], want_obs.astype(np.float16) if half else want_obs, f"observe t={t}")
        u3 = np.stack([oracle.uniforms(1, seed, int(b), t)[0] for b in lanes])  # [n, 3]: row action, column action, chance
        drawn = oracle.pick(pol[t], np.ascontiguousarray(u3[:, t & 1]))
        if lane_ids is not None:  # a compact batch stops drawing once a lane is absorbed (its slots show action 0; nothing reads them)
            drawn = np.where(idx[t] != 0, drawn, 0)
        np.testing.assert_array_equal(drawn, act[t], err_msg=f"sample t={t}")
        assert ((pol[t] > 0) == (want_mask > 0)).all()
        if t & 1:
            nxt, r = oracle.transition(arrs["index"], arrs["chance"], arrs["value"], idx[t], act[t - 1], act[t], np.ascontiguousarray(u3[:, 2]))
            np.testing.assert_array_equal(nxt, idx[t + 1] if t + 1 < T else nxt_last)
            assert_bits_equal(rew[t], r, f"reward t={t}")
        else:
            assert (rew[t] == 0).all()
            if t + 1 < T:
                np.testing.assert_array_equal(idx[t + 1], idx[t])


@pytest.fixture(scope="module")
def c2():
    """BASELINE.json configs[1]: depth-6 ternary tree, batch 2^20, MLP width 256."""
    from environment.episode import Episodes
    from environment.tree import Tree
    from nn.net import MLP

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    tree = Tree(device=dev, max_actions=3, max_transitions=1, depth_bound=6)
    tree.generate_native(seed=0)
    net = MLP(3, 256, device=dev)
    ep = Episodes(tree, 1 << 20, seed=99)
    ep.generate(net)
    return tree, net, ep


def test_c2_rollout_properties_and_sampled_lanes(c2):
    tree, net, ep = c2
    B = 1 << 20
    assert tree.index_tensor.shape[0] == 66431 and ep.t_eff + 1 == 12
    alive = ep.alive.cpu().numpy()
    assert (alive[:12] == B).all() and alive[12] == 0  # regular tree: every episode lasts exactly 2 * depth steps
    assert (ep.rewards[:11] == 0).all() and (ep.rewards[11].abs() == 1).all()  # +-1 terminal payoffs on the last column step
    assert (ep.indices[0] == 1).all() and (ep.states.indices == 0).all()
    depth_of = lambda idx: idx  # noqa: E731
    s = ep.policy.sum(-1)
    assert torch.allclose(s, torch.ones_like(s), atol=1e-5)
    assert (ep.mask_bits == 7).all()
    lanes = np.random.default_rng(0).choice(B, size=2048, replace=False)
    lanes[:4] = (0, 1, B - 1, B // 2)
    _check_lanes_against_oracle(ep, _arrays(tree), lanes, seed=99, C=1)
    # rollouts are reproducible and lane-addressed: a 2-way split with lane offsets replays the same episodes
    from environment.episode import Episodes

    half = B // 2
    shard = Episodes(tree, half, seed=99, lane_offset=half)
    shard.generate(net)
    assert torch.equal(shard.indices, ep.indices[:, half:]) and torch.equal(shard.action_idx, ep.action_idx[:, half:])
    assert torch.equal(shard.rewards, ep.rewards[:, half:])


def test_c2_learner_is_lane_independent_and_matches_oracle_on_a_subset(c2):
    """rnad_learn_fused at 12.6 M (t, b): any subset of lanes run on its own gives the same bits; the oracle composition
    (policy head -> process_policy -> v_trace x2 -> losses) agrees on that subset."""
    import rnad_hip
    from nn.net import MLP
    from oracle import oracle

    tree, net, ep = c2
    dev = ep.indices.device
    T, B, A = 12, 1 << 20, 3
    torch.manual_seed(1)
    nets = [net] + [MLP(3, 256, device=dev) for _ in range(3)]
    with torch.no_grad():
        logit, v = nets[0].forward_logits(ep.observations)
        _, vt = nets[1].forward_logits(ep.observations, want_logits=False)
        lr, _ = nets[2].forward_logits(ep.observations, want_value=False)
        lr_, _ = nets[3].forward_logits(ep.observations, want_value=False)
    norm = ep.valid_counts
    assert norm.tolist() == [6.0 * B, 6.0 * B]
    hp = rnad_hip.make_learn_params(alpha=0.3, eta=0.2)
    full = rnad_hip.learn_fused(ep.indices, ep.mask_bits, ep.action_idx, ep.rewards, ep.policy, logit, v.view(T, B), vt.view(T, B), lr, lr_, norm, hp)
    lanes = torch.as_tensor(np.random.default_rng(1).choice(B, size=4096, replace=False), device=dev)
    sub = lambda x: x.view(T, B, -1)[:, lanes].contiguous()  # noqa: E731
    sub2 = lambda x: x.view(T, B)[:, lanes].contiguous()  # noqa: E731
    part = rnad_hip.learn_fused(sub2(ep.indices), sub2(ep.mask_bits), sub2(ep.action_idx), sub2(ep.rewards), sub(ep.policy), sub(logit),
                                sub2(v), sub2(vt), sub(lr), sub(lr_), norm, hp, want_aux=True)
    assert torch.equal(part[0], full[0][:, lanes]) and torch.equal(part[1], full[1][:, lanes])
    # oracle composition on the subset, fed with the GPU's own pi bits where the reference's functions are discontinuous
    n = lambda t: t.cpu().numpy()  # noqa: E731
    masks = np.ones((T, 4096, A), np.float32)
    pi = n(part[3])
    pip = oracle.process_policy(pi, masks, 32, 0.03)
    _, log_pi = oracle.policy_head(n(sub(logit)), masks)
    _, log_r = oracle.policy_head(n(sub(lr)), masks)
    _, log_r_ = oracle.policy_head(n(sub(lr_)), masks)
    lpol = log_pi - (np.float32(0.3) * log_r + np.float32(1 - 0.3) * log_r_)
    valid = (n(sub2(ep.indices)) != 0).astype(np.float32)
    turns = np.broadcast_to((np.arange(T) % 2)[:, None], (T, 4096)).astype(np.int64)
    a_oh = np.eye(A, dtype=np.float32)[n(sub2(ep.action_idx))]
    for p in range(2):
        rew = n(sub2(ep.rewards)) * (1 if p == 0 else -1)
        vt_p, _, q_p = oracle.vtrace(n(sub2(vt))[..., None], valid, turns, n(sub(ep.policy)), pip, lpol, a_oh, rew, p, 0.2, 1.0, 1.0, 1.0, 1.0)
        np.testing.assert_allclose(n(part[4][p]), vt_p[..., 0], rtol=1e-5, atol=5e-6)
        np.testing.assert_allclose(n(part[5][p]), q_p, rtol=1e-5, atol=5e-6)


def test_c2_update_step_runs_and_is_finite(c2):
    import os
    import tempfile

    from environment.episode import Buffer
    from learn.rnad import RNaD

    tree, _, _ = c2
    os.environ["RNAD_SAVE_DIR"] = tempfile.mkdtemp(prefix="rnad_test_")
    rn = RNaD(tree=tree, device=tree.device, directory_name="full", batch_size=1 << 20, eta=0.2, b1_adam=0.0, lr=1e-3)
    rn.initialize()
    before = [p.detach().clone() for p in rn.net.parameters()]
    log = {}
    rn.train_step(Buffer(1), alpha=0.5, log=log)
    assert all(torch.isfinite(p).all() for p in rn.net.parameters())
    assert any(not torch.equal(a, b) for a, b in zip(before, rn.net.parameters()))
    assert np.isfinite(log["loss_v"]) and np.isfinite(log["loss_nerd"]) and log["traj_len"] == 12
    # target net moved by gamma_averaging * (net - target)
    for pt, pn, p0 in zip(rn.net_target.parameters(), rn.net.parameters(), before):
        np.testing.assert_allclose(pt.detach().cpu().numpy(), (0.001 * pn + 0.999 * p0).detach().cpu().numpy(), rtol=1e-5, atol=1e-7)


def _modes_agree(tree, batch, tmp_path, obs_half=False, width=256, modes=(False, "forward", True)):
    """One RNaD.train_step from the same weights and rollout seed in the three net-evaluation modes (eager, no graph): dense and
    "forward" give identical gradients; the default (per-row sums, bucketed rollout) gives them up to fp32 summation order."""
    import os

    from environment.episode import Buffer
    from learn.rnad import RNaD

    os.environ["RNAD_SAVE_DIR"] = str(tmp_path)
    grads, losses = {}, {}
    for mode in modes:
        torch.manual_seed(11)
        rn = RNaD(tree=tree, device=tree.device, directory_name=f"modes{mode}{int(obs_half)}", batch_size=batch, eta=0.2, b1_adam=0.0, lr=1e-3,
                  net_params={"type": "MLP", "max_actions": tree.max_actions, "width": width})
        rn.initialize()
        rn.tabular, rn.use_graph, rn.obs_half = mode, False, obs_half
        with torch.no_grad():
            for p in rn.net_reg_.parameters():
                p.mul_(1.01)
        rn.fused_optimizer = False  # the gradients are read where torch's optimizer.step() would be called
        captured = {}
        real = rn.optimizer.step
        rn.optimizer.step = lambda: (captured.update(g=[p.grad.detach().clone() for p in rn.net.parameters()]), real())[1]
        rn.train_step(Buffer(1), alpha=0.4)
        torch.cuda.synchronize()
        grads[mode] = captured["g"]
        import rnad_hip

        assert (rn.last_episodes.buckets is not None) == (mode is True and rnad_hip.bucket_plan(tree.handle(), batch) is not None)
        assert all(torch.isfinite(p).all() for p in rn.net.parameters())
        del rn
        torch.cuda.empty_cache()
    if False in modes:
        for a, b in zip(grads[False], grads["forward"]):
            assert torch.equal(a, b)
    # (the per-row mode evaluates the tables with other kernels -- legal fold, csrc/mlp_rows.hip -- whose logits differ from the per-slot
    # kernels' in the last bit: of the ~5e7 inverse-CDF draws of a 2^22 batch a handful then fall on the other side of a boundary, i.e. the
    # two modes learn from batches that differ in a few episodes.  Achieved: ~1.1e-5 of the largest entry.  r05 measured the flips directly,
    # against the CPU port (tests/test_hip_e2e.py, profiles/r05_e2e_stats.jsonl): 1.06 lanes per 10^7 decisions, i.e. ~8 of the 2^22 lanes
    # of this test's largest batch -- a flipped lane moves a gradient entry by about its own share of the sum, 12 slots of 5e7, which is
    # what the 2.5e-5 of the largest entry below allows for; the comparison that does NOT depend on which side of a boundary a draw falls
    # is the one against the port from the same weights, where the parameters agree to 9.2e-7 after eight steps)
    for a, b in zip(grads["forward"], grads[True]):
        scale = a.abs().max().item() + 1e-12
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-4, atol=2.5e-5 * scale)


def test_c2_tree_at_2_to_22_lanes_on_one_gpu_keeps_the_default_mode(c2, tmp_path):
    """The single-GPU point of BASELINE.json configs[2] (one 2^22 batch): the bucketed learner takes 2^22 lanes per call, so the
    default mode stays in effect (no fall-back to "forward" above 2^21) and gives the per-slot mode's gradients."""
    import os

    from environment.episode import Buffer
    from learn.rnad import RNaD

    tree, _, _ = c2
    B = 1 << 22
    os.environ["RNAD_SAVE_DIR"] = str(tmp_path)
    probe = RNaD(tree=tree, device=tree.device, directory_name="probe22", batch_size=B, eta=0.2, b1_adam=0.0, lr=1e-3)
    probe.initialize()
    assert probe._tabular_mode(12, B) is True
    buf = Buffer(1)
    for i in range(5):  # three eager steps, the capture, one replay
        probe.train_step(buf, alpha=0.5)
        probe.total_steps += 1
    torch.cuda.synchronize()
    assert probe._graph["graph"] is not None and not probe._graph["failed"]
    assert probe.last_episodes._compact is not None and probe.last_episodes.batch_size == B
    assert probe.last_episodes.alive.tolist() == [B] * 12 + [0]
    assert all(torch.isfinite(p).all() for p in probe.net.parameters())
    del probe, buf
    torch.cuda.empty_cache()
    _modes_agree(tree, B, tmp_path, modes=("forward", True))


def test_c2_fp32_update_modes_agree_full_size(c2, tmp_path):
    """configs[1] at its full size in fp32: the default step (compact bucketed rollout, k_bucket_learn, one backward over the rows)
    against the dense and "forward" modes from the same weights and rollout seed."""
    tree, _, _ = c2
    _modes_agree(tree, 1 << 20, tmp_path)


def _compact_rollout_is_the_lane_ordered_one(tree, net, ep, seed, C, rng_seed):
    """The rollout the default step runs (k_bucket_keys -> sort -> k_bucket_rollout_compact, acting on the pi columns of the row
    records) at 2^20 lanes: the same episodes as the lane-ordered rollout `ep`, column j holding lane lane_ids[j]; its expanded
    fields (rnad_bucket_expand) against the oracle on sampled columns."""
    import rnad_hip
    from environment.episode import Episodes

    h, A, B = tree.handle(), tree.max_actions, ep.batch_size
    assert rnad_hip.bucket_plan(h, B) is not None
    with torch.no_grad():
        logit, v = rnad_hip.mlp_forward(net.pack(), net.width, h.observations_table(), A)
    hp = rnad_hip.make_learn_params(alpha=0.5, eta=0.2)
    rec, fast = rnad_hip.bucket_records(h, logit, v, v, logit, logit, hp, fast=True)
    comp = Episodes(tree, B, seed=seed)
    comp.generate(net, tabular=True, bucketed=True, store_values=False, policy_table=(rec, rnad_hip.policy_column(A)), compact=True)
    assert comp._compact is not None and comp._compact[0].policy is None and comp.t_eff == ep.t_eff
    perm = comp.lane_ids.long()
    assert torch.equal(torch.sort(perm).values, torch.arange(B, device=perm.device))
    assert torch.equal(comp.indices, ep.indices[:, perm]) and torch.equal(comp.alive, ep.alive)
    assert torch.equal(comp.states.indices, ep.states.indices[perm])
    live = comp.indices != 0
    # the dense fields, written by rnad_bucket_expand on first access
    assert torch.equal(comp.policy, ep.policy[:, perm]) and torch.equal(comp.rewards, ep.rewards[:, perm])
    assert torch.equal(comp.mask_bits, ep.mask_bits[:, perm])
    assert torch.equal(comp.action_idx[live], ep.action_idx[:, perm][live]) and (comp.action_idx[~live] == 0).all()
    cols = np.random.default_rng(rng_seed).choice(B, size=2048, replace=False)
    cols[:3] = (0, B - 1, B // 2)
    _check_lanes_against_oracle(comp, _arrays(tree), cols, seed=seed, C=C, lane_ids=comp.lane_ids)
    return comp, rec, fast, hp


def test_c2_compact_bucketed_rollout_full_size(c2):
    tree, net, ep = c2
    comp, rec, fast, hp = _compact_rollout_is_the_lane_ordered_one(tree, net, ep, seed=99, C=1, rng_seed=7)
    # ... and the learner on it: the compact variant's tables are the dense-record variant's, bit for bit, at 12.6 M slots
    import rnad_hip

    h, T = tree.handle(), comp.t_eff + 1
    got = rnad_hip.learn_bucketed_compact(h, comp.buckets, comp._compact[0], T, rec, fast, comp.valid_counts, hp)
    want = rnad_hip.learn_bucketed(h, comp.buckets, comp.indices, comp.action_idx, comp.rewards, comp.policy, rec, comp.valid_counts, hp)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    assert torch.isfinite(got[0]).all() and float(got[0].abs().sum()) > 0


def test_c2_fp16_observations_full_size(c2, tmp_path):
    """The single-GPU half of BASELINE.json configs[4]: c2 tree, batch 2^20, observations stored as fp16, arithmetic fp32."""
    from environment.episode import Episodes

    tree, net, ep32 = c2
    B = 1 << 20
    ep = Episodes(tree, B, seed=99, obs_half=True)
    ep.generate(net, tabular=False)  # every lane through K1 (fp16 stores) and the fused MLP (fp16 loads)
    assert ep.observations.dtype == torch.float16 and ep.t_eff + 1 == 12
    lanes = np.random.default_rng(3).choice(B, size=2048, replace=False)
    _check_lanes_against_oracle(ep, _arrays(tree), lanes, seed=99, C=1, half=True)
    # the tabular actor on the fp16-rounded observation table plays the same episodes
    tab = Episodes(tree, B, seed=99, obs_half=True)
    tab.generate(net, tabular=True)
    for key in ("indices", "policy", "action_idx", "rewards", "observations"):
        assert torch.equal(getattr(tab, key), getattr(ep, key)), key
    del ep, tab
    torch.cuda.empty_cache()
    _modes_agree(tree, B, tmp_path, obs_half=True)


@pytest.fixture(scope="module")
def c4():
    """BASELINE.json configs[3]: depth-8 5x5 tree, chance branching 4, pruned (reference main.py:37) to ~10^6 states, batch 2^20."""
    from environment.episode import Episodes
    from environment.tree import Tree
    from nn.net import MLP

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    tree = Tree(device=dev, max_actions=5, max_transitions=4, depth_bound=8, transition_threshold=0.1)
    tree.generate_native(seed=0, prune=(7, 8))
    net = MLP(5, 256, device=dev)
    ep = Episodes(tree, 1 << 20, seed=41)
    ep.generate(net)
    return tree, net, ep


def test_c4_full_size_rollout(c4):
    import rnad_hip
    from environment.episode import Episodes

    tree, net, ep = c4
    B, S = 1 << 20, tree.index_tensor.shape[0]
    assert S == 959540 and 8 * S <= 16 * B  # the tabular actor applies
    T = ep.t_eff + 1
    assert T % 2 == 0 and 6 <= T <= 16
    alive = ep.alive.cpu().numpy()
    assert alive[0] == B and (np.diff(alive) <= 0).all() and alive[T] == 0 and alive[T - 1] > 0
    np.testing.assert_array_equal(alive[:T], (ep.indices != 0).sum(1).cpu().numpy())
    assert alive[:T].sum() < 0.6 * T * B  # ragged: well under half of the slots of the padded buffer are live
    assert (ep.rewards[0::2] == 0).all()
    s = ep.policy.sum(-1)
    assert torch.allclose(s, torch.ones_like(s), atol=1e-5)
    lanes = np.random.default_rng(4).choice(B, size=2048, replace=False)
    lanes[:3] = (0, B - 1, B // 3)
    _check_lanes_against_oracle(ep, _arrays(tree), lanes, seed=41, C=4)
    # every lane through the net at every step (the reference's way) plays the same episodes
    dense = Episodes(tree, B, seed=41)
    dense.generate(net, tabular=False)
    for key in ("indices", "action_idx", "rewards", "mask_bits"):
        assert torch.equal(getattr(dense, key), getattr(ep, key)), key
    assert torch.equal(dense.policy, ep.policy)
    del dense
    # and the bucket-ordered rollout is the same batch, permuted
    if rnad_hip.bucket_plan(tree.handle(), B) is not None:
        buc = Episodes(tree, B, seed=41)
        buc.generate(net, bucketed=True)
        perm = buc.lane_ids.long()
        for key in ("indices", "action_idx", "rewards", "mask_bits", "policy"):
            assert torch.equal(getattr(buc, key), getattr(ep, key)[:, perm]), key
        assert torch.equal(buc.alive, ep.alive)


def test_c4_compact_bucketed_rollout_full_size(c4):
    tree, net, ep = c4
    if __import__("rnad_hip").bucket_plan(tree.handle(), 1 << 20) is None:
        pytest.skip("this tree cannot be bucketed")
    _compact_rollout_is_the_lane_ordered_one(tree, net, ep, seed=41, C=4, rng_seed=8)


def test_c4_full_size_learner_is_lane_independent(c4):
    """rnad_learn_fused at 2^20 lanes x T (A = 5): a subset of lanes run on its own gives the same bits, and the oracle's
    composition of the reference functions agrees on it."""
    import rnad_hip
    from nn.net import MLP
    from oracle import oracle

    tree, net, ep = c4
    dev = ep.indices.device
    T, B, A = ep.t_eff + 1, 1 << 20, 5
    torch.manual_seed(1)
    nets = [net] + [MLP(5, 256, device=dev) for _ in range(3)]
    with torch.no_grad():
        logit, v = nets[0].forward_logits(ep.observations)
        _, vt = nets[1].forward_logits(ep.observations, want_logits=False)
        lr, _ = nets[2].forward_logits(ep.observations, want_value=False)
        lr_, _ = nets[3].forward_logits(ep.observations, want_value=False)
    norm = ep.valid_counts
    hp = rnad_hip.make_learn_params(alpha=0.3, eta=0.2)
    full = rnad_hip.learn_fused(ep.indices, ep.mask_bits, ep.action_idx, ep.rewards, ep.policy, logit, v.view(T, B), vt.view(T, B), lr, lr_, norm, hp)
    n = 2048
    lanes = torch.as_tensor(np.random.default_rng(5).choice(B, size=n, replace=False), device=dev)
    sub = lambda x: x.view(T, B, -1)[:, lanes].contiguous()  # noqa: E731
    sub2 = lambda x: x.view(T, B)[:, lanes].contiguous()  # noqa: E731
    part = rnad_hip.learn_fused(sub2(ep.indices), sub2(ep.mask_bits), sub2(ep.action_idx), sub2(ep.rewards), sub(ep.policy), sub(logit),
                                sub2(v), sub2(vt), sub(lr), sub(lr_), norm, hp, want_aux=True)
    assert torch.equal(part[0], full[0][:, lanes]) and torch.equal(part[1], full[1][:, lanes])
    c = lambda t: t.cpu().numpy()  # noqa: E731
    masks = c(ep.masks[:, lanes])
    pi = c(part[3])
    pip = oracle.process_policy(pi, masks, 32, 0.03)
    _, log_pi = oracle.policy_head(c(sub(logit)), masks)
    _, log_r = oracle.policy_head(c(sub(lr)), masks)
    _, log_r_ = oracle.policy_head(c(sub(lr_)), masks)
    lpol = log_pi - (np.float32(0.3) * log_r + np.float32(1 - 0.3) * log_r_)
    valid = (c(sub2(ep.indices)) != 0).astype(np.float32)
    turns = np.broadcast_to((np.arange(T) % 2)[:, None], (T, n)).astype(np.int64)
    a_oh = np.eye(A, dtype=np.float32)[c(sub2(ep.action_idx))]
    for p in range(2):
        rew = c(sub2(ep.rewards)) * (1 if p == 0 else -1)
        vt_p, _, q_p = oracle.vtrace(c(sub2(vt))[..., None], valid, turns, c(sub(ep.policy)), pip, lpol, a_oh, rew, p, 0.2, 1.0, 1.0, 1.0, 1.0)
        np.testing.assert_allclose(c(part[4][p]), vt_p[..., 0], rtol=1e-5, atol=5e-6)
        np.testing.assert_allclose(c(part[5][p]), q_p, rtol=1e-5, atol=5e-6)


def test_c4_full_size_update_modes_agree(c4, tmp_path):
    tree, _, _ = c4
    _modes_agree(tree, 1 << 20, tmp_path)


@pytest.mark.parametrize("half", (False, True))
def test_c4_like_tree_five_actions_four_chance_outcomes(half):
    """BASELINE.json configs[3] shape (5x5 actions, chance branching 4, pruned) at a size the oracle finishes in seconds;
    half=True is configs[4]'s fp16 observation buffer."""
    from environment.episode import Buffer, Episodes
    from environment.tree import Tree
    from learn.rnad import RNaD
    from nn.net import MLP

    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    tree = Tree(device=dev, max_actions=5, max_transitions=4, depth_bound=4, transition_threshold=0.2)
    tree.generate_native(seed=11, prune=(1, 2))
    tree.assert_index_is_tree()
    S = tree.index_tensor.shape[0]
    assert S > 2000
    B, seed = 1 << 15, 5
    net = MLP(5, 128, device=dev)
    ep = Episodes(tree, B, seed=seed, obs_half=half)
    ep.generate(net)
    T = ep.t_eff + 1
    assert 2 <= T <= 8 and T % 2 == 0
    alive = ep.alive.cpu().numpy()
    assert alive[0] == B and (np.diff(alive) <= 0).all() and alive[T] == 0
    np.testing.assert_array_equal(alive[:T], (ep.indices != 0).sum(1).cpu().numpy())
    lanes = np.random.default_rng(2).choice(B, size=1024, replace=False)
    _check_lanes_against_oracle(ep, _arrays(tree), lanes, seed=seed, C=4, half=half)
    if not half:
        import os
        import tempfile

        os.environ["RNAD_SAVE_DIR"] = tempfile.mkdtemp(prefix="rnad_test_")
        rn = RNaD(tree=tree, device=dev, directory_name="c4", batch_size=B, eta=0.2, b1_adam=0.0, lr=1e-3,
                  net_params={"type": "MLP", "max_actions": 5, "width": 128})
        rn.initialize()
        rn.train_step(Buffer(1), alpha=1.0, log={})
        assert all(torch.isfinite(p).all() for p in rn.net.parameters())


def test_short_run_follows_the_reference_schedule(tmp_path, monkeypatch):
    """RNaD.run() with the fixture's settings (2 regularisation updates x 3 steps on the c1 tree): same (m, n, alpha)
    sequence, same net_reg / net_reg_ rotation and EMA bookkeeping as the reference's recorded run."""
    from _gpu import DEV, golden_tree
    from learn.rnad import RNaD

    g = load("run_c1")
    tree, _ = golden_tree("c1")
    monkeypatch.setenv("RNAD_SAVE_DIR", str(tmp_path))
    torch.manual_seed(21)
    rn = RNaD(tree=tree, device=DEV, directory_name="golden", batch_size=int(g["batch"]), eta=0.2, bounds=[2], delta_m=[3], lr=1e-2,
              gamma_averaging=0.1, b1_adam=0.0, net_params={"type": "MLP", "max_actions": 2, "width": 16})
    seen = []
    step = rn.train_step

    def spy(buffer, alpha, log=None):
        nets = {k: [p.detach().clone() for p in getattr(rn, k).parameters()] for k in ("net", "net_target", "net_reg", "net_reg_")}
        step(buffer, alpha, log=log)
        seen.append(dict(m=rn.m, n=rn.n, alpha=alpha, before=nets,
                         after={k: [p.detach().clone() for p in getattr(rn, k).parameters()] for k in nets}))

    rn.train_step = spy
    rn.run(checkpoint_mod=1, expl_mod=1, log_mod=10**9)
    assert len(seen) == int(g["n_steps"]) == 6
    for i, st in enumerate(seen):
        assert (st["m"], st["n"]) == tuple(g[f"s{i}_mn"]) and st["alpha"] == pytest.approx(float(g[f"s{i}_alpha"]))
        for pt, pn, p0 in zip(st["after"]["net_target"], st["after"]["net"], st["before"]["net_target"]):  # rnad.py:516-523
            assert torch.allclose(pt, 0.1 * pn + 0.9 * p0, rtol=1e-5, atol=1e-7)
        for a, b in zip(st["before"]["net_reg"], st["after"]["net_reg"]):  # regularisation nets only move between updates
            assert torch.equal(a, b)
    # rotation at the end of m = 0 (rnad.py:528-531): net_reg_ <- net_reg (still the initial net), net_reg <- net_target
    for a, b in zip(seen[3]["before"]["net_reg_"], seen[0]["before"]["net_reg"]):
        assert torch.equal(a, b)
    for a, b in zip(seen[3]["before"]["net_reg"], seen[2]["after"]["net_target"]):
        assert torch.equal(a, b)
    assert rn.m == 2 and rn.n == 0 and rn.total_steps == 6
    assert len(rn.nashconv_history) == 1 and 0 <= rn.nashconv_history[0][2] <= 2.0  # NashConv logged at m = 1 (rnad.py:490)
    # checkpoints in the reference's layout: saved_runs/<dir>/params and <m>/<n>
    assert (tmp_path / "saved_runs" / "golden" / "params").exists()
    assert sorted(p.name for p in (tmp_path / "saved_runs" / "golden" / "1").iterdir()) == ["0", "1", "2"]
    ck = torch.load(tmp_path / "saved_runs" / "golden" / "1" / "2", weights_only=False)
    assert sorted(ck) == ["net", "net_params", "net_reg", "net_reg_", "net_target", "optimizer", "total_steps"]
    # resuming picks up where the run stopped (rnad.py:243-272)
    rn2 = RNaD(tree=tree, device=DEV, directory_name="golden", batch_size=int(g["batch"]), bounds=[3], delta_m=[3], b1_adam=0.0,
               net_params={"type": "MLP", "max_actions": 2, "width": 16})
    rn2.initialize()
    assert (rn2.m, rn2.n) == (1, 2) and rn2.eta == 0.2 and rn2.lr == 1e-2


def test_on_policy_shortcut_gives_the_same_update(tmp_path, monkeypatch):
    """RNaD.reuse_actor_outputs: the rollout's logits / values ARE the learner's forward outputs -> identical parameters."""
    from environment.episode import Buffer
    from environment.tree import Tree
    from learn.rnad import RNaD

    dev = torch.device("cuda:0")
    monkeypatch.setenv("RNAD_SAVE_DIR", str(tmp_path))
    tree = Tree(device=dev, max_actions=3, max_transitions=2, depth_bound=3, transition_threshold=0.2)
    tree.generate_native(seed=8)
    out = []
    for reuse in (False, True):
        torch.manual_seed(5)
        rn = RNaD(tree=tree, device=dev, directory_name=f"reuse{int(reuse)}", batch_size=8192, eta=0.2, b1_adam=0.0, lr=1e-3,
                  net_params={"type": "MLP", "max_actions": 3, "width": 64})
        rn.initialize()
        rn.reuse_actor_outputs = reuse
        rn.tabular = False  # the shortcut replaces the learner's per-slot forward: compare it with the per-slot (dense) mode
        buf = Buffer(1)
        for i in range(3):
            rn.train_step(buf, alpha=0.3 * i)
            rn.total_steps += 1
        assert (rn.last_episodes.actor_logits is not None) == reuse
        out.append([p.detach().clone() for p in rn.net.parameters()] + [p.detach().clone() for p in rn.net_target.parameters()])
    for a, b in zip(*out):
        assert torch.equal(a, b)


def test_run_replays_the_reference_run_given_its_noise(tmp_path, monkeypatch):
    """The reference's recorded 6-step RNaD.run (tests/golden/run_c1.npz): same initial weights + the Exp(1) noise it consumed
    -> the same rollouts, the same gradients and the same nets after Adam / EMA / rotation, step by step."""
    import environment.episode as episode
    from _gpu import DEV, golden_tree, gpu
    from learn.rnad import RNaD

    g = load("run_c1")
    tree, _ = golden_tree("c1")
    monkeypatch.setenv("RNAD_SAVE_DIR", str(tmp_path))
    keys = ("value_fc0.weight", "value_fc0.bias", "value_fc1.weight", "value_fc1.bias", "policy_fc0.weight", "policy_fc0.bias",
            "policy_fc1.weight", "policy_fc1.bias")
    sd = lambda prefix: {k: gpu(g[prefix + k.replace(".", "_")]) for k in keys}  # noqa: E731
    rn = RNaD(tree=tree, device=DEV, directory_name="replay", batch_size=int(g["batch"]), eta=0.2, bounds=[2], delta_m=[3], lr=1e-2,
              gamma_averaging=0.1, b1_adam=0.0, net_params={"type": "MLP", "max_actions": 2, "width": 16})
    rn.initialize()
    rn.use_graph = False  # Episodes.generate is patched below to inject the reference's noise: nothing to capture
    for n in (rn.net, rn.net_target, rn.net_reg, rn.net_reg_):  # the reference's initial net (all four start equal, rnad.py:226-231)
        n.load_state_dict(sd("w0_"))
    step = {"i": 0}
    real_generate = episode.Episodes.generate

    def generate_with_reference_noise(self, net, **kw):
        i = step["i"]
        real_generate(self, net, noise_action=gpu(g[f"s{i}_noise_action"]), noise_chance=gpu(g[f"s{i}_noise_chance"]), trim=False)
        T = g[f"s{i}_indices"].shape[0]
        same = (self.indices[:T].cpu().numpy() == g[f"s{i}_indices"]).all(0) & (self.action_idx[:T].cpu().numpy() == g[f"s{i}_actions"].argmax(-1)).all(0)
        step.setdefault("same", []).append(same.mean())
        if i == 0:  # identical weights in -> the identical episode batch out
            assert same.all()
            assert_bits_equal(self.rewards[:T].cpu().numpy(), g["s0_rewards"], "rewards")

    monkeypatch.setattr(episode.Episodes, "generate", generate_with_reference_noise)
    real_step = rn.train_step

    def checked_step(buffer, alpha, log=None):
        i = step["i"]
        assert (rn.m, rn.n) == tuple(g[f"s{i}_mn"]) and alpha == pytest.approx(float(g[f"s{i}_alpha"]))
        real_step(buffer, alpha, log=log)
        for tag, net in (("net", rn.net), ("net_target", rn.net_target), ("net_reg", rn.net_reg), ("net_reg_", rn.net_reg_)):
            for k, p in net.state_dict().items():
                want = g[f"s{i}_{tag}_" + k.replace(".", "_")]
                close = np.isclose(p.detach().cpu().numpy(), want, rtol=1e-4, atol=2e-5)
                assert close.mean() >= (1.0 if i == 0 else 0.97), (i, tag, k, close.mean())
        step["i"] += 1

    rn.train_step = checked_step
    rn._RNaD__resume(checkpoint_mod=10**9, expl_mod=10**9, log_mod=10**9)
    assert step["i"] == 6 and min(step["same"]) > 0.9
    for tag, net in (("net", rn.net), ("target", rn.net_target), ("reg", rn.net_reg), ("reg_", rn.net_reg_)):  # after the last rotation
        for k, p in net.state_dict().items():
            close = np.isclose(p.detach().cpu().numpy(), g[f"final_{tag}_" + k.replace(".", "_")], rtol=1e-4, atol=2e-5)
            assert close.mean() >= 0.97, (tag, k, close.mean())


def test_main_script_runs_like_the_reference_driver(tmp_path):
    """r-nad_amd/main.py == reference main.py's call pattern (Tree(...).generate/save, RNaD(...).run for several etas)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
    env = dict(os.environ, RNAD_SAVE_DIR=str(tmp_path))
    out = subprocess.run([sys.executable, os.path.join(root, "r-nad_amd", "main.py"), "--updates", "3", "--steps", "20", "--etas", "0", "0.2"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("eta=")]
    assert len(lines) == 2 and all("NashConv by update" in ln for ln in lines)
    assert (tmp_path / "saved_trees" / "small_tree" / "tree.tar").exists()
    runs = sorted(p.name for p in (tmp_path / "saved_runs").iterdir())
    assert len(runs) == 2 and runs[0].endswith("eta=0.0") or runs[0].endswith("eta=0")


def test_off_policy_replay_buffer_path(tmp_path, monkeypatch):
    """n_batches_per_buffer = 2, buffer_mod = 2 (reference rnad.py:66-67, :502-507): the learner sees collated samples of older
    rollouts, so the V-trace importance ratios differ from 1; the update must run and stay finite."""
    from environment.tree import Tree
    from learn.rnad import RNaD

    dev = torch.device("cuda:0")
    monkeypatch.setenv("RNAD_SAVE_DIR", str(tmp_path))
    tree = Tree(device=dev, max_actions=3, max_transitions=2, depth_bound=3, transition_threshold=0.2)
    tree.generate_native(seed=9, prune=(1, 3))
    torch.manual_seed(1)
    np.random.seed(1)
    rn = RNaD(tree=tree, device=dev, directory_name="offpolicy", batch_size=2048, eta=0.2, b1_adam=0.0, lr=1e-3, bounds=[1], delta_m=[6],
              n_batches_per_buffer=2, buffer_mod=2, net_params={"type": "MLP", "max_actions": 3, "width": 64})
    rn.keep_last_log = True
    rn.run(checkpoint_mod=10**9, expl_mod=1, log_mod=1)
    assert rn.total_steps == 6 and rn.m == 1
    assert all(torch.isfinite(p).all() for p in rn.net.parameters())
    assert rn.last_log is not None and np.isfinite(rn.last_log["loss_v"]) and rn.last_log["actor_learner_kld"] >= 0
