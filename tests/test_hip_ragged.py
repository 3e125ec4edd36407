"""Ragged trajectories: the live-row lists (rnad_compact_valid) and the *_rows MLP kernels that skip absorbed (t, b) slots.

What must hold: the list is exactly the ascending positions with indices != 0; listed rows get the dense kernels' results bit
for bit (the per-sample arithmetic is the same), unlisted rows are left alone; weight gradients equal the dense ones with the
masked rows' upstream gradients zeroed (different summation grouping: compared at 1e-5); a rollout / an update that skips
absorbed slots gives the same valid trajectory bits and the same update as the dense one."""
import os
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("N,p", [(0, 0.5), (1, 1.0), (1, 0.0), (63, 0.5), (2047, 0.3), (2048, 0.9), (2049, 0.5), (5000, 0.0), (5000, 1.0),
                                 (1_000_003, 0.4), (2048 * 2048, 0.2), (2048 * 2048 + 1, 0.2), (3 * 2048 * 1024 + 17, 0.05)])
# (up to 2048 chunks of 2048 positions the ordered write adds up the chunk counts before its own itself; beyond, a scan launch does)
def test_compact_valid_is_the_ascending_nonzero_list(N, p):
    import rnad_hip

    rng = np.random.default_rng(N + int(100 * p))
    flags = rng.random(N) < p
    idx = np.where(flags, rng.integers(1, 1000, size=N), 0).astype(np.int32)
    live = rnad_hip.compact_valid(torch.from_numpy(idx).to(DEV))
    n = int(live.count.item())
    want = np.flatnonzero(idx)
    assert n == want.size
    np.testing.assert_array_equal(live.rows[:n].cpu().numpy(), want)


@pytest.mark.parametrize("A,W,N", [(3, 256, 10_000), (2, 64, 77), (5, 128, 4097), (3, 32, 64), (7, 64, 1000)])
@pytest.mark.parametrize("half", (False, True))
def test_mlp_rows_match_the_dense_kernels_on_listed_rows(A, W, N, half):
    import rnad_hip

    g = torch.Generator().manual_seed(A * 1000 + W + N)
    K = 2 * A * A
    shapes = [(W, K), (W,), (1, W), (1,), (W, K), (W,), (A, W), (A,)]
    w = [(torch.randn(s, generator=g) / s[-1] ** 0.5).to(DEV) for s in shapes]
    x = torch.randn((N, 2, A, A), generator=g).to(DEV)
    if half:
        x = x.half()
    flags = torch.rand((N,), generator=g) < 0.4
    idx = torch.where(flags, torch.ones(N, dtype=torch.int32), torch.zeros(N, dtype=torch.int32)).to(DEV)
    live = rnad_hip.compact_valid(idx)
    packed = rnad_hip.mlp_pack(w, A)
    ld, vd = rnad_hip.mlp_forward(packed, W, x, A)
    ll, vl = rnad_hip.mlp_forward(packed, W, x, A, live=live)
    m = flags.to(DEV)
    assert torch.equal(ll[m], ld[m]) and torch.equal(vl[m], vd[m])  # same arithmetic per sample: same bits
    assert (ll[~m] == 0).all() and (vl[~m] == 0).all()                # rows that are not listed are not written
    lp, none = rnad_hip.mlp_forward(packed, W, x, A, want_value=False, live=live)
    assert none is None and torch.equal(lp, ll)
    if not rnad_hip.mlp_backward_supported(A, W):
        return
    dl = torch.randn((N, A), generator=g).to(DEV)
    dv = torch.randn((N, 1), generator=g).to(DEV)
    garbage = torch.full_like(dl, float("nan"))  # rows that are not listed must not even be read
    got = rnad_hip.mlp_backward(packed, w, x, A, torch.where(m[:, None], dl, garbage), torch.where(m[:, None], dv, garbage[:, :1]), live=live)
    want = rnad_hip.mlp_backward(packed, w, x, A, dl * m[:, None], dv * m[:, None])
    for a, b in zip(got, want):
        scale = float(b.abs().max()) + 1e-6
        assert torch.isfinite(a).all()
        np.testing.assert_allclose(a.cpu().numpy() / scale, b.cpu().numpy() / scale, rtol=0, atol=2e-6)


def _ragged_tree(A=3, C=2, depth=5, seed=4):
    from environment.tree import Tree

    tree = Tree(device=DEV, max_actions=A, max_transitions=C, depth_bound=depth, transition_threshold=0.2)
    tree.generate_native(seed=seed, prune=(1, 2))
    assert not tree.handle().uniform_length
    return tree


def test_uniform_length_flag():
    from environment.tree import Tree

    tree = Tree(device=DEV, max_actions=3, max_transitions=1, depth_bound=4)
    tree.generate_native(seed=0)
    assert tree.handle().uniform_length
    assert not _ragged_tree().handle().uniform_length


@pytest.mark.parametrize("half", (False, True))
def test_rollout_that_skips_absorbed_lanes_keeps_every_valid_slot(half):
    from environment.episode import Episodes
    from nn.net import MLP

    torch.manual_seed(1)
    tree = _ragged_tree()
    net = MLP(3, 64, device=DEV)
    B = 20_000
    dense = Episodes(tree, B, seed=9, obs_half=half)
    dense.generate(net, trim=False, tabular=False)
    skip = Episodes(tree, B, seed=9, obs_half=half)
    skip.generate(net, trim=False, skip_absorbed=True, tabular=False)
    T = dense.t_eff + 1
    assert skip.t_eff == dense.t_eff
    valid = dense.indices[:T] != 0
    assert 0.05 < valid.float().mean().item() < 0.95  # really ragged
    assert torch.equal(skip.indices[:T], dense.indices[:T])
    assert torch.equal(skip.alive, dense.alive)
    assert torch.equal(skip.rewards[:T], dense.rewards[:T])  # rewards of absorbed lanes are 0 either way
    assert torch.equal(skip.observations[:T], dense.observations[:T]) and torch.equal(skip.mask_bits[:T], dense.mask_bits[:T])
    for name in ("action_idx", "values"):
        a, b = getattr(skip, name)[:T], getattr(dense, name)[:T]
        assert torch.equal(a[valid], b[valid]), name
    assert torch.equal(skip.policy[:T][valid], dense.policy[:T][valid])
    assert torch.isfinite(skip.policy[:T]).all() and torch.isfinite(skip.values[:T]).all()


def test_rollout_without_the_actor_value_head_changes_nothing_else():
    from environment.episode import Episodes
    from nn.net import MLP

    torch.manual_seed(2)
    tree = _ragged_tree()
    net = MLP(3, 64, device=DEV)
    full = Episodes(tree, 5000, seed=4)
    full.generate(net, tabular=False)
    lean = Episodes(tree, 5000, seed=4)
    lean.generate(net, store_values=False, tabular=False)
    assert lean.t_eff == full.t_eff
    for name in ("indices", "observations", "mask_bits", "policy", "action_idx", "rewards", "alive"):
        assert torch.equal(getattr(lean, name), getattr(full, name)), name
    assert (lean.values == 0).all() and (full.values != 0).any()


@pytest.mark.parametrize("reuse", (False, True))
def test_update_that_skips_absorbed_slots_is_the_dense_update(reuse):
    """Same learner, same nets: gradients from (dense rollout, dense update) vs (rollout and update that skip absorbed slots)."""
    from environment.episode import Episodes
    from learn.rnad import RNaD

    tree = _ragged_tree()
    os.environ["RNAD_SAVE_DIR"] = tempfile.mkdtemp(prefix="rnad_test_")
    B = 1 << 14
    torch.manual_seed(5)
    rn = RNaD(tree=tree, device=DEV, directory_name=f"ragged{int(reuse)}", batch_size=B, eta=0.2, b1_adam=0.0, lr=1e-3,
              net_params={"type": "MLP", "max_actions": 3, "width": 64})
    rn.initialize()
    with torch.no_grad():  # make the four nets differ, as they do after the first outer iteration
        for i, m in enumerate((rn.net_target, rn.net_reg, rn.net_reg_)):
            for p_ in m.parameters():
                p_.add_(0.05 * (i + 1) * torch.randn_like(p_))
    rn.reuse_actor_outputs = reuse
    rn.tabular = False  # the per-slot forwards are what this test is about
    grads, losses = [], []
    for skip in (False, True):
        ep = Episodes(tree, B, seed=3)
        ep.generate(rn.net, trim=False, keep_logits=reuse, skip_absorbed=skip and not reuse, tabular=False)
        ep._actor_tag = (id(rn.net), rn.total_steps)
        rn.skip_absorbed = skip
        rn.optimizer.zero_grad()
        rn._RNaD__learn(ep, 0.4)
        grads.append([p_.grad.detach().clone() for p_ in rn.net.parameters()])
    for a, b in zip(*grads):
        scale = float(b.abs().max()) + 1e-12
        assert torch.isfinite(a).all() and float(b.abs().max()) > 0
        np.testing.assert_allclose(a.cpu().numpy() / scale, b.cpu().numpy() / scale, rtol=0, atol=2e-6)


def test_one_reg_net_evaluation_when_the_other_cannot_matter(monkeypatch):
    """rnad.py:382 with alpha == 1 (weight 0 on net_reg_), alpha == 0 (weight 0 on net_reg) or identical reg nets: evaluating
    one net for both operands gives the same gradient bits as evaluating both."""
    from environment.episode import Episodes
    from learn.rnad import RNaD

    tree = _ragged_tree()
    os.environ["RNAD_SAVE_DIR"] = tempfile.mkdtemp(prefix="rnad_test_")
    B = 1 << 12
    torch.manual_seed(7)
    rn = RNaD(tree=tree, device=DEV, directory_name="alias", batch_size=B, eta=0.2, b1_adam=0.0,
              net_params={"type": "MLP", "max_actions": 3, "width": 64})
    rn.initialize()
    ep = Episodes(tree, B, seed=3)
    ep.generate(rn.net, trim=False)
    rn.tabular = False

    def grads(alpha):
        rn.optimizer.zero_grad()
        rn._RNaD__learn(ep, alpha)
        return [p_.grad.detach().clone() for p_ in rn.net.parameters()]

    assert rn._reg_nets_identical()
    one = grads(0.3)                                                   # identical nets: one evaluation
    monkeypatch.setattr(RNaD, "_reg_nets_identical", lambda self: False)
    two = grads(0.3)                                                   # forced: both evaluated
    assert all(torch.equal(a, b) for a, b in zip(one, two))
    monkeypatch.undo()
    with torch.no_grad():
        for p_ in rn.net_reg_.parameters():
            p_.add_(0.1 * torch.randn_like(p_))
    assert not rn._reg_nets_identical()                                # the version-keyed cache noticed
    g1 = grads(1)
    with torch.no_grad():
        for p_ in rn.net_reg_.parameters():
            p_.add_(0.1 * torch.randn_like(p_))
    g1b = grads(1)                                                     # alpha == 1: net_reg_ must not matter
    assert all(torch.equal(a, b) for a, b in zip(g1, g1b))
    g0 = grads(0)
    with torch.no_grad():
        for p_ in rn.net_reg.parameters():
            p_.add_(0.1 * torch.randn_like(p_))
    g0b = grads(0)                                                     # alpha == 0: net_reg must not matter
    assert all(torch.equal(a, b) for a, b in zip(g0, g0b))
    assert not all(torch.equal(a, b) for a, b in zip(g0, g1))


# ------------------------------------------------------------------------------------------------ tabular evaluation
@pytest.mark.parametrize("half", (False, True))
@pytest.mark.parametrize("ragged", (False, True))
def test_tabular_rollout_is_the_dense_rollout(half, ragged):
    """One actor evaluation per (player, state) + per-lane gathers == one evaluation per lane and step, bit for bit."""
    from environment.episode import Episodes
    from environment.tree import Tree
    from nn.net import MLP

    torch.manual_seed(3)
    if ragged:
        tree = _ragged_tree()
    else:
        tree = Tree(device=DEV, max_actions=3, max_transitions=1, depth_bound=4)
        tree.generate_native(seed=2)
    net = MLP(3, 64, device=DEV)
    B = 30_000
    dense = Episodes(tree, B, seed=11, obs_half=half)
    dense.generate(net, trim=False, tabular=False)
    tab = Episodes(tree, B, seed=11, obs_half=half)
    tab.generate(net, trim=False, tabular=True)
    T = dense.t_eff + 1
    for name in ("indices", "observations", "mask_bits", "policy", "action_idx", "rewards", "values", "alive"):
        assert torch.equal(getattr(tab, name)[:T], getattr(dense, name)[:T]), name  # absorbed lanes too: state 0 has a row as well
    lean = Episodes(tree, B, seed=11, obs_half=half)
    lean.generate(net, trim=False, tabular=True, store_values=False)
    assert (lean.values == 0).all() and torch.equal(lean.policy[:T], dense.policy[:T])


@pytest.mark.parametrize("ragged", (False, True))
def test_tabular_update_is_the_dense_update(ragged):
    from environment.episode import Episodes
    from environment.tree import Tree
    from learn.rnad import RNaD

    if ragged:
        tree = _ragged_tree()
    else:
        tree = Tree(device=DEV, max_actions=3, max_transitions=1, depth_bound=4)
        tree.generate_native(seed=2)
    os.environ["RNAD_SAVE_DIR"] = tempfile.mkdtemp(prefix="rnad_test_")
    B = 1 << 15
    torch.manual_seed(6)
    rn = RNaD(tree=tree, device=DEV, directory_name=f"tab{int(ragged)}", batch_size=B, eta=0.2, b1_adam=0.0,
              net_params={"type": "MLP", "max_actions": 3, "width": 64})
    rn.initialize()
    with torch.no_grad():
        for i, m in enumerate((rn.net_target, rn.net_reg, rn.net_reg_)):
            for p_ in m.parameters():
                p_.add_(0.05 * (i + 1) * torch.randn_like(p_))
    ep = Episodes(tree, B, seed=3)
    ep.generate(rn.net, trim=False)
    assert 8 * tree.handle().S <= (ep.t_eff + 1) * B
    out = {}
    for tabular in (False, "forward", True):
        rn.tabular = tabular
        rn.skip_absorbed = False
        rn.optimizer.zero_grad()
        rn._RNaD__learn(ep, 0.4)
        out[tabular] = [p_.grad.detach().clone() for p_ in rn.net.parameters()]
    for a, b in zip(out["forward"], out[False]):
        assert torch.equal(a, b)  # deduplicated forwards, per-slot backward: the dense path's bits
    for a, b in zip(out[True], out[False]):
        scale = float(b.abs().max()) + 1e-12
        assert torch.isfinite(a).all() and float(b.abs().max()) > 0
        np.testing.assert_allclose(a.cpu().numpy() / scale, b.cpu().numpy() / scale, rtol=0, atol=2e-6)
    rn.tabular = True  # fixed-point row sums: integer atomics, so a second run gives the same bits
    rn.optimizer.zero_grad()
    rn._RNaD__learn(ep, 0.4)
    assert all(torch.equal(p_.grad, b) for p_, b in zip(rn.net.parameters(), out[True]))
    if ragged:  # forward mode + skipped absorbed slots in the backward
        rn.tabular, rn.skip_absorbed = "forward", True
        rn.optimizer.zero_grad()
        rn._RNaD__learn(ep, 0.4)
        for p_, b in zip(rn.net.parameters(), out[False]):
            scale = float(b.abs().max()) + 1e-12
            np.testing.assert_allclose(p_.grad.cpu().numpy() / scale, b.cpu().numpy() / scale, rtol=0, atol=2e-6)


def test_tabular_training_steps_track_the_dense_ones():
    from environment.episode import Buffer
    from environment.tree import Tree
    from learn.rnad import RNaD

    tree = Tree(device=DEV, max_actions=3, max_transitions=1, depth_bound=4)
    tree.generate_native(seed=2)
    os.environ["RNAD_SAVE_DIR"] = tempfile.mkdtemp(prefix="rnad_test_")
    finals = []
    for tabular in (False, True):
        torch.manual_seed(8)
        rn = RNaD(tree=tree, device=DEV, directory_name=f"tabrun{int(tabular)}", batch_size=1 << 14, eta=0.2, b1_adam=0.0, lr=1e-4,
                  net_params={"type": "MLP", "max_actions": 3, "width": 64})
        rn.initialize()
        rn.tabular = tabular
        buf = Buffer(1)
        for i in range(5):
            rn.train_step(buf, alpha=0.2 * i)
            rn.total_steps += 1
        finals.append([p_.detach().clone() for p_ in rn.net.parameters()])
    for a, b in zip(*finals):
        # Adam (b1 = 0) moves every weight by ~lr per step whatever the gradient's size: a last-bit difference in a tiny gradient
        # can flip a step, so compare at a few lr
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=1e-3)
        assert torch.isfinite(a).all()


@pytest.mark.parametrize("scale,max_norm", [(1.0, 1e3), (100.0, 1.0), (0.0, 1.0)])
def test_clip_grad_norm_matches_torch(scale, max_norm):
    import rnad_hip

    g = torch.Generator().manual_seed(3)
    shapes = [(256, 18), (256,), (1, 256), (1,), (256, 18), (256,), (3, 256), (3,)]
    n = sum(int(np.prod(s)) for s in shapes)
    flat = (torch.randn((n,), generator=g) * scale).to(DEV)
    params, off = [], 0
    for s in shapes:
        p_ = torch.nn.Parameter(torch.zeros(s, device=DEV))
        k = int(np.prod(s))
        p_.grad = flat[off: off + k].view(s).clone()
        params.append(p_)
        off += k
    want_norm = torch.nn.utils.clip_grad_norm_(params, max_norm)
    mine = flat.clone()
    rnad_hip.clip_grad_norm(mine, max_norm)
    want = torch.cat([p_.grad.reshape(-1) for p_ in params])
    if float(want_norm) <= max_norm:
        assert torch.equal(mine, flat) and torch.equal(want, flat)  # coefficient clamped to exactly 1
    else:
        np.testing.assert_allclose(mine.cpu().numpy(), want.cpu().numpy(), rtol=2e-6, atol=0)


@pytest.mark.parametrize("ragged", (False, True))
def test_logged_statistics_do_not_depend_on_the_mode(ragged):
    """A logging step (rnad.py:427-452: per-slot logit / policy statistics) through the tables gives the dense path's numbers."""
    from environment.episode import Episodes
    from environment.tree import Tree
    from learn.rnad import RNaD

    if ragged:
        tree = _ragged_tree()
    else:
        tree = Tree(device=DEV, max_actions=3, max_transitions=1, depth_bound=4)
        tree.generate_native(seed=2)
    os.environ["RNAD_SAVE_DIR"] = tempfile.mkdtemp(prefix="rnad_test_")
    B = 1 << 13
    torch.manual_seed(9)
    rn = RNaD(tree=tree, device=DEV, directory_name=f"log{int(ragged)}", batch_size=B, eta=0.2, b1_adam=0.0,
              net_params={"type": "MLP", "max_actions": 3, "width": 64})
    rn.initialize()
    rn.fold_legal = False  # the table evaluations with the very kernels of the per-slot modes: identical net outputs in every mode
    with torch.no_grad():
        for i, m in enumerate((rn.net_target, rn.net_reg, rn.net_reg_)):
            for p_ in m.parameters():
                p_.add_(0.05 * (i + 1) * torch.randn_like(p_))
    ep = Episodes(tree, B, seed=3)
    ep.generate(rn.net, trim=False)
    logs, grads = {}, {}
    for mode in (False, "forward", True, "folded"):
        if mode == "folded":
            if not tree.handle().legal_foldable:
                continue
            rn.tabular, rn.fold_legal = True, True
        else:
            rn.tabular = mode
        rn.optimizer.zero_grad()
        logs[mode] = {}
        rn._RNaD__learn(ep, 0.4, log=logs[mode])
        grads[mode] = [p_.grad.detach().clone() for p_ in rn.net.parameters()]
    if "folded" in logs:  # the legal fold sums the first layer in another order: the same statistics to fp32 rounding
        for k, want in logs[True].items():
            assert abs(logs["folded"][k] - want) <= 2e-6 * max(1.0, abs(want)), k
        for a, b in zip(grads["folded"], grads[True]):
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-5 * (b.abs().max().item() + 1e-12))
        del logs["folded"], grads["folded"]
    assert set(logs[False]) == set(logs["forward"]) == set(logs[True]) and "entropy" in logs[False]
    for k, want in logs[False].items():
        if k.startswith("loss"):  # fp64 atomic partial sums: the last bits vary from launch to launch
            assert abs(logs["forward"][k] - want) <= 1e-9 * max(1.0, abs(want)), k
            assert abs(logs[True][k] - want) <= 1e-9 * max(1.0, abs(want)), k
        elif k == "gradient_norm":
            assert logs["forward"][k] == want
            assert abs(logs[True][k] - want) <= 1e-5 * want
        else:
            assert logs["forward"][k] == want and logs[True][k] == want, k
    assert all(torch.equal(a, b) for a, b in zip(grads["forward"], grads[False]))


@pytest.mark.parametrize("ragged", (False, True))
def test_forward_batch_through_the_observation_table(ragged):
    """MLP.forward_batch (the reference learner's entry point, net.py:64-85) on a small tree: values are the per-slot
    evaluation's bits, autograd gives the per-slot weight gradients up to summation order."""
    from environment.episode import Episodes
    from environment.tree import Tree
    from nn.net import MLP

    torch.manual_seed(4)
    if ragged:
        tree = _ragged_tree()
    else:
        tree = Tree(device=DEV, max_actions=3, max_transitions=1, depth_bound=4)
        tree.generate_native(seed=2)
    net = MLP(3, 64, device=DEV)
    ep = Episodes(tree, 1 << 13, seed=2)
    ep.generate(net)
    T, B, A = ep.t_eff + 1, ep.batch_size, 3
    assert 8 * tree.handle().S <= T * B
    logit, log_pi, pi, v = net.forward_batch(ep)                       # through the table, with autograd
    want_l, want_v = net.forward_logits(ep.observations[:T])           # per slot (FusedMLP autograd node)
    assert torch.equal(logit.reshape(-1, A), want_l) and torch.equal(v.reshape(-1, 1), want_v)
    with torch.no_grad():
        l2, lp2, p2, v2 = net.forward_batch(ep)
    assert torch.equal(l2, logit) and torch.equal(p2, pi) and torch.equal(lp2, log_pi) and torch.equal(v2, v)
    g = torch.Generator().manual_seed(1)
    valid = (ep.indices[:T] != 0).float()
    dl = (torch.randn((T, B, A), generator=g).to(DEV) * valid[..., None]).contiguous()
    dv = (torch.randn((T, B, 1), generator=g).to(DEV) * valid[..., None]).contiguous()
    net.zero_grad()
    torch.autograd.backward([logit, v], [dl, dv])
    got = [p_.grad.clone() for p_ in net.parameters()]
    net.zero_grad()
    torch.autograd.backward([want_l, want_v], [dl.view(-1, A), dv.view(-1, 1)])
    for a, p_ in zip(got, net.parameters()):
        scale = float(p_.grad.abs().max()) + 1e-12
        np.testing.assert_allclose(a.cpu().numpy() / scale, p_.grad.cpu().numpy() / scale, rtol=0, atol=2e-6)
