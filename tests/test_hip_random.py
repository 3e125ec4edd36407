"""Randomised GPU-vs-oracle sweeps over the whole supported shape range (A = 1..8 actions, C = 1..8 chance outcomes,
T up to 32 steps, ragged validity, masked actions): integer results and same-op-order fp32 results must be bit-identical."""
import numpy as np
import pytest
import torch

from _util import assert_bits_equal

pytestmark = pytest.mark.gpu


def _random_tree(rng, A, C, S):
    """Reference-layout tables of a random forward-pointing tree (children have larger ids), ragged legality."""
    index = np.zeros((S, C, A, A), np.int64)
    chance = np.zeros((S, C, A, A), np.float32)
    value = rng.standard_normal((S, C, A, A)).astype(np.float32)
    legal = np.zeros((S, 1, A, A), np.float32)
    ev = np.zeros((S, 1, A, A), np.float32)
    nxt = 2
    for s in range(1, S):
        ra, ca = rng.integers(1, A + 1), rng.integers(1, A + 1)
        legal[s, 0, :ra, :ca] = 1
        p = rng.dirichlet(np.ones(C) * 0.7, size=(A, A)).astype(np.float32)
        p[p < 0.15] = 0
        p[..., 0] = np.where(p.sum(-1) == 0, 1, p[..., 0])
        p = p / p.sum(-1, keepdims=True)
        chance[s] = np.moveaxis(p, 2, 0) * legal[s]
        for r in range(ra):
            for c in range(ca):
                for t in range(C):
                    if chance[s, t, r, c] > 0 and nxt < S and rng.random() < 0.5:
                        index[s, t, r, c] = nxt
                        nxt += 1
        ev[s, 0] = (value[s] * chance[s]).sum(0)
    legal[0, 0, 0, 0] = 1
    chance[0, 0, 0, 0] = 1
    value[0] = 0
    # every state id must be reachable for rnad_tree_create's level pass only if referenced; unreferenced ids are fine
    return dict(index=index, value=value, chance=chance, expected_value=ev, legal=legal, root_value=np.zeros((S, 1), np.float32),
                solution=np.zeros((S, 2 * A), np.float32))


@pytest.mark.parametrize("A,C", [(1, 1), (2, 3), (4, 2), (6, 5), (7, 8), (8, 8), (8, 1)])
def test_env_kernels_on_random_trees(A, C):
    import rnad_hip
    from _gpu import DEV, gpu, tree_from_arrays
    from oracle import oracle

    rng = np.random.default_rng(100 * A + C)
    S, B = 300, 5000
    arrs = _random_tree(rng, A, C, S)
    tree = tree_from_arrays(arrs)
    h = tree.handle()
    idx = rng.integers(0, S, size=B)
    for player in (0, 1):
        want, want_mask = oracle.observe(arrs["expected_value"], arrs["legal"], idx, np.full(B, player))
        bits = torch.empty((B,), dtype=torch.uint8, device=DEV)
        mask = torch.empty((B, A), dtype=torch.float32, device=DEV)
        got = rnad_hip.observe(h, gpu(idx, torch.int32), player, mask_bits=bits, mask=mask)
        assert_bits_equal(got.cpu().numpy(), want, f"observe A={A} p={player}")
        assert_bits_equal(mask.cpu().numpy(), want_mask, "mask")
        np.testing.assert_array_equal(bits.cpu().numpy(), (want_mask.astype(np.int64) * (1 << np.arange(A))).sum(-1))
    lg = arrs["legal"][idx, 0]
    r = np.array([rng.choice(np.flatnonzero(lg[b, :, 0])) for b in range(B)])
    c = np.array([rng.choice(np.flatnonzero(lg[b, 0, :])) for b in range(B)])
    for mode in ("seeded", "explicit"):
        noise = rng.exponential(size=(B, C)).astype(np.float32) if mode == "explicit" else oracle.chance_uniform(B, 11, 7, 3)
        nxt, rew = rnad_hip.transition(h, gpu(idx, torch.int32), gpu(r, torch.int32), gpu(c, torch.int32),
                                       noise=gpu(noise) if mode == "explicit" else None, seed=11, lane0=7, step=3)
        want_next, want_rew = oracle.transition(arrs["index"], arrs["chance"], arrs["value"], idx, r, c, noise)
        np.testing.assert_array_equal(nxt.cpu().numpy(), want_next)
        assert_bits_equal(rew.cpu().numpy(), want_rew, "reward")


@pytest.mark.parametrize("A,T,B", [(1, 2, 100), (2, 32, 777), (3, 12, 4096), (4, 7, 1000), (5, 16, 513), (8, 9, 300)])
def test_learner_kernels_on_random_trajectories(A, T, B):
    """process_policy, both players' v_trace and the closed-form loss gradients, single kernels and the fused pass."""
    import rnad_hip
    from _gpu import DEV, gpu
    from oracle import oracle

    rng = np.random.default_rng(A * 1000 + T)
    lengths = rng.integers(0, T + 1, size=B)
    lengths[:3] = (0, T, 1)
    valid = (np.arange(T)[:, None] < lengths[None, :]).astype(np.float32)
    masks = (rng.random((T, B, A)) < 0.75).astype(np.float32)
    masks[..., 0] = 1

    def pol():
        x = rng.dirichlet(np.ones(A) * 0.6, size=(T, B)).astype(np.float32) * masks
        return (x / x.sum(-1, keepdims=True)).astype(np.float32)

    mu, pi = pol(), pol()
    pip_want = oracle.process_policy(pi, masks, 32, 0.03)
    pip = rnad_hip.process_policy(gpu(pi).view(-1, A), gpu(masks).view(-1, A), 32, 0.03)
    assert_bits_equal(pip.cpu().numpy().reshape(T, B, A), pip_want, "process_policy")
    logpi = (rng.standard_normal((T, B, A)) * 0.4).astype(np.float32) * masks
    act = np.array([[rng.choice(A, p=mu[t, b]) for b in range(B)] for t in range(T)])
    v = rng.standard_normal((T, B)).astype(np.float32)
    reward = (rng.standard_normal((T, B)) * (rng.random((T, B)) < 0.3)).astype(np.float32)
    turns = np.broadcast_to((np.arange(T) % 2)[:, None], (T, B)).astype(np.int64)
    a_oh = np.eye(A, dtype=np.float32)[act]
    hp = dict(eta=0.3, lambda_=0.95, c=0.9, rho=1.1, gamma=0.99)
    # the oracle and the kernels both sum in index order, so they agree bitwise for every A (torch itself deviates at A >= 5)
    for p in range(2):
        rew = reward if p == 0 else -reward
        vt_w, hp_w, q_w = oracle.vtrace(v[..., None], valid, turns, mu, pip_want, logpi, a_oh, rew, p, **hp)
        vt, has, q = rnad_hip.vtrace(gpu(v), gpu(valid), None, gpu(mu), gpu(pip_want), gpu(logpi), gpu(act, torch.int32), gpu(rew), p, **hp)
        np.testing.assert_array_equal(has.cpu().numpy(), hp_w)
        assert_bits_equal(vt.cpu().numpy(), vt_w[..., 0], f"v_target p{p}")
        assert_bits_equal(q.cpu().numpy(), q_w, f"q p{p}")


def test_rollout_step_count_limits():
    """T_cap = 32 steps (BASELINE.json configs[0] names 32-step episodes as the capacity) on a depth-16 chain tree."""
    from _gpu import DEV, tree_from_arrays
    from environment.episode import Episodes
    from nn.net import MLP

    A, C, depth = 2, 1, 16
    S = depth + 1
    arrs = dict(index=np.zeros((S, C, A, A), np.int64), value=np.zeros((S, C, A, A), np.float32), chance=np.zeros((S, C, A, A), np.float32),
                expected_value=np.zeros((S, 1, A, A), np.float32), legal=np.zeros((S, 1, A, A), np.float32),
                root_value=np.zeros((S, 1), np.float32), solution=np.zeros((S, 2 * A), np.float32))
    arrs["legal"][0, 0, 0, 0] = 1
    arrs["chance"][0, 0, 0, 0] = 1
    for s in range(1, S):
        arrs["legal"][s] = 1
        arrs["chance"][s] = 1
        arrs["index"][s, 0, 0, :] = s + 1 if s + 1 < S else 0  # row action 0 continues down the chain, row action 1 ends the game
        arrs["value"][s, 0, 1, :] = 1.0
        arrs["value"][s, 0, 0, :] = -1.0 if s + 1 == S else 0.0
    tree = tree_from_arrays(arrs, depth_bound=depth)
    assert tree.handle().max_depth == depth
    torch.manual_seed(0)
    ep = Episodes(tree, 4096, seed=3)
    ep.generate(MLP(A, 32, device=DEV))
    T = ep.t_eff + 1
    assert T <= 32 and T % 2 == 0
    alive = ep.alive.cpu().numpy()
    assert alive[0] == 4096 and (np.diff(alive) <= 0).all() and alive[T] == 0
    # a lane is rewarded exactly once, on the column step where it leaves the tree
    assert ((ep.rewards != 0).sum(0) == 1).all()


@pytest.mark.parametrize("A,C,W", [(2, 3, 32), (3, 2, 64), (4, 2, 64), (5, 1, 32), (7, 3, 32)])
def test_net_evaluation_modes_agree_on_random_trees(A, C, W, tmp_path, monkeypatch):
    """RNaD.tabular False / "forward" / True on random forward-pointing trees with ragged legality (state ids in no particular
    tree order, episodes of every length): identical rollouts, forward == dense bit for bit, True to summation order."""
    from _gpu import DEV, tree_from_arrays
    from environment.episode import Episodes
    from learn.rnad import RNaD

    rng = np.random.default_rng(7 * A + C)
    tree = tree_from_arrays(_random_tree(rng, A, C, 400))
    monkeypatch.setenv("RNAD_SAVE_DIR", str(tmp_path))
    B = 1 << 13
    torch.manual_seed(A + C)
    rn = RNaD(tree=tree, device=DEV, directory_name=f"rand{A}{C}", batch_size=B, eta=0.2, b1_adam=0.0,
              net_params={"type": "MLP", "max_actions": A, "width": W})
    rn.initialize()
    with torch.no_grad():
        for i, m in enumerate((rn.net_target, rn.net_reg, rn.net_reg_)):
            for p_ in m.parameters():
                p_.add_(0.05 * (i + 1) * torch.randn_like(p_))
    dense = Episodes(tree, B, seed=5)
    dense.generate(rn.net, trim=False, tabular=False)
    tab = Episodes(tree, B, seed=5)
    tab.generate(rn.net, trim=False, tabular=True)
    T = dense.t_eff + 1
    for name in ("indices", "observations", "mask_bits", "policy", "action_idx", "rewards", "values", "alive"):
        assert torch.equal(getattr(tab, name)[:T], getattr(dense, name)[:T]), name
    assert 8 * tree.handle().S <= T * B
    grads = {}
    for mode in (False, "forward", True):
        rn.tabular = mode
        rn.optimizer.zero_grad()
        rn._RNaD__learn(dense, 0.3)
        grads[mode] = [p_.grad.detach().clone() for p_ in rn.net.parameters()]
    for a, b in zip(grads["forward"], grads[False]):
        assert torch.equal(a, b)
    for a, b in zip(grads[True], grads[False]):
        scale = float(b.abs().max()) + 1e-12
        assert torch.isfinite(a).all()
        np.testing.assert_allclose(a.cpu().numpy() / scale, b.cpu().numpy() / scale, rtol=0, atol=2e-6)
