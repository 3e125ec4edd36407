"""Parity of the HIP path (through the C-ABI / ctypes binding) with the reference's golden vectors and the CPU oracle.

Run on the GPU box:  python -m pytest tests -m gpu -x -q
Bar: integer / index / byte results bit-exact; fp32 results bit-exact where the kernel replays the reference's op order
on identical inputs, else within TOL = 1e-5 (exp/log go through the device libm).
"""
import json

import numpy as np
import pytest
import torch

from _util import TREES, assert_bits_equal, load, load_tree, mlp_weights

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def G():
    import _gpu

    return _gpu


@pytest.fixture(scope="module")
def hip():
    import rnad_hip

    return rnad_hip


# ------------------------------------------------------------------------------------------------ K1 observe
@pytest.mark.parametrize("name", TREES)
def test_observe_bit_exact_vs_reference(G, hip, name):
    tree, _ = G.golden_tree(name)
    ro = load("rollout_" + name)
    T, B, A = int(ro["t_eff"]) + 1, ro["indices"].shape[1], tree.max_actions
    for t in range(T):
        idx = G.gpu(ro["indices"][t], torch.int32)
        bits = torch.empty((B,), dtype=torch.uint8, device=G.DEV)
        mask = torch.empty((B, A), dtype=torch.float32, device=G.DEV)
        obs = hip.observe(tree.handle(), idx, t & 1, mask_bits=bits, mask=mask)
        assert_bits_equal(G.cpu(obs), ro["observations"][t], f"obs t={t}")  # incl. -0.0 in the column player's view
        assert_bits_equal(G.cpu(mask), ro["masks"][t], f"mask t={t}")
        np.testing.assert_array_equal(G.cpu(bits), G.mask_bits_of(ro["masks"][t]))


@pytest.mark.parametrize("name,B", [("small", 100_003), ("a5", 4097), ("c1", 1), ("ragged", 255)])
def test_observe_ragged_batches_vs_oracle(G, hip, name, B):
    """Odd batch sizes: partial last block, scalar tail, unaligned output slices."""
    from oracle import oracle

    tree, g = G.golden_tree(name)
    rng = np.random.default_rng(B)
    idx = rng.integers(0, g["index"].shape[0], size=B)
    A = tree.max_actions
    for player in (0, 1):
        want, want_mask = oracle.observe(g["expected_value"], g["legal"], idx, np.full(B, player))
        got = hip.observe(tree.handle(), G.gpu(idx, torch.int32), player)
        assert_bits_equal(G.cpu(got), want, f"{name} B={B} p={player}")
        # an output buffer that starts at a non-16-byte-aligned address
        buf = torch.zeros((B * 2 * A * A + 1,), dtype=torch.float32, device=G.DEV)
        view = buf[1:].view(B, 2, A, A)
        hip.observe(tree.handle(), G.gpu(idx, torch.int32), player, obs=view)
        assert_bits_equal(G.cpu(view), want, "unaligned")
        half = hip.observe(tree.handle(), G.gpu(idx, torch.int32), player, half=True)
        assert_bits_equal(G.cpu(half), want.astype(np.float16), "fp16 observations")
    empty = hip.observe(tree.handle(), torch.empty((0,), dtype=torch.int32, device=G.DEV), 0)
    assert empty.shape == (0, 2, A, A)


# ------------------------------------------------------------------------------------------------ K3 / K2
@pytest.mark.parametrize("name", TREES)
def test_sampler_and_transition_replay_reference_noise(G, hip, name):
    tree, _ = G.golden_tree(name)
    ro = load("rollout_" + name)
    T = int(ro["t_eff"]) + 1
    act = ro["actions"].argmax(-1)
    for t in range(T):
        a = hip.sample(G.gpu(ro["policy"][t]), noise=G.gpu(ro["noise_action"][t]))
        np.testing.assert_array_equal(G.cpu(a), act[t])
        if t & 1:
            alive = torch.zeros((1,), dtype=torch.int32, device=G.DEV)
            nxt, rew = hip.transition(tree.handle(), G.gpu(ro["indices"][t], torch.int32), G.gpu(act[t - 1], torch.int32),
                                      G.gpu(act[t], torch.int32), noise=G.gpu(ro["noise_chance"][t]), alive=alive)
            want_next = ro["indices"][t + 1] if t + 1 < T else np.zeros_like(ro["indices"][t])
            np.testing.assert_array_equal(G.cpu(nxt), want_next)
            assert_bits_equal(G.cpu(rew), ro["rewards"][t], f"reward t={t}")  # -0.0 where value < 0 and not terminal
            assert int(alive.item()) == int((want_next != 0).sum())


def test_seeded_draws_match_oracle_bit_for_bit(G, hip):
    from oracle import oracle

    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 4, 5, 8):
        B = 5000
        p = rng.dirichlet(np.ones(n), size=B).astype(np.float32)
        p[rng.random((B, n)) < 0.2] = 0
        p[:, 0] = np.maximum(p[:, 0], 1e-3)
        for seed, lane0, step, stream in ((1, 0, 0, 0), (2**40 + 17, 2**33, 11, 1), (99, 123456, 31, 0)):
            got = hip.sample(G.gpu(p), seed=seed, lane0=lane0, step=step, stream_id=stream)
            u = oracle.chance_uniform(B, seed, lane0, step) if stream else oracle.action_uniform(B, seed, lane0, step)
            want = oracle.pick(p, u)
            assert (p[np.arange(B), want] > 0).all()  # a category of weight zero is never drawn
            np.testing.assert_array_equal(G.cpu(got), want)


def test_seeded_draws_are_the_numpy_definition_on_adversarial_weights(G, hip):
    """rnad_sample(seed=...) on the device against the contract written out in numpy (tests/_numpy_rng.py: philox4x32-10, the slot's
    uniform, the fp32 inverse CDF) -- NOT through include/rnad_rng.h, which kernels and oracle share.  Zeros anywhere, weights of 1e-30
    and 1e30, exact ties, all mass on the last category; 64-bit seeds and lane offsets; every category count the kernels are built for."""
    from tests import _numpy_rng as nprng

    rng = np.random.default_rng(21)
    for n in (1, 2, 3, 4, 5, 8):
        B = 65_537
        p = nprng.adversarial_weights(rng, B, n)
        for seed, lane0, step, stream in ((1, 0, 0, 0), (2**40 + 17, 2**33, 11, 1), (99, 123456, 31, 0), (2**63 + 5, 2**40, 6, 0), (7, 1, 1, 1)):
            got = G.cpu(hip.sample(G.gpu(p), seed=seed, lane0=lane0, step=step, stream_id=stream))
            want = nprng.pick(p, nprng.slot_uniform(B, seed, lane0, step, stream))
            np.testing.assert_array_equal(got, want, err_msg=f"n={n} seed={seed} lane0={lane0} step={step} stream={stream}")
            assert (p[np.arange(B), got] > 0).all()  # a category of weight zero is never drawn


def test_seeded_draws_follow_the_policy_on_the_device(G, hip):
    """torch.multinomial's contract: the drawn categories are distributed as the weights.  2^20 seeded draws per decision slot (row
    player's action, column player's action, chance) against a fixed policy: chi-square, and no draw of a zero weight."""
    n = 1 << 20
    p = np.array([0.03, 0.17, 0.0, 0.45, 0.35], np.float32)
    probs = G.gpu(np.repeat(p[None], n, 0))
    for step, stream in ((0, 0), (1, 0), (1, 1), (7, 0)):
        got = G.cpu(hip.sample(probs, seed=2024, lane0=0, step=step, stream_id=stream))
        cnt = np.bincount(got, minlength=5).astype(np.float64)
        assert cnt[2] == 0
        live = p > 0
        chi2 = (((cnt - n * p) ** 2)[live] / (n * p[live])).sum()
        assert chi2 < 21.1, (step, stream, chi2)  # 3 degrees of freedom: P(chi2 > 21.1) ~ 1e-4


@pytest.mark.parametrize("name", ("small", "ragged", "a5"))
def test_seeded_transition_matches_oracle(G, hip, name):
    from oracle import oracle

    tree, g = G.golden_tree(name)
    S, C, A, _ = g["index"].shape
    rng = np.random.default_rng(5)
    B = 20_001
    idx = rng.integers(0, S, size=B)
    legal = g["legal"][idx, 0]
    r = np.array([rng.choice(np.flatnonzero(legal[b, :, 0])) for b in range(B)])
    c = np.array([rng.choice(np.flatnonzero(legal[b, 0, :])) for b in range(B)])
    seed, lane0, step = 77, 10**6, 5
    nxt, rew = hip.transition(tree.handle(), G.gpu(idx, torch.int32), G.gpu(r, torch.int32), G.gpu(c, torch.int32), seed=seed,
                              lane0=lane0, step=step)
    want_next, want_rew = oracle.transition(g["index"], g["chance"], g["value"], idx, r, c, oracle.chance_uniform(B, seed, lane0, step))
    np.testing.assert_array_equal(G.cpu(nxt), want_next)
    assert_bits_equal(G.cpu(rew), want_rew, "reward")


@pytest.mark.parametrize("name", TREES)
def test_policy_head(G, hip, name):
    ro = load("rollout_" + name)
    T = int(ro["t_eff"]) + 1
    for t in range(T):
        pol = hip.policy_head(G.gpu(ro["logits"][t]), mask=G.gpu(ro["masks"][t]))
        np.testing.assert_allclose(G.cpu(pol), ro["policy"][t], rtol=TOL, atol=1e-7)
        pol2 = hip.policy_head(G.gpu(ro["logits"][t]), mask_bits=G.gpu(G.mask_bits_of(ro["masks"][t])))
        assert torch.equal(pol, pol2)
        assert ((G.cpu(pol) == 0) == (ro["masks"][t] == 0)).all()


# ------------------------------------------------------------------------------------------------ fused MLP
@pytest.mark.parametrize("name", TREES)
def test_mlp_forward_vs_reference(G, hip, name):
    """The reference net's own logits / values (recorded in the rollout fixtures) from the fused MFMA kernel."""
    tree, _ = G.golden_tree(name)
    ro = load("rollout_" + name)
    A = tree.max_actions
    w = [G.gpu(x) for x in mlp_weights(ro)]
    T, B = ro["logits"].shape[:2]
    logits, value = hip.mlp_forward(hip.mlp_pack(w, A), w[0].shape[0], G.gpu(ro["observations"]), A)
    np.testing.assert_allclose(G.cpu(logits).reshape(T, B, A), ro["logits"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(G.cpu(value).reshape(T, B), ro["values"], rtol=TOL, atol=TOL)


@pytest.mark.parametrize("A,W,N", [(3, 256, 100_001), (2, 64, 31), (5, 256, 4096), (3, 32, 1), (4, 128, 77_777)])
def test_mlp_forward_shapes_vs_oracle_and_torch(G, hip, A, W, N):
    from oracle import oracle

    rng = np.random.default_rng(A * 1000 + W)
    K = 2 * A * A
    shapes = [(W, K), (W,), (1, W), (1,), (W, K), (W,), (A, W), (A,)]
    w = [(rng.standard_normal(s) / np.sqrt(s[-1])).astype(np.float32) for s in shapes]
    x = rng.standard_normal((N, 2, A, A)).astype(np.float32)
    want_l, want_v = oracle.mlp_forward(w, x, A)
    wg = [G.gpu(a) for a in w]
    packed = hip.mlp_pack(wg, A)
    logits, value = hip.mlp_forward(packed, W, G.gpu(x), A)
    np.testing.assert_allclose(G.cpu(logits), want_l, rtol=TOL, atol=TOL)
    np.testing.assert_allclose(G.cpu(value)[:, 0], want_v, rtol=TOL, atol=TOL)
    xt = G.gpu(x).view(N, K)
    tl = torch.relu(xt @ wg[4].T + wg[5]) @ wg[6].T + wg[7]
    np.testing.assert_allclose(G.cpu(logits), G.cpu(tl), rtol=TOL, atol=TOL)
    lh, vh = hip.mlp_forward(packed, W, G.gpu(x).half(), A)  # fp16 observations, fp32 arithmetic
    want_lh, want_vh = oracle.mlp_forward(w, x.astype(np.float16).astype(np.float32), A)
    np.testing.assert_allclose(G.cpu(lh), want_lh, rtol=TOL, atol=TOL)
    np.testing.assert_allclose(G.cpu(vh)[:, 0], want_vh, rtol=TOL, atol=TOL)


@pytest.mark.parametrize("A,W,N", [(3, 256, 50_001), (2, 64, 333), (3, 32, 5), (1, 32, 1000), (3, 128, 20_000), (3, 512, 9_000), (4, 64, 3_000),
                                   (5, 128, 12_345), (5, 256, 4_000), (6, 96, 777), (7, 64, 500), (8, 32, 300)])
def test_mlp_backward_vs_torch_autograd(G, hip, A, W, N):
    """rnad_mlp_backward == autograd through the four Linear layers (fp64 reference on the host for the tolerance)."""
    rng = np.random.default_rng(A * 100 + W + N)
    K = 2 * A * A
    shapes = [(W, K), (W,), (1, W), (1,), (W, K), (W,), (A, W), (A,)]
    w = [(rng.standard_normal(s) / np.sqrt(s[-1])).astype(np.float32) for s in shapes]
    x = rng.standard_normal((N, 2, A, A)).astype(np.float32)
    dl = (rng.standard_normal((N, A)) * (rng.random((N, 1)) < 0.7)).astype(np.float32)
    dv = rng.standard_normal((N, 1)).astype(np.float32)
    wt = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in w]
    xt = torch.tensor(x.reshape(N, K), dtype=torch.float64)
    value = torch.relu(xt @ wt[0].T + wt[1]) @ wt[2].T + wt[3]
    logits = torch.relu(xt @ wt[4].T + wt[5]) @ wt[6].T + wt[7]
    torch.autograd.backward([logits, value], [torch.tensor(dl, dtype=torch.float64), torch.tensor(dv, dtype=torch.float64)])
    wg = [G.gpu(a).requires_grad_(True) for a in w]
    packed = hip.mlp_pack(wg, A)
    lg, vg = hip.FusedMLP.apply(G.gpu(x), A, packed, *wg)
    np.testing.assert_allclose(G.cpu(lg), logits.detach().numpy(), rtol=TOL, atol=TOL)
    torch.autograd.backward([lg, vg], [G.gpu(dl), G.gpu(dv)])
    # a hidden unit whose pre-activation is within fp32 rounding of 0 for some sample has an ambiguous relu gate there:
    # leave those units out of the comparison (and make sure they are rare)
    with torch.no_grad():
        zs = [(xt @ wt[i].T + wt[i + 1]).abs().min(0).values.numpy() for i in (0, 4)]
    ok = [z > 2e-6 for z in zs]
    assert min(o.mean() for o in ok) > 0.5
    rows = {0: ok[0], 1: ok[0], 4: ok[1], 5: ok[1]}
    for i, (got, want, shape) in enumerate(zip(wg, wt, shapes)):
        ref, g = want.grad.numpy(), G.cpu(got.grad)
        if i in rows:
            ref, g = ref[rows[i]], g[rows[i]]
        scale = np.abs(ref).max() + 1e-12
        np.testing.assert_allclose(g, ref, rtol=1e-4, atol=1e-5 * scale, err_msg=str(shape))
    lo, vo = hip.mlp_forward(packed, W, G.gpu(x), A, want_value=False)  # single heads
    assert vo is None and torch.equal(lo, lg.detach())
    lo, vo = hip.mlp_forward(packed, W, G.gpu(x), A, want_logits=False)
    assert lo is None and torch.equal(vo, vg.detach())
    lh, vh = hip.FusedMLP.apply(G.gpu(x).half(), A, packed, *[g.detach().requires_grad_(True) for g in wg])  # fp16 observations
    assert lh.shape == (N, A) and vh.shape == (N, 1)


def test_mlp_backward_lds_transpose_kernel_too():
    """csrc/mlp_bwd.hip (RNAD_MLP_BWD=lds), the kernel the register-resident one replaced, is kept as a cross-check: the choice
    is made once per process, so it is exercised in a child process."""
    import os
    import subprocess
    import sys

    env = dict(os.environ, RNAD_MLP_BWD="lds")
    here = os.path.dirname(os.path.realpath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_hip_parity.py"), "-q", "-x", "-k",
                        "test_mlp_backward_vs_torch_autograd and not lds"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


# ------------------------------------------------------------------------------------------------ rollout driver
def _check_traj(G, traj, ro, T, exact_policy):
    np.testing.assert_array_equal(G.cpu(traj.indices[:T]), ro["indices"])
    assert (G.cpu(traj.indices[T]) == 0).all()
    assert_bits_equal(G.cpu(traj.observations[:T]), ro["observations"], "observations")
    np.testing.assert_array_equal(G.cpu(traj.actions[:T]), ro["actions"].argmax(-1))
    assert_bits_equal(G.cpu(traj.rewards[:T]), ro["rewards"], "rewards")
    assert_bits_equal(G.cpu(traj.values[:T]), ro["values"], "values")
    np.testing.assert_array_equal(G.cpu(traj.mask_bits[:T]), G.mask_bits_of(ro["masks"]))
    if exact_policy:
        assert_bits_equal(G.cpu(traj.policy[:T]), ro["policy"], "policy")
    else:
        np.testing.assert_allclose(G.cpu(traj.policy[:T]), ro["policy"], rtol=TOL, atol=1e-7)
    alive = G.cpu(traj.alive)
    np.testing.assert_array_equal(alive[:T], (ro["indices"] != 0).sum(1))
    assert alive[T] == 0


@pytest.mark.parametrize("name", TREES)
@pytest.mark.parametrize("mode", ("policy", "logits"))
def test_rollout_steps_replay_reference_episode(G, hip, name, mode):
    """Every tensor Episodes.generate records, reproduced step by step from the reference's net outputs and noise."""
    tree, _ = G.golden_tree(name)
    ro = load("rollout_" + name)
    T, B = int(ro["t_eff"]) + 1, ro["indices"].shape[1]
    assert tree.handle().max_depth * 2 == T
    traj = hip.Trajectory(tree.handle(), B, T, G.DEV)
    hip.rollout_begin(tree.handle(), traj)
    for t in range(T):
        kw = dict(policy=G.gpu(ro["policy"][t])) if mode == "policy" else dict(logits=G.gpu(ro["logits"][t]))
        hip.rollout_step(tree.handle(), traj, t, G.gpu(ro["values"][t]), noise_action=G.gpu(ro["noise_action"][t]),
                         noise_chance=G.gpu(ro["noise_chance"][t]), **kw)
    hip.rollout_end(tree.handle(), traj)
    _check_traj(G, traj, ro, T, exact_policy=mode == "policy")


@pytest.mark.parametrize("name", TREES)
def test_episodes_generate_with_a_self_sampling_net(G, name):
    """The generic path: any module returning (logits, policy, value, actions) like reference nn/net.py:37-51."""
    from environment.episode import Episodes

    tree, _ = G.golden_tree(name)
    ro = load("rollout_" + name)
    T, B = int(ro["t_eff"]) + 1, ro["indices"].shape[1]
    net = G.ReplayNet(G.gpu(ro["logits"]), G.gpu(ro["policy"]), G.gpu(ro["values"]), G.gpu(ro["actions"].argmax(-1)))
    ep = Episodes(tree, B)
    ep.generate(net, noise_chance=G.gpu(ro["noise_chance"]))
    assert ep.t_eff == int(ro["t_eff"]) and ep.finished
    _check_traj(G, ep._traj, ro, T, exact_policy=True)
    # the reference's attribute surface
    assert ep.indices.shape == (T, B) and ep.turns.shape == (T, B) and ep.turns.dtype == torch.int64
    np.testing.assert_array_equal(G.cpu(ep.turns), ro["turns"])
    assert_bits_equal(G.cpu(ep.actions), ro["actions"], "one-hot actions")
    assert_bits_equal(G.cpu(ep.masks.contiguous()), ro["masks"], "masks")
    assert ep.q_estimates.shape == ro["policy"].shape and ep.v_estimates.shape == ro["rewards"].shape
    np.testing.assert_allclose(G.cpu(ep.valid_counts), [(ro["indices"][0::2] != 0).sum(), (ro["indices"][1::2] != 0).sum()])


@pytest.mark.parametrize("name", TREES)
def test_seeded_rollout_equals_cpu_oracle_rollout(G, name):
    """Fast path end to end with the seeded noise stream: same logits in -> the same episode as the C oracle, bit for bit;
    and sharding the batch over two 'ranks' by lane offset reproduces the unsharded episode."""
    from environment.episode import Episodes
    from oracle import oracle

    tree, g = G.golden_tree(name)
    ro = load("rollout_" + name)
    A = tree.max_actions
    B, seed = 1000, 4242
    w = mlp_weights(ro)
    T_cap = 2 * tree.handle().max_depth
    ref = oracle.rollout(g, w, B, T_cap, seed, want_logits=True)
    T = ref["T"]
    net = G.ReplayNet(G.gpu(ref["logits"]), None, G.gpu(ref["values"]), fast=True)
    ep = Episodes(tree, B, seed=seed)
    ep.generate(net)
    assert ep.t_eff + 1 == T
    np.testing.assert_array_equal(G.cpu(ep.indices), ref["indices"])
    np.testing.assert_array_equal(G.cpu(ep.action_idx), ref["actions"])
    assert_bits_equal(G.cpu(ep.observations), ref["observations"], "observations")
    assert_bits_equal(G.cpu(ep.rewards), ref["rewards"], "rewards")
    np.testing.assert_allclose(G.cpu(ep.policy), ref["policy"], rtol=TOL, atol=1e-7)
    half = B // 2
    for rank in range(2):
        sl = slice(rank * half, (rank + 1) * half)
        net = G.ReplayNet(G.gpu(ref["logits"][:, sl]), None, G.gpu(ref["values"][:, sl]), fast=True)
        shard = Episodes(tree, half, seed=seed, lane_offset=rank * half)
        shard.generate(net)
        np.testing.assert_array_equal(G.cpu(shard.indices), ref["indices"][:, sl])
        np.testing.assert_array_equal(G.cpu(shard.action_idx), ref["actions"][:, sl])


@pytest.mark.parametrize("name", ("small", "c1"))
def test_real_mlp_rollout_agrees_with_oracle(G, name):
    """With the PyTorch-ROCm MLP in the loop logits differ from the CPU's in the last bits; lanes may then only diverge
    where an argmax is decided by ~1 ulp.  Everything that agrees on its action history must agree exactly."""
    from environment.episode import Episodes
    from oracle import oracle

    tree, g = G.golden_tree(name)
    ro = load("rollout_" + name)
    B, seed = 4096, 7
    net = G.mlp_from(ro, tree.max_actions)
    ep = Episodes(tree, B, seed=seed)
    ep.generate(net)
    ref = oracle.rollout(g, mlp_weights(ro), B, 2 * tree.handle().max_depth, seed)
    assert ep.t_eff + 1 == ref["T"]
    same = (G.cpu(ep.action_idx) == ref["actions"]).all(0) & (G.cpu(ep.indices) == ref["indices"]).all(0)
    assert same.mean() > 0.995
    np.testing.assert_allclose(G.cpu(ep.policy)[:, same], ref["policy"][:, same], rtol=1e-4, atol=1e-6)
    assert_bits_equal(G.cpu(ep.rewards)[:, same], ref["rewards"][:, same], "rewards of agreeing lanes")
    np.testing.assert_allclose(G.cpu(ep.values)[:, same], ref["values"][:, same], rtol=1e-4, atol=1e-5)


def test_states_api(G):
    """States(tree, B).observations() / .step() / .terminal as a user loop would drive them (episode.py:18-125)."""
    from environment.episode import States

    tree, _ = G.golden_tree("small")
    ro = load("rollout_small")
    T, B = int(ro["t_eff"]) + 1, ro["indices"].shape[1]
    st = States(tree, B)
    assert st.indices.dtype == torch.int32 and (st.indices == 1).all() and not st.terminal
    act = ro["actions"].argmax(-1)
    t = 0
    while not st.terminal:
        obs = st.observations()
        assert_bits_equal(G.cpu(obs), ro["observations"][t], f"obs {t}")
        rew = st.step(G.gpu(act[t]), noise=G.gpu(ro["noise_chance"][t]) if t & 1 else None)
        assert_bits_equal(G.cpu(rew), ro["rewards"][t], f"reward {t}")
        t += 1
    assert t == T and (st.player_to_move == 0).all()


# ------------------------------------------------------------------------------------------------ K4 / K5 / K6
def test_process_policy_edge_cases_bit_exact(G, hip):
    g = load("process_policy")
    for key in g:
        if key.startswith("out_"):
            n_disc, eps = key[5:].split("_e")
            out = hip.process_policy(G.gpu(g["policy"]).view(-1, 4), G.gpu(g["mask"]).view(-1, 4), int(n_disc), float(eps))
            assert_bits_equal(G.cpu(out).reshape(g[key].shape), g[key], key)


LEARN = ("c1_eta0.2", "small_eta0", "small_eta0.2", "ragged_eta0.5", "a5_eta0.2")


def _learn(name):
    g = load("learn_" + name)
    ro = load("rollout_" + name.split("_")[0])
    hp = dict(eta=float(g["eta"]), lambda_=1.0, c=float(g.get("hp_c_bar", 1.0)), rho=float(g.get("hp_roh_bar", 1.0)),
              gamma=float(g.get("hp_vtrace_gamma", 1.0)))
    return g, ro, hp, float(g.get("hp_neurd_clip", 1e3)), float(g.get("hp_beta", 2.0))


@pytest.mark.parametrize("name", LEARN)
def test_vtrace_wrapper_vs_reference(G, name):
    import learn.vtrace as vtrace

    g, ro, hp, _, _ = _learn(name)
    A = ro["policy"].shape[-1]
    pid, valid = G.gpu(ro["turns"]), G.gpu(g["valid"])
    assert_bits_equal(G.cpu(vtrace.process_policy(G.gpu(g["pi"]), G.gpu(ro["masks"]), 32, 0.03)), g["pi_processed"], "pi_processed")
    for p in range(2):
        reward = G.gpu(ro["rewards"] if p == 0 else -ro["rewards"])
        vt, has, q = vtrace.v_trace(G.gpu(g["v_target_net"]), valid, pid, G.gpu(ro["policy"]), G.gpu(g["pi_processed"]),
                                    G.gpu(g["log_policy_reg"]), vtrace._player_others(pid, valid, p), G.gpu(ro["actions"]), reward,
                                    p, **hp)
        assert has.dtype == torch.int64 and vt.shape == g[f"v_target_p{p}"].shape
        np.testing.assert_array_equal(G.cpu(has), g[f"has_played_p{p}"])
        if A <= 4:  # torch sums <= 4 terms in index order: same bits
            assert_bits_equal(G.cpu(vt), g[f"v_target_p{p}"], "v_target")
            assert_bits_equal(G.cpu(q), g[f"q_p{p}"], "q")
        else:
            np.testing.assert_allclose(G.cpu(vt), g[f"v_target_p{p}"], rtol=TOL, atol=TOL)
            np.testing.assert_allclose(G.cpu(q), g[f"q_p{p}"], rtol=TOL, atol=TOL)


def test_vtrace_offpolicy_synthetic(G, hip):
    g = load("vtrace_synth")
    T, B, A = g["mu"].shape
    for tag in ("a", "b"):
        hp = json.loads(str(g[f"{tag}_hp"]))
        for p in range(2):
            reward = G.gpu(g["reward"] if p == 0 else -g["reward"])
            for acts in (G.gpu(g["actions"], torch.int32), G.gpu(np.eye(A, dtype=np.float32)[g["actions"]])):
                vt, has, q = hip.vtrace(G.gpu(g["v"]).view(T, B), G.gpu(g["valid"]), G.gpu(g["player_id"], torch.int32), G.gpu(g["mu"]),
                                        G.gpu(g["pi"]), G.gpu(g["logpi_reg"]), acts, reward, p, **hp)
                np.testing.assert_array_equal(G.cpu(has), g[f"{tag}_has_played_p{p}"])
                assert_bits_equal(G.cpu(vt).reshape(T, B, 1), g[f"{tag}_v_target_p{p}"], "v_target")
                assert_bits_equal(G.cpu(q), g[f"{tag}_q_p{p}"], "q")


@pytest.mark.parametrize("name", LEARN + ("synth",))
def test_losses_and_autograd_vs_reference(G, name):
    import learn.vtrace as vtrace

    if name == "synth":
        g = load("vtrace_synth")
        pid, masks, pi = g["player_id"], g["mask"], g["pi"]
        vts, hps, qs = [g["b_v_target_p0"], g["b_v_target_p1"]], [g["b_has_played_p0"], g["b_has_played_p1"]], [g["b_q_p0"], g["b_q_p1"]]
        clip, thr = float(g["nerd_clip"]), float(g["nerd_threshold"])
    else:
        g, ro, _, clip, thr = _learn(name)
        pid, masks, pi = ro["turns"], ro["masks"], g["pi_processed"]
        vts, hps, qs = [g["v_target_p0"], g["v_target_p1"]], [g["has_played_p0"], g["has_played_p1"]], [g["q_p0"], g["q_p1"]]
    v = G.gpu(g["v"]).requires_grad_(True)
    logit = G.gpu(g["logit"]).requires_grad_(True)
    T, B = g["valid"].shape
    loss_v = vtrace.get_loss_v([v] * 2, [G.gpu(x) for x in vts], [G.gpu(x) for x in hps])
    loss_n = vtrace.get_loss_nerd([logit] * 2, [G.gpu(pi)] * 2, [G.gpu(x) for x in qs], G.gpu(g["valid"]), G.gpu(pid), G.gpu(masks),
                                  [torch.ones((T, B, 1), device=G.DEV)] * 2, clip=clip, threshold=thr)
    (loss_v + loss_n).backward()
    np.testing.assert_allclose(loss_v.item(), g["loss_v"], rtol=TOL)
    np.testing.assert_allclose(loss_n.item(), g["loss_nerd"], rtol=TOL, atol=1e-7)
    np.testing.assert_allclose(G.cpu(v.grad), g["dv"], rtol=TOL, atol=1e-8)
    np.testing.assert_allclose(G.cpu(logit.grad), g["dlogit"], rtol=TOL, atol=1e-8)


def _net_logits(g, ro, tag, A):
    from oracle import oracle

    logits, value = oracle.mlp_forward(mlp_weights(g, f"w_{tag}_"), ro["observations"], A)
    return logits.reshape(ro["observations"].shape[:2] + (A,)), value.reshape(ro["observations"].shape[:2])


@pytest.mark.parametrize("name", LEARN)
def test_fused_learner_kernel_vs_reference_learn(G, hip, name):
    """rnad_learn_fused == the reference's RNaD.__learn tensor program: same dL/dlogit, dL/dv, losses, targets."""
    g, ro, hp, clip, thr = _learn(name)
    A = ro["policy"].shape[-1]
    T, B = g["valid"].shape
    ep_args = dict(indices=G.gpu(ro["indices"], torch.int32), mask_bits=G.gpu(G.mask_bits_of(ro["masks"])),
                   actions=G.gpu(ro["actions"].argmax(-1), torch.int32), rewards=G.gpu(ro["rewards"]), mu=G.gpu(ro["policy"]))
    lr, _ = _net_logits(g, ro, "reg", A)
    lr_, _ = _net_logits(g, ro, "reg_", A)
    norm = G.gpu(np.array([g["has_played_p0"].sum(), g["has_played_p1"].sum()], np.float64))
    params = hip.make_learn_params(alpha=float(g["alpha"]), eta=hp["eta"], lambda_=1.0, c=hp["c"], rho=hp["rho"], gamma=hp["gamma"],
                                   clip=clip, threshold=thr)
    dlogit, dv, losses, pi, vt, q = hip.learn_fused(
        logit=G.gpu(g["logit"]), v=G.gpu(g["v"]).view(T, B), v_target_net=G.gpu(g["v_target_net"]).view(T, B), logit_reg=G.gpu(lr),
        logit_reg_=G.gpu(lr_), norm=norm, hp=params, want_aux=True, **ep_args)
    np.testing.assert_allclose(G.cpu(pi), g["pi"], rtol=TOL, atol=1e-7)
    for p in range(2):
        np.testing.assert_allclose(G.cpu(vt[p]), g[f"v_target_p{p}"][..., 0], rtol=1e-5, atol=5e-6)
        np.testing.assert_allclose(G.cpu(q[p]), g[f"q_p{p}"], rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(G.cpu(losses), [g["loss_v"], g["loss_nerd"]], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(G.cpu(dv), g["dv"][..., 0], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(G.cpu(dlogit), g["dlogit"], rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize("name", ("small_eta0.2", "ragged_eta0.5"))
def test_fused_kernel_equals_composition_of_the_single_kernels(G, hip, name):
    """Bit-level: the fused pass is the same arithmetic as policy_head -> process_policy -> v_trace x2 -> losses."""
    g, ro, hp, clip, thr = _learn(name)
    A = ro["policy"].shape[-1]
    T, B = g["valid"].shape
    bits = G.gpu(G.mask_bits_of(ro["masks"]))
    idx = G.gpu(ro["indices"], torch.int32)
    acts = G.gpu(ro["actions"].argmax(-1), torch.int32)
    lr, _ = _net_logits(g, ro, "reg", A)
    lr_, _ = _net_logits(g, ro, "reg_", A)
    logit, v, vtn = G.gpu(g["logit"]), G.gpu(g["v"]).view(T, B), G.gpu(g["v_target_net"]).view(T, B)
    alpha = float(g["alpha"])
    norm = G.gpu(np.array([g["has_played_p0"].sum(), g["has_played_p1"].sum()], np.float64))
    params = hip.make_learn_params(alpha=alpha, eta=hp["eta"], lambda_=1.0, c=hp["c"], rho=hp["rho"], gamma=hp["gamma"], clip=clip,
                                   threshold=thr, w_v=0.7, w_n=1.3)
    dlogit, dv, losses, pi, vt, q = hip.learn_fused(idx, bits, acts, G.gpu(ro["rewards"]), G.gpu(ro["policy"]), logit, v, vtn, G.gpu(lr),
                                                    G.gpu(lr_), norm, params, want_aux=True)
    flat = bits.view(-1)
    pi2, logp = hip.policy_head(logit.view(-1, A), mask_bits=flat, want_log=True)
    _, logr = hip.policy_head(G.gpu(lr).view(-1, A), mask_bits=flat, want_log=True)
    _, logr_ = hip.policy_head(G.gpu(lr_).view(-1, A), mask_bits=flat, want_log=True)
    assert torch.equal(pi.view(-1, A), pi2)
    legal = G.gpu(ro["masks"])
    pip = hip.process_policy(pi2, legal.view(-1, A), 32, 0.03)
    lpol = (logp - (alpha * logr + (1 - alpha) * logr_)).view(T, B, A)
    valid = (idx != 0).float()
    dl2, dv2 = torch.empty_like(dlogit), torch.empty_like(dv)
    loss2 = torch.zeros((2,), dtype=torch.float64, device=G.DEV)
    turn = (torch.arange(T, device=G.DEV) % 2).view(T, 1).expand(T, B)
    for p in range(2):
        rew = G.gpu(ro["rewards"] if p == 0 else -ro["rewards"])
        vt_p, _, q_p = hip.vtrace(vtn, valid, None, G.gpu(ro["policy"]), pip.view(T, B, A), lpol.contiguous(), acts, rew, p, **hp)
        assert torch.equal(vt_p, vt[p]) and torch.equal(q_p, q[p])
        m = (valid * (turn == p)).contiguous()
        hip.loss_v(v.reshape(-1), vt_p.view(-1), m.view(-1), norm[p:p + 1], 0.7, loss2[0:1], dv2.view(-1), p > 0)
        hip.loss_nerd(logit.view(-1, A), pip, q_p.view(-1, A), m.view(-1), legal.view(-1, A), norm[p:p + 1], clip, thr, 1.3, loss2[1:2],
                      dl2.view(-1, A), p > 0)
    assert torch.equal(dv2, dv)
    assert torch.equal(dl2, dlogit)
    np.testing.assert_allclose(G.cpu(loss2), G.cpu(losses), rtol=1e-12)


@pytest.mark.parametrize("name", LEARN)
def test_rnad_learn_step_parameter_gradients(G, name):
    """RNaD.__learn (MLP forwards in PyTorch-ROCm + fused kernel + autograd.backward) == the reference's gradients."""
    from learn.rnad import RNaD

    g, ro, hp, clip, thr = _learn(name)
    tree, _ = G.golden_tree(name.split("_")[0])
    A = tree.max_actions
    rn = RNaD.__new__(RNaD)
    rn.tree, rn.device = tree, G.DEV
    rn.net, rn.net_target = G.mlp_from(g, A, "w_net_"), G.mlp_from(g, A, "w_target_")
    rn.net_reg, rn.net_reg_ = G.mlp_from(g, A, "w_reg_"), G.mlp_from(g, A, "w_reg__")
    rn.eta, rn.c_bar, rn.roh_bar, rn.vtrace_gamma = hp["eta"], hp["c"], hp["rho"], hp["gamma"]
    rn.neurd_clip, rn.beta, rn.grad_clip = clip, thr, 10**3
    rn.value_weight, rn.neurd_weight, rn.epsilon_threshold, rn.n_discrete = 1, 1, 0.03, 32
    ep = G.episodes_from_golden(tree, ro)
    log = {}
    rn._RNaD__learn(ep, float(g["alpha"]), log=log)
    for k, p in rn.net.named_parameters():
        want = g["g_net_" + k.replace(".", "_")]
        scale = np.abs(want).max() + 1e-12
        np.testing.assert_allclose(G.cpu(p.grad), want, rtol=1e-4, atol=2e-6 * scale, err_msg=k)
    np.testing.assert_allclose(log["loss_v"], g["loss_v"], rtol=2e-5)
    np.testing.assert_allclose(log["loss_nerd"], g["loss_nerd"], rtol=1e-4, atol=1e-6)


# ------------------------------------------------------------------------------------------------ NashConv
@pytest.mark.parametrize("name", TREES)
def test_nashconv_vs_reference(G, name):
    from util.metric import NashConvData

    tree, tg = G.golden_tree(name)
    g = load("nashconv_" + name)
    net = G.mlp_from(g, tree.max_actions)
    data = NashConvData(tree)
    data.get_nashconv_from_net(tree, net)
    np.testing.assert_allclose(G.cpu(data.joint_policy), g["joint_policy"], rtol=1e-4, atol=1e-6)
    np.testing.assert_array_equal(G.cpu(data.depth), g["depth"])
    np.testing.assert_allclose(G.cpu(data.row_best), g["row_best"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(G.cpu(data.col_best), g["col_best"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(G.cpu(data.reach_probability), g["reach_probability"], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose((data.row_best[1] + data.col_best[1]).item(), g["nashconv"], rtol=1e-4, atol=1e-5)
    means = data.mean_nashconv_by_depth()
    np.testing.assert_array_equal(sorted(means), g["mean_depths"])
    np.testing.assert_allclose([means[k] for k in sorted(means)], g["mean_values"], rtol=1e-4, atol=1e-5)
    # exact joint policy in -> exact arithmetic out (same op order as the reference's recursion)
    data2 = NashConvData(tree)
    data2.joint_policy = G.gpu(g["joint_policy"])
    data2.get_nashconv(tree, data2.joint_policy)
    np.testing.assert_allclose(G.cpu(data2.row_best), g["row_best"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(G.cpu(data2.col_best), g["col_best"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(G.cpu(data2.reach_probability), g["reach_probability"], rtol=TOL, atol=1e-9)
    # the tree's own solution is a Nash equilibrium
    sol = NashConvData(tree)
    sol.joint_policy = tree.solution_tensor.clone()
    sol.get_nashconv(tree, sol.joint_policy)
    np.testing.assert_allclose(G.cpu(sol.row_best), g["sol_row_best"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(G.cpu(sol.col_best), g["sol_col_best"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(G.cpu(sol.reach_probability), g["sol_reach"], rtol=TOL, atol=1e-9)
    assert abs((sol.row_best[1] + sol.col_best[1]).item()) < 1e-5


@pytest.mark.parametrize("A", (2, 3, 4, 5, 6))
def test_reference_test_nashconv_semantics(G, A):
    """reference tests/test_nashconv.py:13-36 as written: Tree(A, 1, depth 3).generate(); get_nashconv(tree, solution)
    with data.joint_policy still zero -> nashconv == 0 and sum(reach) == 2."""
    from environment.tree import Tree
    from util.metric import NashConvData

    import random

    np.random.seed(A)
    random.seed(A)
    tree = Tree(device=G.DEV, max_actions=A, max_transitions=1, depth_bound=3)
    tree.generate()
    data = NashConvData(tree)
    data.get_nashconv(tree, tree.solution_tensor)
    assert (data.row_best[1] + data.col_best[1]).item() == 0
    # the reference asserts == 2 exactly; mixed solutions such as (1/3, 1/3, 1/3) sum to 2 only up to fp32 rounding
    assert abs(torch.sum(data.reach_probability).item() - 2) < 1e-6


def test_loss_wrappers_take_the_full_signature(G):
    """get_loss_v / get_loss_nerd with a DIFFERENT tensor per list entry and non-trivial per-row importance weights (the reference's only call
    site passes the same tensor twice and weights of one, learn/rnad.py:407-422; r05 asserted exactly that) against the formulas of
    learn/vtrace.py:352-431 evaluated with torch's autograd on the same device, in float64."""
    import learn.vtrace as vtrace

    torch.manual_seed(3)
    T, B, A = 5, 257, 3
    dev = G.DEV
    valid = (torch.rand((T, B), device=dev) > 0.2).float()
    pid = torch.randint(0, 2, (T, B), device=dev)
    legal = (torch.rand((T, B, A), device=dev) > 0.25).float()
    legal[..., 0] = 1.0
    vs = [torch.randn((T, B, 1), device=dev, requires_grad=True) for _ in range(2)]
    vts = [torch.randn((T, B, 1), device=dev) for _ in range(2)]
    hps = [(torch.rand((T, B), device=dev) > 0.5).float() for _ in range(2)]
    logits = [torch.randn((T, B, A), device=dev, requires_grad=True) for _ in range(2)]
    pis = [torch.softmax(torch.randn((T, B, A), device=dev), -1) for _ in range(2)]
    qs = [torch.randn((T, B, A), device=dev) for _ in range(2)]
    isc = [0.5 + torch.rand((T, B, 1), device=dev) for _ in range(2)]
    clip, thr = 0.7, 0.9

    def reference(vs, logits):
        d = torch.float64
        loss = 0.0
        for v, vt, m in zip(vs, vts, hps):
            n = m.to(d).sum()
            loss = loss + (m.to(d).unsqueeze(-1) * (v.to(d) - vt.to(d)) ** 2).sum() / (n + (n == 0))
        loss_n = 0.0
        for k, (lg, pi, q, c) in enumerate(zip(logits, pis, qs, isc)):
            q, pi, lg64, lgl = q.to(d), pi.to(d), lg.to(d), legal.to(d)
            adv = (c.to(d) * (q - (pi * q).sum(-1, keepdim=True))).clamp(-clip, clip).detach()
            z = lg64 - (lg64 * lgl).mean(-1, keepdim=True)
            force = ((z > -thr) * adv.clamp(max=0.0) + (z < thr) * adv.clamp(min=0.0)).detach()
            row = (lgl * z * force).sum(-1)
            m = (valid * (pid == k)).to(d)
            n = m.sum()
            loss_n = loss_n - (row * m).sum() / (n + (n == 0))
        return loss, loss_n

    got_v = vtrace.get_loss_v(vs, vts, hps)
    got_n = vtrace.get_loss_nerd(logits, pis, qs, valid, pid, legal, isc, clip=clip, threshold=thr)
    (got_v + got_n).backward()
    got_grads = [x.grad.clone() for x in vs + logits]
    for x in vs + logits:
        x.grad = None
    want_v, want_n = reference(vs, logits)
    (want_v + want_n).backward()
    np.testing.assert_allclose(got_v.item(), want_v.item(), rtol=TOL)
    np.testing.assert_allclose(got_n.item(), want_n.item(), rtol=TOL, atol=1e-7)
    for got, x in zip(got_grads, vs + logits):
        np.testing.assert_allclose(G.cpu(got), G.cpu(x.grad), rtol=TOL, atol=1e-8)
    with pytest.raises(AssertionError, match="per action"):
        vtrace.get_loss_nerd(logits, pis, qs, valid, pid, legal, [torch.ones((T, B, A), device=dev)] * 2, clip=clip, threshold=thr)
