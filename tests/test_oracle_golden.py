"""Pin the CPU oracle (oracle/rnad_oracle.c) against fixtures captured from the imported reference.

CPU-only.  Integer/index results must be bit-exact; float results are compared bitwise where the
oracle replays the reference's op order (observe, transition rewards, process_policy, v_trace) and
to 1e-5 elsewhere (anything through exp/log or a long reduction).
"""
import json

import numpy as np
import pytest

from _util import TREES, assert_bits_equal, load, load_tree, mlp_weights
from oracle import oracle

TOL = 1e-5


@pytest.mark.parametrize("name", TREES)
def test_observe_and_masks(name):
    tree, ro = load_tree(name), load("rollout_" + name)
    T = int(ro["t_eff"]) + 1
    for t in range(T):
        obs, mask = oracle.observe(tree["expected_value"], tree["legal"], ro["indices"][t], ro["turns"][t])
        assert_bits_equal(obs, ro["observations"][t], f"obs t={t}")  # includes the -0.0 of the column view
        assert_bits_equal(mask, ro["masks"][t], f"mask t={t}")


@pytest.mark.parametrize("name", TREES)
def test_sampler_and_transition_bit_exact(name):
    tree, ro = load_tree(name), load("rollout_" + name)
    T = int(ro["t_eff"]) + 1
    act = ro["actions"].argmax(-1)
    assert (ro["actions"].sum(-1) == 1).all()
    for t in range(T):
        a = oracle.sample(ro["policy"][t], ro["noise_action"][t])
        np.testing.assert_array_equal(a, act[t])
        if t % 2 == 1:
            nxt, rew = oracle.transition(tree["index"], tree["chance"], tree["value"], ro["indices"][t], act[t - 1],
                                         act[t], ro["noise_chance"][t])
            if t + 1 < T:
                np.testing.assert_array_equal(nxt, ro["indices"][t + 1])
            else:
                assert (nxt == 0).all()
            assert_bits_equal(rew, ro["rewards"][t], f"reward t={t}")
        else:
            assert (ro["rewards"][t] == 0).all()
            if t + 1 < T:
                np.testing.assert_array_equal(ro["indices"][t + 1], ro["indices"][t])


@pytest.mark.parametrize("name", TREES)
def test_mlp_and_policy_head(name):
    tree, ro = load_tree(name), load("rollout_" + name)
    A = tree["index"].shape[-1]
    T = int(ro["t_eff"]) + 1
    w = mlp_weights(ro)
    for t in range(T):
        logits, value = oracle.mlp_forward(w, ro["observations"][t], A)
        np.testing.assert_allclose(logits, ro["logits"][t], rtol=TOL, atol=TOL)
        np.testing.assert_allclose(value, ro["values"][t], rtol=TOL, atol=TOL)
        pol, _ = oracle.policy_head(ro["logits"][t], ro["masks"][t])
        np.testing.assert_allclose(pol, ro["policy"][t], rtol=TOL, atol=1e-7)
        assert ((pol == 0) == (ro["masks"][t] == 0)).all()


def test_process_policy_edge_cases():
    g = load("process_policy")
    for key in g:
        if not key.startswith("out_"):
            continue
        n_disc, eps = key[5:].split("_e")
        out = oracle.process_policy(g["policy"], g["mask"], int(n_disc), float(eps))
        assert_bits_equal(out, g[key], key)


LEARN = ("c1_eta0.2", "small_eta0", "small_eta0.2", "ragged_eta0.5", "a5_eta0.2")


def _learn_inputs(name):
    g = load("learn_" + name)
    ro = load("rollout_" + name.split("_")[0])
    hp = dict(eta=float(g["eta"]), lambda_=1.0, c=float(g.get("hp_c_bar", 1.0)), rho=float(g.get("hp_roh_bar", 1.0)),
              gamma=float(g.get("hp_vtrace_gamma", 1.0)))
    clip = float(g.get("hp_neurd_clip", 1e3))
    thr = float(g.get("hp_beta", 2.0))
    return g, ro, hp, clip, thr


@pytest.mark.parametrize("name", LEARN)
def test_learn_policy_head_and_process_policy(name):
    g, ro, *_ = _learn_inputs(name)
    pol, logp = oracle.policy_head(g["logit"], ro["masks"])
    np.testing.assert_allclose(pol, g["pi"], rtol=TOL, atol=1e-7)
    np.testing.assert_allclose(logp, g["log_pi"], rtol=TOL, atol=TOL)
    # process_policy is discontinuous: feed it the reference's own pi bits
    assert_bits_equal(oracle.process_policy(g["pi"], ro["masks"], 32, 0.03), g["pi_processed"], "pi_processed")


@pytest.mark.parametrize("name", LEARN)
def test_vtrace_matches_reference(name):
    g, ro, hp, _, _ = _learn_inputs(name)
    for p in range(2):
        reward = ro["rewards"] if p == 0 else -ro["rewards"]
        vt, has, q = oracle.vtrace(g["v_target_net"], g["valid"], ro["turns"], ro["policy"], g["pi_processed"],
                                   g["log_policy_reg"], ro["actions"], reward, p, **hp)
        np.testing.assert_array_equal(has, g[f"has_played_p{p}"])
        np.testing.assert_allclose(vt, g[f"v_target_p{p}"], rtol=TOL, atol=TOL)
        np.testing.assert_allclose(q, g[f"q_p{p}"], rtol=TOL, atol=TOL)


def test_vtrace_is_bitwise_on_three_actions():
    """With A == 3 torch's inner-dim sums run in index order, so the straight loop is bit-exact."""
    for name in ("small_eta0.2", "c1_eta0.2"):
        g, ro, hp, _, _ = _learn_inputs(name)
        for p in range(2):
            reward = ro["rewards"] if p == 0 else -ro["rewards"]
            vt, _, q = oracle.vtrace(g["v_target_net"], g["valid"], ro["turns"], ro["policy"], g["pi_processed"],
                                     g["log_policy_reg"], ro["actions"], reward, p, **hp)
            assert_bits_equal(vt, g[f"v_target_p{p}"], f"{name} v_target p{p}")
            assert_bits_equal(q, g[f"q_p{p}"], f"{name} q p{p}")


@pytest.mark.parametrize("name", LEARN)
def test_losses_and_closed_form_grads(name):
    g, ro, hp, clip, thr = _learn_inputs(name)
    lv, dv = oracle.loss_v(g["v"], g["v_target_p0"], g["v_target_p1"], g["has_played_p0"], g["has_played_p1"])
    ln, dl = oracle.loss_nerd(g["logit"], g["pi_processed"], g["q_p0"], g["q_p1"], g["valid"], ro["turns"],
                              ro["masks"], clip, thr)
    np.testing.assert_allclose(lv, g["loss_v"], rtol=TOL)
    np.testing.assert_allclose(ln, g["loss_nerd"], rtol=TOL, atol=1e-7)
    np.testing.assert_allclose(dv, g["dv"], rtol=TOL, atol=1e-8)
    np.testing.assert_allclose(dl, g["dlogit"], rtol=TOL, atol=1e-8)


def test_vtrace_offpolicy_synthetic():
    g = load("vtrace_synth")
    a_oh = np.eye(g["mu"].shape[-1], dtype=np.float32)[g["actions"]]
    for tag in ("a", "b"):
        hp = json.loads(str(g[f"{tag}_hp"]))
        for p in range(2):
            reward = g["reward"] if p == 0 else -g["reward"]
            vt, has, q = oracle.vtrace(g["v"], g["valid"], g["player_id"], g["mu"], g["pi"], g["logpi_reg"], a_oh,
                                       reward, p, **hp)
            np.testing.assert_array_equal(has, g[f"{tag}_has_played_p{p}"])
            assert_bits_equal(vt, g[f"{tag}_v_target_p{p}"], f"{tag} vt p{p}")
            assert_bits_equal(q, g[f"{tag}_q_p{p}"], f"{tag} q p{p}")
    lv, dv = oracle.loss_v(g["v"], g["b_v_target_p0"], g["b_v_target_p1"], g["b_has_played_p0"], g["b_has_played_p1"])
    ln, dl = oracle.loss_nerd(g["logit"], g["pi"], g["b_q_p0"], g["b_q_p1"], g["valid"], g["player_id"], g["mask"],
                              float(g["nerd_clip"]), float(g["nerd_threshold"]))
    np.testing.assert_allclose(lv, g["loss_v"], rtol=TOL)
    np.testing.assert_allclose(ln, g["loss_nerd"], rtol=TOL)
    np.testing.assert_allclose(dv, g["dv"], rtol=TOL, atol=1e-8)
    np.testing.assert_allclose(dl, g["dlogit"], rtol=TOL, atol=1e-8)


@pytest.mark.parametrize("name", TREES)
def test_nashconv(name):
    tree, g = load_tree(name), load("nashconv_" + name)
    jp = g["joint_policy"]
    rb, cb, reach, depth = oracle.nashconv(tree["index"], tree["value"], tree["chance"], tree["legal"], jp[1], jp)
    np.testing.assert_array_equal(depth, g["depth"])
    np.testing.assert_allclose(rb, g["row_best"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(cb, g["col_best"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(reach, g["reach_probability"], rtol=TOL, atol=1e-9)
    np.testing.assert_allclose(rb[1] + cb[1], g["nashconv"], rtol=TOL, atol=TOL)
    # the tree's own solution is a Nash equilibrium: NashConv ~ 0 and row_best[root] == root value
    sol = tree["solution"]
    rb, cb, reach, depth = oracle.nashconv(tree["index"], tree["value"], tree["chance"], tree["legal"], sol[1], sol)
    np.testing.assert_allclose(rb, g["sol_row_best"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(cb, g["sol_col_best"], rtol=TOL, atol=TOL)
    assert abs(rb[1] + cb[1]) < 1e-5
    np.testing.assert_allclose(rb[1], tree["root_value"][1, 0], atol=1e-5)


def test_nashconv_reference_test_semantics():
    """reference tests/test_nashconv.py: get_nashconv(tree, solution) with data.joint_policy still zero
    recurses with the zero table (util/metric.py:148-151), giving exactly 0 and total reach 2."""
    for name in ("c1",):
        tree = load_tree(name)
        zeros = np.zeros_like(tree["solution"])
        rb, cb, reach, _ = oracle.nashconv(tree["index"], tree["value"], tree["chance"], tree["legal"],
                                           tree["solution"][1], zeros)
        assert rb[1] + cb[1] == 0
        assert reach.sum() == 2


@pytest.mark.parametrize("name", TREES)
def test_on_policy_step_losses(name):
    """tests/golden/onpolicy_*.npz (make_onpolicy.py): the learner net plays the batch it then learns from.  The oracle's
    composition -- observe -> MLP x4 -> policy heads -> process_policy -> v_trace x2 -> losses -- reproduces the recorded acting
    policy (actor == learner) and the reference's two losses."""
    tree, g = load_tree(name), load("onpolicy_" + name)
    A = tree["index"].shape[-1]
    idx, masks = g["indices"], g["masks"]
    T, B = idx.shape
    turns = np.broadcast_to((np.arange(T) % 2)[:, None], (T, B)).astype(np.int64)
    obs = np.stack([oracle.observe(tree["expected_value"], tree["legal"], idx[t], turns[t])[0] for t in range(T)])
    out = {}
    for tag in ("net", "target", "reg", "reg_"):
        lg, v = oracle.mlp_forward(mlp_weights(g, f"w_{tag}_"), obs.reshape((T * B,) + obs.shape[2:]), A)
        out[tag] = (lg.reshape(T, B, A), v.reshape(T, B, 1))
    pi, log_pi = oracle.policy_head(out["net"][0], masks)
    live = idx != 0
    np.testing.assert_allclose(pi[live], g["policy"][live], rtol=TOL, atol=1e-7)  # on-policy: mu is the learner's pi
    _, log_r = oracle.policy_head(out["reg"][0], masks)
    _, log_r_ = oracle.policy_head(out["reg_"][0], masks)
    alpha = float(g["alpha"])
    lpol = log_pi - (np.float32(alpha) * log_r + np.float32(1 - alpha) * log_r_)
    pip = oracle.process_policy(pi, masks, 32, 0.03)
    valid = live.astype(np.float32)
    a_oh = np.eye(A, dtype=np.float32)[g["actions"]]
    hp = dict(eta=float(g["eta"]), lambda_=1.0, c=float(g.get("hp_c_bar", 1.0)), rho=float(g.get("hp_roh_bar", 1.0)),
              gamma=float(g.get("hp_vtrace_gamma", 1.0)))
    vt, has, q = [], [], []
    for p in range(2):
        rew = g["rewards"] if p == 0 else -g["rewards"]
        a, b, c = oracle.vtrace(out["target"][1], valid, turns, g["policy"], pip, lpol, a_oh, rew, p, **hp)
        vt.append(a), has.append(b), q.append(c)
    lv, _ = oracle.loss_v(out["net"][1], vt[0], vt[1], has[0], has[1])
    ln, _ = oracle.loss_nerd(out["net"][0], pip, q[0], q[1], valid, turns, masks, float(g.get("hp_neurd_clip", 1e3)), float(g.get("hp_beta", 2.0)))
    np.testing.assert_allclose(lv, g["loss_v"], rtol=1e-4)
    np.testing.assert_allclose(ln, g["loss_nerd"], rtol=1e-3, atol=1e-6)


@pytest.mark.parametrize("name", TREES)
def test_torch_transcription_gives_the_reference_gradients(name):
    """oracle/torch_port.py (bench.py's cpu_baseline.torch_cpu leg: the reference's op sequence in plain torch on the host cores) on the
    on-policy fixtures: the reference's own losses and parameter gradients; its rollout plays legal episodes of the tree."""
    import torch

    from oracle import torch_port as tp

    tree, g = load_tree(name), load("onpolicy_" + name)
    A = tree["index"].shape[-1]
    W = int(g["width"])
    nets = []
    for tag in ("net", "target", "reg", "reg_"):
        n = tp.TorchMLP(A, W)
        n.load_state_dict({k: torch.as_tensor(g[f"w_{tag}_" + k.replace(".", "_")]) for k in n.state_dict()})
        nets.append(n)
    tt = {k: torch.as_tensor(tree[k]) for k in ("index", "value", "chance", "expected_value", "legal")}
    tt["index"] = tt["index"].long()
    idx = torch.as_tensor(g["indices"]).long()
    T, B = idx.shape
    ep = dict(indices=idx, observations=torch.stack([tp.observe(tt, idx[t], t & 1) for t in range(T)]), policy=torch.as_tensor(g["policy"]),
              actions=torch.nn.functional.one_hot(torch.as_tensor(g["actions"]).long(), A).float(), rewards=torch.as_tensor(g["rewards"]),
              masks=torch.as_tensor(g["masks"]))
    assert torch.equal(ep["observations"][:, :, 1, :, 0], ep["masks"])
    hp = dict(tp.HP, eta=float(g["eta"]), c=float(g.get("hp_c_bar", 1.0)), rho=float(g.get("hp_roh_bar", 1.0)), gamma=float(g.get("hp_vtrace_gamma", 1.0)),
              clip=float(g.get("hp_neurd_clip", tp.HP["clip"])), threshold=float(g.get("hp_beta", 2.0)))
    loss, lv, ln = tp.learn_losses(nets, ep, float(g["alpha"]), hp)
    loss.backward()
    np.testing.assert_allclose(float(lv.detach()), g["loss_v"], rtol=2e-5)
    np.testing.assert_allclose(float(ln.detach()), g["loss_nerd"], rtol=1e-4, atol=1e-6)
    for k, p in nets[0].named_parameters():
        want = g["g_net_" + k.replace(".", "_")]
        np.testing.assert_allclose(p.grad.numpy(), want, rtol=1e-4, atol=2e-6 * np.abs(want).max(), err_msg=k)
    # the rollout leg: every transition it plays is one of the tree's, rewards only on the step into state 0
    torch.manual_seed(0)
    played = tp.play(tt, nets[0], 64, 2 * int(tree["depth_bound"]) if "depth_bound" in tree else 16)
    st, rew = played["indices"], played["rewards"]
    assert (st[0] == 1).all() and ((rew != 0).sum(0) <= 1).all()
    for t in range(1, st.shape[0]):
        moved = st[t] != st[t - 1]
        assert not moved[(t & 1) == 1].any() if (t & 1) == 1 else True  # the row player's step leaves the state as it is
        kids = tt["index"][st[t - 1]].reshape(st.shape[1], -1)
        assert ((kids == st[t].unsqueeze(-1)).any(-1) | ~moved).all()
