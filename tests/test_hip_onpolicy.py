"""The kernels the default step runs -- rnad_bucket_records -> rnad_learn_bucketed_compact (k_bucket_learn<A, true, .> on the row's fast
record) -> rnad_mlp_backward over the 2S rows -- pinned DIRECTLY against the reference: tests/golden/onpolicy_*.npz are batches the
reference's learner net played itself (Episodes.generate(net), then RNaD.__learn on that batch: the on-policy step of
learn/rnad.py:502-510) with the reference's parameter gradients and losses; run_c1.npz is the reference's own six-step RNaD.run.
The recorded trajectory is put into the compact layout on the host (states, 3 bits of action per step, one reward per lane, lanes
stably sorted by bucket) exactly as rnad_rollout_bucketed_compact leaves it."""
import numpy as np
import pytest
import torch

from _util import load, mlp_weights

pytestmark = pytest.mark.gpu

ONPOLICY = ("c1", "small", "ragged", "a5")


def _bare_rnad(G, tree, nets, hp):
    from learn.rnad import RNaD

    rn = RNaD.__new__(RNaD)
    rn.tree, rn.device = tree, G.DEV
    rn.net, rn.net_target, rn.net_reg, rn.net_reg_ = nets
    rn.eta, rn.c_bar, rn.roh_bar, rn.vtrace_gamma = hp["eta"], hp["c"], hp["rho"], hp["gamma"]
    rn.neurd_clip, rn.beta, rn.grad_clip = hp["clip"], hp["thr"], 10**3
    rn.value_weight, rn.neurd_weight, rn.epsilon_threshold, rn.n_discrete = 1, 1, 0.03, 32
    rn.tabular, rn.tabular_gate = True, 0
    return rn


def _default_path_update(G, rn, tree, indices, actions, rewards, alpha, log, fold=False):
    """What RNaD._step_body does around its rollout, with a recorded batch in the rollout's place: the nets on the 2S rows, the row
    records (their pi columns are the actor), the compact batch, __learn."""
    import rnad_hip

    ep, perm = G.compact_episodes_from_recorded(tree, indices, actions, rewards)
    tables = rn._table_outputs(alpha, False, want_target_logits=log is not None, fold=fold)  # fold: the legal fold of DESIGN.md section 5.3
    tables["records"], tables["fast_records"] = rnad_hip.bucket_records(
        tree.handle(), tables["logit"], tables["v"], tables["v_target"], tables["logit_reg"], tables["logit_reg_"], rn._learn_params(alpha), fast=True)
    ep._compact = (ep._compact[0], tables["records"])
    for p in rn.net.parameters():
        p.grad = None
    rn._RNaD__learn(ep, alpha, log=log, tables=tables)
    assert ep._compact is not None and ep.buckets is not None
    return ep, perm, tables


def _hp_of(g):
    return dict(eta=float(g["eta"]), c=float(g.get("hp_c_bar", 1.0)), rho=float(g.get("hp_roh_bar", 1.0)), gamma=float(g.get("hp_vtrace_gamma", 1.0)),
                clip=float(g.get("hp_neurd_clip", 1e3)), thr=float(g.get("hp_beta", 2.0)))


def _pad_width(w, W):
    """The eight Linear tensors of a width-w MLP embedded in a width-W one: the extra hidden units have zero weights and biases,
    so they output relu(0) = 0, contribute exactly 0 to both heads and receive exactly zero gradients (z > 0 is false)."""
    vw0, vb0, vw1, vb1, pw0, pb0, pw1, pb1 = w
    n = vw0.shape[0]

    def rows(x):
        out = np.zeros((W,) + x.shape[1:], x.dtype)
        out[:n] = x
        return out

    def cols(x):
        out = np.zeros((x.shape[0], W), x.dtype)
        out[:, :n] = x
        return out

    return [rows(vw0), rows(vb0), cols(vw1), vb1, rows(pw0), rows(pb0), cols(pw1), pb1]


def _mlp(G, weights, A):
    from nn.net import MLP
    from oracle import oracle

    net = MLP(A, weights[0].shape[0], device=G.DEV)
    net.load_state_dict(dict(zip(oracle.MLP_KEYS, [torch.as_tensor(x) for x in weights])))
    return net


@pytest.mark.parametrize("fold", (False, True))
@pytest.mark.parametrize("logged", (False, True))
@pytest.mark.parametrize("level", (None, 40, 8, 2))
@pytest.mark.parametrize("name", ONPOLICY)
def test_compact_default_path_gives_the_reference_gradients_on_its_own_on_policy_batch(name, level, logged, fold, monkeypatch):
    """level: the table size the planner picks for these small batches (None: one group, nothing above it) or a forced one (cuts at
    every depth: upper rows, shared group roots, several items per bucket).  logged: the LOSSES variant of the learner kernel.
    fold: the table forwards and the backward through the legal-fold instantiations of the MLP kernels (what the default step uses
    wherever the tree allows it)."""
    import _gpu as G
    import rnad_hip

    g = load("onpolicy_" + name)
    tree, _ = G.golden_tree(name)
    A = tree.max_actions
    B = g["indices"].shape[1]
    if level is not None:
        monkeypatch.setenv("RNAD_BUCKET_ROWS", str(level))
        if rnad_hip.bucket_plan(tree.handle(), B) is None:
            pytest.skip(f"a table of {level} rows does not fit this tree / the LDS with A = {A}")
    if fold and not tree.handle().legal_foldable:
        pytest.skip("this tree's legal planes differ from state to state: no fold")
    hp = _hp_of(g)
    nets = [G.mlp_from(g, A, f"w_{tag}_") for tag in ("net", "target", "reg", "reg_")]
    rn = _bare_rnad(G, tree, nets, hp)
    rn.fold_legal = fold
    log = {} if logged else None
    ep, perm, tables = _default_path_update(G, rn, tree, g["indices"], g["actions"], g["rewards"], float(g["alpha"]), log, fold=fold)
    # the premise of the compact learner: the acting policy the reference recorded IS the learner's pi of the slot's row
    T, S = g["indices"].shape[0], tree.handle().S
    rows = g["indices"].astype(np.int64) + (np.arange(T) % 2)[:, None] * S
    col = rnad_hip.policy_column(A)
    pi_rows = G.cpu(tables["records"])[:, col:col + A][rows]
    live = g["indices"] != 0
    np.testing.assert_allclose(pi_rows[live], g["policy"][live], rtol=1e-5, atol=1e-7, err_msg="actor == learner policy rows")
    for k, p in rn.net.named_parameters():
        want = g["g_net_" + k.replace(".", "_")]
        scale = np.abs(want).max() + 1e-12
        np.testing.assert_allclose(G.cpu(p.grad), want, rtol=1e-4, atol=2e-6 * scale, err_msg=k)
    if logged:
        np.testing.assert_allclose(log["loss_v"], g["loss_v"], rtol=2e-5)
        np.testing.assert_allclose(log["loss_nerd"], g["loss_nerd"], rtol=1e-4, atol=1e-6)
        # the dense fields rnad_bucket_expand derives from the compact batch are the recorded ones
        sel = torch.as_tensor(perm, device=G.DEV)
        live_t = torch.as_tensor(live, device=G.DEV)[:, sel]
        assert torch.equal(ep.action_idx[live_t].cpu(), torch.as_tensor(g["actions"][:, perm][live[:, perm]], dtype=torch.int32))
        assert torch.equal(ep.rewards.cpu(), torch.as_tensor(g["rewards"][:, perm]))
        assert torch.equal(ep.masks.cpu(), torch.as_tensor(g["masks"][:, perm]))


@pytest.mark.parametrize("fold", (False, True))
@pytest.mark.parametrize("step", range(6))
def test_compact_default_path_on_every_step_of_the_reference_run(step, fold):
    """tests/golden/run_c1.npz: the reference's RNaD.run (2 regularisation updates x 3 steps, B = 64, on-policy).  Step i with the
    nets the reference had going into it (learner = actor: s{i}_w_actor; target: the EMA after step i - 1; reg / reg_: as rotated) and
    the batch it recorded -> the gradients it recorded (rnad.py:456: after clip_grad_norm_, which does not bind at 10^3).  The run's
    MLP is 16 wide; the fused kernels take multiples of 32, so it is embedded in a 32-wide net (exact: see _pad_width)."""
    import _gpu as G

    g = load("run_c1")
    tree, _ = G.golden_tree("c1")
    A, i = 2, step
    prev = f"s{i - 1}_net_target_" if i else "w0_"
    nets = [_mlp(G, _pad_width(mlp_weights(g, prefix), 32), A)
            for prefix in (f"s{i}_w_actor_", prev, f"s{i}_net_reg_", f"s{i}_net_reg__")]
    rn = _bare_rnad(G, tree, nets, dict(eta=float(g["eta"]), c=1.0, rho=1.0, gamma=1.0, clip=1e3, thr=2.0))
    rn.fold_legal = fold
    assert tree.handle().legal_foldable
    _default_path_update(G, rn, tree, g[f"s{i}_indices"], g[f"s{i}_actions"].argmax(-1), g[f"s{i}_rewards"], float(g[f"s{i}_alpha"]), None,
                         fold=fold)
    for k, p in rn.net.named_parameters():
        want = g[f"s{i}_grads_" + k.replace(".", "_")]
        got = G.cpu(p.grad)
        real = got[:16] if k.endswith("fc0.weight") or k.endswith("fc0.bias") else (got[:, :16] if k.endswith("fc1.weight") else got)
        pad = got[16:] if k.endswith("fc0.weight") or k.endswith("fc0.bias") else (got[:, 16:] if k.endswith("fc1.weight") else got[:0])
        assert (pad == 0).all(), f"{k}: the padded hidden units must receive exactly zero gradients"
        scale = np.abs(want).max() + 1e-12
        np.testing.assert_allclose(real, want, rtol=1e-4, atol=2e-6 * scale, err_msg=k)
