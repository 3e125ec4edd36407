"""rnad_mlp_rows_records (csrc/mlp_rows.hip): the table forwards of a tabular update (learn/rnad.py:373,378 on the 2S rows of the tree) and
rnad_bucket_records in one launch.  Checked against the two launches it replaces: the net outputs against k_mlp_forward's (another order of
the second-layer sums: a tolerance), the records against rnad_bucket_records evaluated on the fused kernel's OWN net outputs (the same
function of the same floats: bit for bit) -- on small trees at several widths, with and without the legal fold, with a row list and the
logits taken from a table (the lazy-rows variant), at the full configs[1] size and on a table of many chunks per workgroup."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")


def _nets(A, width, seed):
    from nn.net import MLP

    torch.manual_seed(seed)
    return [MLP(A, width, device=DEV) for _ in range(4)]  # learner, target, reg, reg_


def _hp(hip, alpha=0.3):
    return hip.make_learn_params(alpha=alpha, eta=0.2, clip=1e3, threshold=2.0, eps_threshold=0.03, n_disc=16)


def _reference(hip, h, nets, W, table, A, fold, hp):
    """What the fused launch replaces: k_mlp_forward (learner: both heads, target: value head) and the regularisation tables."""
    packs = hip.mlp_pack_many([n._weights() for n in nets], A, fold=bool(fold))
    outs = hip.mlp_forward_multi(packs[:2], W, table, A, [(True, True), (False, True)], fold=h if fold else False)
    regs = hip.mlp_forward_multi(packs[2:], W, table, A, [(True, False), (True, False)], fold=h if fold else False)
    return packs, outs[0][0], outs[0][1], outs[1][1], regs[0][0], regs[1][0]


def _check(hip, h, width, fold, half=False, alpha=0.3, seed=11):
    A = h.A
    nets = _nets(A, width, seed)
    table = h.observations_table(half)
    hp = _hp(hip, alpha)
    packs, logit, v, vt, lr, lr2 = _reference(hip, h, nets, width, table, A, fold, hp)
    if not hip.mlp_rows_records_supported(A, width, fold):
        pytest.skip("shape takes the two launches")
    out = hip.mlp_rows_records(h, packs[0], packs[1], width, table, lr, lr2, hp, fold=h if fold else False)
    for name, a, b in (("logit", out["logit"], logit), ("v", out["v"], v), ("v_target", out["v_target"], vt)):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-5, atol=3e-6, err_msg=name)
    # the records: rnad_bucket_records on the fused kernel's own outputs, bit for bit (row_records.hpp is shared)
    rec, fast = hip.bucket_records(h, out["logit"], out["v"], out["v_target"], lr, lr2, hp, fast=True)
    assert torch.equal(out["records"].view(torch.int32), rec.view(torch.int32)), "records differ from rnad_bucket_records"
    assert torch.equal(out["fast_records"].view(torch.int32), fast.view(torch.int32)), "fast records differ"
    assert torch.equal(out["records"]._policy_rows.view(torch.int32), rec._policy_rows.view(torch.int32)), "policy rows differ"
    return nets, packs, out, (lr, lr2, hp, table)


@pytest.mark.parametrize("fold", (True, False))
@pytest.mark.parametrize("width", (32, 64, 256))
@pytest.mark.parametrize("name", ("ternary4", "a5c4", "binary", "pruned"))
def test_fused_rows_forward_and_records(name, width, fold):
    import rnad_hip as hip
    from test_hip_bucket import TREES, _native_tree

    tree = _native_tree(**TREES[name])
    h = tree.handle()
    if fold and not h.legal_foldable:
        pytest.skip("not foldable")
    _check(hip, h, width, fold)


def test_fused_rows_fp16_observations_and_step_params():
    import rnad_hip as hip
    from test_hip_bucket import TREES, _native_tree

    tree = _native_tree(**TREES["ternary4"])
    h = tree.handle()
    nets, packs, out, (lr, lr2, hp, table) = _check(hip, h, 256, True, half=True)
    # alpha from device memory (the captured step): the same records as with alpha in the struct
    sp = torch.zeros((2,), dtype=torch.int64, device=DEV)
    hip.step_params_set(sp, 77, 0.3)
    other = hip.make_learn_params(alpha=0.9, eta=0.2, clip=1e3, threshold=2.0, eps_threshold=0.03, n_disc=16)
    got = hip.mlp_rows_records(h, packs[0], packs[1], 256, table, lr, lr2, other, step_params=sp, fold=h)
    assert torch.equal(got["fast_records"].view(torch.int32), out["fast_records"].view(torch.int32))
    assert torch.equal(got["records"].view(torch.int32), out["records"].view(torch.int32))


# split: RNAD_MLP_SPLIT -- None: the library's choice (r06: the split-precision first layer on rows staged in LDS where the first layer has
# more than 16 input features, i.e. on "a5c4"; the fp32 chains at A = 3), "1" / "0": forced on (wherever a build without scratch exists) / off
@pytest.mark.parametrize("split", (None, "1", "0"))
@pytest.mark.parametrize("name", ("ternary4", "a5c4"))
def test_fused_rows_with_a_row_list_and_the_logits_from_a_table(name, split, monkeypatch):
    """The lazy-rows variant: a staged actor wrote the learner's logits; the two value heads and the records on the listed rows only."""
    import rnad_hip as hip

    if split is not None:
        monkeypatch.setenv("RNAD_MLP_SPLIT", split)
    from test_hip_bucket import TREES, _native_tree

    tree = _native_tree(**TREES[name])
    h = tree.handle()
    A, W, S = h.A, 256, h.S
    nets = _nets(A, W, 5)
    table = h.observations_table()
    hp = _hp(hip)
    packs, logit, v, vt, lr, lr2 = _reference(hip, h, nets, W, table, A, True, hp)
    flags = torch.zeros((2 * S,), dtype=torch.int32, device=DEV)
    flags[::3] = 1
    flags[1] = 1
    rows = hip.compact_valid(flags)
    sel = flags.bool()
    out = hip.mlp_rows_records(h, packs[0], packs[1], W, table, lr, lr2, hp, fold=h, rows=rows, logit_tab=logit)
    assert out["logit"] is logit and out["policy_rows"] is None
    np.testing.assert_allclose(out["v"][sel].cpu().numpy(), v[sel].cpu().numpy(), rtol=1e-5, atol=3e-6)
    np.testing.assert_allclose(out["v_target"][sel].cpu().numpy(), vt[sel].cpu().numpy(), rtol=1e-5, atol=3e-6)
    rec, fast = hip.bucket_records(h, logit, out["v"], out["v_target"], lr, lr2, hp, fast=True, rows=rows)
    assert torch.equal(out["records"][sel].view(torch.int32), rec[sel].view(torch.int32))
    assert torch.equal(out["fast_records"][sel].view(torch.int32), fast[sel].view(torch.int32))
    # all rows listed, logits still from the table: the value heads of the full variant
    if not hip.mlp_rows_records_supported(A, W, True):
        return
    every = hip.compact_valid(torch.ones((2 * S,), dtype=torch.int32, device=DEV))
    full = hip.mlp_rows_records(h, packs[0], packs[1], W, table, lr, lr2, hp, fold=h)
    part = hip.mlp_rows_records(h, packs[0], packs[1], W, table, lr, lr2, hp, fold=h, rows=every, logit_tab=full["logit"])
    assert torch.equal(part["v"], full["v"]) and torch.equal(part["v_target"], full["v_target"])
    assert torch.equal(part["fast_records"].view(torch.int32), full["fast_records"].view(torch.int32))


@pytest.mark.parametrize("depth,fold,split", ((6, True, None), (7, True, None), (6, False, None), (6, True, "1"), (7, True, "1")))
def test_fused_rows_full_size(depth, fold, split, monkeypatch):
    """configs[1] (2S = 132 862 rows: every workgroup's rows in one chunk) and a depth-7 tree (1.2 M rows: many chunks per workgroup);
    split "1": the split-precision first layer forced on at A = 3 (8-wave workgroups, rows staged in LDS)."""
    import rnad_hip as hip

    if split is not None:
        monkeypatch.setenv("RNAD_MLP_SPLIT", split)
    from test_hip_bucket import _native_tree

    tree = _native_tree(A=3, C=1, depth=depth, seed=0)
    h = tree.handle()
    assert 2 * h.S == (132862 if depth == 6 else 1195744)
    _check(hip, h, 256, fold)


def test_default_step_uses_the_fused_launch_and_trains_like_the_two_launches(monkeypatch):
    """RNaD's default step with and without the fused launch: same episodes (the actor's policy rows agree to rounding, the draws are
    inverse-CDF: compare the updates statistically through the gradients), and the tables of one step directly."""
    import rnad_hip as hip
    from learn.rnad import RNaD
    from test_hip_bucket import _native_tree
    import tempfile, os

    tree = _native_tree(A=3, C=1, depth=4, seed=0)
    os.environ["RNAD_SAVE_DIR"] = tempfile.mkdtemp(prefix="rnad_rows_")
    grads = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("RNAD_FUSED_ROWS", fused)
        torch.manual_seed(0)
        rn = RNaD(tree=tree, device=DEV, directory_name="rows" + fused, batch_size=1 << 14, eta=0.2, b1_adam=0.0, lr=1e-3,
                  net_params={"type": "MLP", "max_actions": 3, "width": 256})
        rn.initialize()
        rn.use_graph = False
        rn.fused_optimizer = False
        from environment.episode import Buffer

        buf = Buffer(1)
        calls = []
        orig = hip.mlp_rows_records

        def spy(*a, **k):
            calls.append(1)
            return orig(*a, **k)

        monkeypatch.setattr(hip, "mlp_rows_records", spy)
        seen = []
        step = rn.optimizer.step
        rn.optimizer.step = lambda *a_, **k_: (seen.append([p.grad.detach().clone() for p in rn.net.parameters()]), step(*a_, **k_))[1]
        rn._step_body(buf, 0.5)
        monkeypatch.setattr(hip, "mlp_rows_records", orig)
        assert len(calls) == (1 if fused == "1" else 0)
        grads[fused] = seen[0]
    # same weights, same seed: the two paths differ in the rounding of the second-layer sums only (a draw that flips on it changes single
    # episodes of 16 384, not the gradient's direction)
    for a, b in zip(grads["1"], grads["0"]):
        scale = b.abs().max().item() + 1e-12
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=2e-2 * scale)


@pytest.mark.parametrize("fold", (True, False))
@pytest.mark.parametrize("width", (64, 256))
@pytest.mark.parametrize("name", ("ternary4", "a5c4", "binary", "pruned"))
def test_rows_actor_is_the_forward_actor(name, width, fold, monkeypatch):
    """rnad_mlp_rows_actor (the staged actor's launch with the mapping of csrc/mlp_rows.hip) against rnad_mlp_forward_actor: logits to
    rounding, the policy rows the policy head of its OWN logits bit for bit (rnad_policy_head: the same function), with a row list (only
    the listed rows are written) and on all rows."""
    import rnad_hip as hip
    from test_hip_bucket import TREES, _native_tree

    tree = _native_tree(**TREES[name])
    h = tree.handle()
    A, S = h.A, h.S
    if fold and not h.legal_foldable:
        pytest.skip("not foldable")
    if not hip.lib().rnad_mlp_rows_actor_supported(A, width, int(fold)):
        pytest.skip("shape takes rnad_mlp_forward_actor")
    net = _nets(A, width, 21)[0]
    table = h.observations_table()
    packed = hip.mlp_pack_many([net._weights()], A, fold=fold)[0]
    stride = int(hip.lib().rnad_bucket_policy_row_stride(A))
    flags = torch.zeros((2 * S,), dtype=torch.int32, device=DEV)
    flags[::5] = 1
    flags[3] = 1
    for rows in (None, hip.compact_valid(flags)):
        out = {}
        for fused in ("1", "0"):
            monkeypatch.setenv("RNAD_ROWS_ACTOR", fused)
            logit = torch.full((2 * S, A), 7.0, device=DEV)
            pol = torch.full((2 * S, stride), 7.0, device=DEV)
            hip.mlp_forward_actor(h, packed, width, table, logit, pol, rows=rows, fold=h if fold else False)
            out[fused] = (logit, pol)
        sel = flags.bool() if rows is not None else torch.ones((2 * S,), dtype=torch.bool, device=DEV)
        np.testing.assert_allclose(out["1"][0][sel].cpu().numpy(), out["0"][0][sel].cpu().numpy(), rtol=1e-5, atol=3e-6)
        np.testing.assert_allclose(out["1"][1][sel].cpu().numpy(), out["0"][1][sel].cpu().numpy(), rtol=1e-5, atol=3e-6)
        assert (out["1"][0][~sel] == 7.0).all() and (out["1"][1][~sel] == 7.0).all(), "rows that are not listed must not be written"
        mask = table[:, 1, :, 0].float().contiguous()  # the mover's legal actions: legal[a][0] of its view of the state
        want = hip.policy_head(out["1"][0].contiguous(), mask=mask)
        got = out["1"][1][:, :A]
        assert torch.equal(got[sel], want[sel]), "policy rows must be the policy head of the kernel's own logits"
        assert (out["1"][1][sel][:, A:] == 0).all()
