"""The legal fold (include/rnad_hip.h): on trees whose observation rows all carry the same legal plane the FOLD instantiations of the
fused MLP kernels evaluate the same function of the same weights from A^2 + 1 input features -- compared here with the plain kernels
row for row (absorbing-state rows included), forward and backward, and through RNaD's default step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")
FOLDABLE = ("ternary4", "a5c4", "binary", "pruned")


def _setup(name, width=64, seed=3):
    import rnad_hip
    from nn.net import MLP
    from test_hip_bucket import TREES, _native_tree

    tree = _native_tree(**TREES[name])
    h = tree.handle()
    assert h.legal_foldable, "the native trees give every state the full action set"
    torch.manual_seed(seed)
    net = MLP(tree.max_actions, width, device=DEV)
    return tree, h, net, rnad_hip


@pytest.mark.parametrize("half", (False, True))
@pytest.mark.parametrize("name", FOLDABLE)
def test_folded_forward_is_the_plain_forward(name, half):
    tree, h, net, hip = _setup(name)
    A, W = tree.max_actions, net.width
    table = h.observations_table(half)
    plain = hip.mlp_forward(net.pack(), W, table, A)
    packed = hip.mlp_pack_many([net._weights()], A, fold=True)[0]
    assert packed.numel() == hip.mlp_packed_size(A, W, fold=True)
    got = hip.mlp_forward(packed, W, table, A, fold=h)
    for a, b in zip(got, plain):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-5, atol=2e-6)
    # the two rows of the absorbing state are the ones whose legal plane differs (e0): they go through the indicator feature
    S = h.S
    legal0 = table[0, 1].float().reshape(-1)
    assert legal0[0] == 1 and legal0[1:].sum() == 0
    for r in (0, S):
        np.testing.assert_allclose(got[0][r].cpu().numpy(), plain[0][r].cpu().numpy(), rtol=1e-5, atol=2e-6)
    # several nets in one launch, and a row list
    other = type(net)(A, W, device=DEV)
    packs = hip.mlp_pack_many([net._weights(), other._weights()], A, fold=True)
    multi = hip.mlp_forward_multi(packs, W, table, A, [(True, True), (False, True)], fold=h)
    assert torch.equal(multi[0][0], got[0]) and torch.equal(multi[0][1], got[1]) and multi[1][0] is None
    np.testing.assert_allclose(multi[1][1].cpu().numpy(), hip.mlp_forward(other.pack(), W, table, A)[1].cpu().numpy(), rtol=1e-5, atol=2e-6)
    flags = torch.zeros((2 * S,), dtype=torch.int32, device=DEV)
    flags[::3] = 1
    rows = hip.compact_valid(flags)
    part = hip.mlp_forward(packed, W, table, A, live=rows, fold=h)
    sel = flags.bool()
    assert torch.equal(part[0][sel], got[0][sel]) and torch.equal(part[1][sel], got[1][sel]) and (part[0][~sel] == 0).all()
    # the premise is a property of the TREE's table: any other observations are refused, loudly
    with pytest.raises(hip.RnadHipError, match="legal_foldable"):
        hip.mlp_forward(packed, W, table.clone(), A, fold=h)
    with pytest.raises(hip.RnadHipError, match="legal_foldable"):
        hip.mlp_forward(packed, W, table, A, fold=True)


@pytest.mark.parametrize("name", FOLDABLE)
def test_folded_backward_gives_the_gradients_of_the_original_tensors(name):
    tree, h, net, hip = _setup(name, seed=5)
    A, W, S = tree.max_actions, net.width, h.S
    table = h.observations_table()
    g = torch.Generator(device=DEV)
    g.manual_seed(1)
    dl = torch.randn((2 * S, A), device=DEV, generator=g)
    dv = torch.randn((2 * S, 1), device=DEV, generator=g)  # (non-zero on the absorbing rows too: the indicator column's gradient matters)
    weights = net._weights()
    want = hip.mlp_backward(net.pack(), weights, table, A, dl, dv)
    packed = hip.mlp_pack_many([weights], A, fold=True)[0]
    got = hip.mlp_backward(packed, weights, table, A, dl, dv, fold=h)
    for name_, a, b in zip(hip.MLP_KEYS, got, want):
        scale = b.abs().max().item() + 1e-12
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-5, atol=2e-6 * scale, err_msg=name_)
    # against autograd through the reference formulation of the net (float64)
    x = table.reshape(2 * S, -1).double()
    ws = [w.detach().double().requires_grad_(True) for w in weights]
    value = torch.relu(x @ ws[0].T + ws[1]) @ ws[2].T + ws[3]
    logits = torch.relu(x @ ws[4].T + ws[5]) @ ws[6].T + ws[7]
    ((value * dv.double()).sum() + (logits * dl.double()).sum()).backward()
    for name_, a, w in zip(hip.MLP_KEYS, got, ws):
        scale = w.grad.abs().max().item() + 1e-12
        np.testing.assert_allclose(a.cpu().numpy(), w.grad.float().cpu().numpy(), rtol=2e-5, atol=2e-6 * scale, err_msg=name_)
    # a row list
    flags = torch.zeros((2 * S,), dtype=torch.int32, device=DEV)
    flags[1::2] = 1
    rows = hip.compact_valid(flags)
    masked = (dl * flags.view(-1, 1), dv * flags.view(-1, 1))
    a = hip.mlp_backward(packed, weights, table, A, dl, dv, live=rows, fold=h)
    b = hip.mlp_backward(packed, weights, table, A, masked[0].contiguous(), masked[1].contiguous(), fold=h)
    for x_, y_ in zip(a, b):
        scale = y_.abs().max().item() + 1e-12
        np.testing.assert_allclose(x_.cpu().numpy(), y_.cpu().numpy(), rtol=2e-5, atol=2e-6 * scale)


def test_a_tree_with_ragged_legality_is_not_folded(tmp_path, monkeypatch):
    """tests/golden/tree_ragged.npz has per-state action counts: no fold, and RNaD's default step says so and runs the plain kernels."""
    import _gpu as G
    from environment.episode import Buffer
    from learn.rnad import RNaD

    tree, _ = G.golden_tree("ragged")
    assert not tree.handle().legal_foldable
    monkeypatch.setenv("RNAD_SAVE_DIR", str(tmp_path))
    rn = RNaD(tree=tree, device=G.DEV, directory_name="ragged", batch_size=1 << 12, eta=0.2, b1_adam=0.0, lr=1e-3,
              net_params={"type": "MLP", "max_actions": tree.max_actions, "width": 64})
    rn.initialize()
    rn.tabular_gate = 0
    assert rn._fold() is False
    buf = Buffer(1)
    for i in range(5):
        rn.train_step(buf, alpha=0.5)
        rn.total_steps += 1
    torch.cuda.synchronize()
    assert rn._packed_cache["maintained"] is False
    assert all(torch.isfinite(p).all() for p in rn.net.parameters())


@pytest.mark.parametrize("name", ("ternary4", "a5c4"))
def test_training_with_and_without_the_fold(name, tmp_path, monkeypatch):
    """Several default steps (graph replay included) with fold_legal on and off from the same weights and seeds: the same episodes at
    first, parameters equal up to the summation order of the first layer."""
    from environment.episode import Buffer
    from learn.rnad import RNaD
    from test_hip_bucket import TREES, _native_tree

    monkeypatch.setenv("RNAD_SAVE_DIR", str(tmp_path))
    tree = _native_tree(**TREES[name])
    out = {}
    for fold in (False, True):
        torch.manual_seed(13)
        rn = RNaD(tree=tree, device=DEV, directory_name=f"fold{int(fold)}", batch_size=1 << 13, eta=0.2, b1_adam=0.0, lr=1e-3,
                  net_params={"type": "MLP", "max_actions": tree.max_actions, "width": 64})
        rn.initialize()
        rn.tabular_gate, rn.fold_legal = 0, fold
        with torch.no_grad():
            for p in rn.net_reg_.parameters():
                p.mul_(1.01)
        buf = Buffer(1)
        grads = None
        for i in range(6):
            rn.train_step(buf, alpha=0.2 * i)
            rn.total_steps += 1
        torch.cuda.synchronize()
        assert rn._fold() is fold and rn._packed_cache["maintained"] is fold
        assert rn._graph["graph"] is not None and not rn._graph["failed"]
        out[fold] = [p.detach().clone() for n in (rn.net, rn.net_target) for p in n.parameters()]
    for a, b in zip(out[False], out[True]):
        scale = a.abs().max().item() + 1e-12
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=2e-4, atol=2e-6 * scale)
