"""Distinct observations (csrc/rows_dedup.hip, RNaD.dedup_rows): rows of the tree with the same observation get the same net outputs
(nn/net.py:37-51 sees nothing but the observation), so the table launch runs on one representative per group and the representatives'
records are copied to their groups -- bit for bit the tables of the launch on all rows --, and the groups' gradients are added up before
one backward over the representatives: the same weight gradient in another fp32 summation order."""
import os
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")


def _tree(depth=4, A=3, C=1, seed=0):
    from test_hip_bucket import _native_tree

    return _native_tree(A=A, C=C, depth=depth, seed=seed)


@pytest.mark.parametrize("half", (False, True))
def test_groups_are_rows_with_identical_observation_bits(half):
    tree = _tree()
    h = tree.handle()
    d = h.obs_dedup(half)
    table = h.observations_table(half).reshape(2 * h.S, -1)
    bits = table.view(torch.int16 if half else torch.int32)
    rep = d.rep_of.long()
    assert d.n_rows == 2 * h.S and d.n_unique < d.n_rows, "the deepest level's +-1 payoff matrices repeat"
    assert torch.equal(rep[rep], rep) and (rep <= torch.arange(d.n_rows, device=DEV)).all(), "a representative is the first row of its group"
    assert torch.equal(bits[rep], bits), "a row and its representative carry the same observation bits"
    uniq = d.uniq.rows.long()
    assert uniq.numel() == d.n_unique == int(d.uniq.count.item()) and torch.equal(uniq, torch.unique(rep))
    assert torch.unique(bits[uniq], dim=0).shape[0] == d.n_unique, "distinct representatives carry distinct observations"
    # the CSR of the groups with more than one row: rows ascending, representative first, every such row exactly once
    start, order = d.multi_start.long().cpu(), d.multi_order.long().cpu()
    size = torch.bincount(rep, minlength=d.n_rows).cpu()
    assert d.n_multi == int((size > 1).sum()) and start[-1] == order.numel() == int(size[size > 1].sum())
    rep_c = rep.cpu()
    for g in range(min(d.n_multi, 50)):
        rows = order[start[g]: start[g + 1]]
        assert (rows[1:] > rows[:-1]).all() and (rep_c[rows] == rows[0]).all() and size[rows[0]] == rows.numel()


@pytest.mark.parametrize("depth,fold", ((4, True), (4, False), (6, True)))
def test_tables_of_the_representatives_expanded_are_the_tables_of_all_rows(depth, fold):
    import rnad_hip as hip
    from test_hip_rows import _hp, _nets, _reference

    tree = _tree(depth)
    h = tree.handle()
    A, W = h.A, 256
    nets = _nets(A, W, 3)
    table = h.observations_table()
    hp = _hp(hip)
    packs, _, _, _, lr, lr2 = _reference(hip, h, nets, W, table, A, fold, hp)
    full = hip.mlp_rows_records(h, packs[0], packs[1], W, table, lr, lr2, hp, fold=h if fold else False)
    d = h.obs_dedup()
    part = hip.mlp_rows_records(h, packs[0], packs[1], W, table, lr, lr2, hp, fold=h if fold else False, rows=d.uniq, alloc_rows=2 * h.S)
    hip.rows_expand(d, [part["fast_records"], part["policy_rows"], part["records"]])
    for k in ("fast_records", "policy_rows", "records"):
        assert torch.equal(part[k].view(torch.int32), full[k].view(torch.int32)), k
    if depth == 6:
        assert d.n_unique < 20000 < d.n_rows == 132862


def test_segment_sum_adds_the_groups_up_reproducibly():
    import rnad_hip as hip

    tree = _tree()
    h = tree.handle()
    d = h.obs_dedup()
    A, N = h.A, d.n_rows
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    dl = torch.randn((N, A), device=DEV, generator=g)
    dv = torch.randn((N, 1), device=DEV, generator=g)
    rep = d.rep_of.long()
    want_l = torch.zeros((N, A), dtype=torch.float64, device=DEV).index_add_(0, rep, dl.double())
    want_v = torch.zeros((N, 1), dtype=torch.float64, device=DEV).index_add_(0, rep, dv.double())
    outs = []
    for _ in range(2):
        a, b = dl.clone(), dv.clone()
        hip.rows_segment_sum(d, A, a, b)
        outs.append((a, b))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "a fixed reduction tree: the same bits on every run"
    uniq = d.uniq.rows.long()
    np.testing.assert_allclose(outs[0][0][uniq].cpu().numpy(), want_l[uniq].cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(outs[0][1][uniq].cpu().numpy(), want_v[uniq].cpu().numpy(), rtol=1e-5, atol=1e-5)
    others = torch.ones(N, dtype=torch.bool, device=DEV)
    others[uniq] = False
    assert torch.equal(outs[0][0][others], dl[others]), "rows that are not representatives are left alone"


def _step(tree, dedup, use_graph=False, steps=1, want_grads=True, in_finish=True):
    from environment.episode import Buffer
    from learn.rnad import RNaD

    os.environ["RNAD_SAVE_DIR"] = tempfile.mkdtemp(prefix="rnad_dedup_")
    torch.manual_seed(7)
    rn = RNaD(tree=tree, device=DEV, directory_name="d", batch_size=1 << 14, eta=0.2, b1_adam=0.0, lr=1e-3,
              net_params={"type": "MLP", "max_actions": tree.max_actions, "width": 64})
    rn.initialize()
    rn.dedup_rows, rn.use_graph, rn.keep_last_tables, rn.group_sums_in_finish = dedup, use_graph, True, in_finish
    grads = []
    if want_grads and not use_graph:
        rn.fused_optimizer = False
        real = rn.optimizer.step
        rn.optimizer.step = lambda *a, **k: (grads.append([p.grad.detach().clone() for p in rn.net.parameters()]), real(*a, **k))[1]
    buf = Buffer(1)
    for i in range(steps):
        rn.train_step(buf, alpha=0.3)
        rn.total_steps += 1
    torch.cuda.synchronize()
    return rn, grads


def test_default_step_with_and_without_dedup():
    """Same episodes (the actor's policy rows are the same bits).  The groups added up by rnad_rows_segment_sum after a finish over all
    rows: the same per-row gradient tables bit for bit; added up inside k_bucket_finish: the same bits in the representatives' rows (the
    only rows the backward reads).  The weight gradients are equal up to the order of the fp32 sums."""
    tree = _tree()
    off, g_off = _step(tree, False)
    d = tree.handle().obs_dedup()
    uniq = d.uniq.rows.long()
    sums = [t.clone() for t in off.last_tables[:2]]
    import rnad_hip as hip

    hip.rows_segment_sum(d, tree.handle().A, *sums)
    for in_finish in (False, True):
        on, g_on = _step(tree, True, in_finish=in_finish)
        assert on.last_episodes is not None and torch.equal(on.last_episodes.indices, off.last_episodes.indices)
        if in_finish:
            assert d.groups_below_cut(tree.handle(), on.last_episodes.buckets.plan), "this tree and batch let finish add the groups up"
            assert torch.equal(on.last_tables[0][uniq], sums[0][uniq]) and torch.equal(on.last_tables[1][uniq], sums[1][uniq])
        else:
            assert torch.equal(on.last_tables[0], off.last_tables[0]) and torch.equal(on.last_tables[1], off.last_tables[1])
        for a, b in zip(g_on[0], g_off[0]):
            scale = b.abs().max().item() + 1e-12
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=2e-6 * scale)
        acc = on.last_episodes.buckets.plan.accumulators[: 2 * tree.handle().S * (tree.handle().A + 1)]
        assert not acc.any(), "finish cleared every accumulator it read: zero again for the next update"


def test_dedup_inside_the_captured_step_and_off_where_it_does_not_pay():
    tree = _tree()
    eager, _ = _step(tree, True, use_graph=False, steps=6, want_grads=False)
    graph, _ = _step(tree, True, use_graph=True, steps=6, want_grads=False)
    assert graph._graph["graph"] is not None and not graph._graph["failed"]
    for a, b in zip(eager.net.parameters(), graph.net.parameters()):
        assert torch.equal(a, b), "replayed steps are the eager ones"
    # a tree with chance nodes and Dirichlet weights: the expected values are all different -- nothing to share
    from test_hip_bucket import TREES, _native_tree

    big = _native_tree(**TREES["a5c4"])
    d = big.handle().obs_dedup()
    probe, _ = _step(big, True)
    assert probe._dedup_now(big.handle(), None, False, None, probe._fold()) is (d if 5 * d.n_unique <= 4 * d.n_rows else None)
