"""Trainer-level guards (r05):
  * a replay buffer of several batches (reference learn/rnad.py:66-67,502-507: n_batches_per_buffer > 1) must not run on record tables
    that were written for the representatives of the distinct observations only;
  * two RNaD objects over one tree and batch size (reference main.py:55-81 builds several trainers over one tree) stepping on two streams
    of one device: each has its own workspaces (rnad_hip.workspace_owner) and its own optimiser ticket -- the same parameters, bit for
    bit, as each trainer stepping alone.
"""
import os
import random
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")


def _tree(depth=4):
    from test_hip_bucket import _native_tree

    return _native_tree(A=3, C=1, depth=depth, seed=0)


def _rnad(tree, B=1 << 13, width=64, seed=7, n_batches=1, name="t"):
    from learn.rnad import RNaD

    os.environ["RNAD_SAVE_DIR"] = tempfile.mkdtemp(prefix="rnad_trainers_")
    torch.manual_seed(seed)
    rn = RNaD(tree=tree, device=DEV, directory_name=name, batch_size=B, eta=0.2, b1_adam=0.0, lr=1e-3, n_batches_per_buffer=n_batches,
              net_params={"type": "MLP", "max_actions": tree.max_actions, "width": width})
    rn.initialize()
    rn.tabular_gate = 0
    rn._seed_base, rn._seed_count = 1000 + seed, 0  # (not from torch's global generator: another trainer built in between would move it)
    torch.cuda.synchronize()  # the nets exist before anything steps on another stream
    return rn


def _run_buffered(tree, steps, dedup, mode=True):
    from environment.episode import Buffer

    rn = _rnad(tree, n_batches=2)
    rn.dedup_rows, rn.tabular, rn.use_graph = dedup, mode, False
    np.random.seed(3)
    random.seed(3)
    buf = Buffer(rn.n_batches_per_buffer)
    used = []
    for _ in range(steps):
        rn.train_step(buf, alpha=0.4)
        rn.total_steps += 1
        used.append(rn._dedup_now(tree.handle(), None, False, False, rn._fold(), rn._plays_what_it_learns(tree.handle(), rn.batch_size, 8, buf)))
    torch.cuda.synchronize()
    return rn, used


def test_a_buffer_of_several_batches_never_learns_from_partial_tables():
    """with n_batches_per_buffer = 2 the second step learns from a collated sample that is not bucket-ordered, through
    rnad_learn_fused_tabular, which gathers the logit / v / v_target TABLES -- valid in every row only when the nets ran on every row."""
    tree = _tree()
    d = tree.handle().obs_dedup()
    assert 5 * d.n_unique <= 4 * d.n_rows, "this tree has enough repeated observations for the distinct-observation launch"
    on, used = _run_buffered(tree, 4, True)
    assert all(u is None for u in used), "distinct observations must be off while the buffer holds several batches"
    off, _ = _run_buffered(tree, 4, False)
    for (k, a), b in zip(on.net.named_parameters(), off.net.parameters()):
        assert torch.isfinite(a).all(), k
        assert torch.equal(a, b), f"{k}: dedup_rows on / off must take the same path here"
    # the step's gradients on a collated sample of the two buffered batches: the per-row mode (what the steps above ran) against the
    # per-slot backward ("forward": bit-identical to the dense, reference-shaped program) on the SAME sample
    from environment.episode import Buffer

    buf = Buffer(2)
    for _ in range(2):
        on.train_step(buf, alpha=0.4)
        on.total_steps += 1
    np.random.seed(5)
    random.seed(5)
    batch = buf.sample(on.batch_size)
    assert getattr(batch, "buckets", None) is None and batch.batch_size == on.batch_size, "a collated sample is not bucket-ordered"
    grads = {}
    for mode in (True, "forward"):
        on.tabular = mode
        on.optimizer.zero_grad(set_to_none=True)
        on._RNaD__learn(batch, 0.4)
        grads[mode] = [p.grad.detach().clone() for p in on.net.parameters()]
    for (k, _), a, b in zip(on.net.named_parameters(), grads[True], grads["forward"]):
        scale = b.abs().max().item() + 1e-12
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=2e-6 * scale, err_msg=k)


def test_partial_tables_are_refused_by_the_table_learner():
    """The guard itself: __learn on a batch that is not bucket-ordered with tables that carry `dedup` raises instead of gathering
    uninitialised rows."""
    from environment.episode import Buffer, Episodes

    tree = _tree()
    rn = _rnad(tree)
    rn.use_graph = False
    buf = Buffer(1)
    rn.train_step(buf, alpha=0.4)
    h = tree.handle()
    tables = rn._table_outputs(0.4, fold=rn._fold(), records_hp=rn._learn_params(0.4), dedup=h.obs_dedup())
    ep = Episodes(tree, rn.batch_size, seed=5)
    ep.generate(rn.net, tabular=True, trim=False)  # lane-ordered: no buckets
    with pytest.raises(RuntimeError, match="representatives"):
        rn._RNaD__learn(ep, 0.4, tables=tables)


def _steps(rn, buf, n, stream=None):
    for _ in range(n):
        if stream is None:
            rn.train_step(buf, alpha=0.3)
        else:
            with torch.cuda.stream(stream):
                rn.train_step(buf, alpha=0.3)
        rn.total_steps += 1


# fresh: the two-stream trainers are the FIRST users of a tree handle of their own (r05 advisor: the solo runs used to warm every cache
# kept on the shared handle -- observation table, distinct observations -- before the two streams met them)
@pytest.mark.parametrize("graph,fresh", ((False, False), (True, False), (False, True)))
def test_two_trainers_on_two_streams_of_one_device(graph, fresh):
    from environment.episode import Buffer

    tree = _tree()
    n = 12
    solo = []
    for seed in (7, 8):
        rn = _rnad(_tree() if fresh else tree, seed=seed, name=f"solo{seed}")
        rn.use_graph = graph
        _steps(rn, Buffer(1), n)
        torch.cuda.synchronize()
        solo.append([p.detach().clone() for p in rn.net.parameters()])
    a, b = _rnad(tree, seed=7, name="a"), _rnad(tree, seed=8, name="b")
    a.use_graph = b.use_graph = graph
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ba, bb = Buffer(1), Buffer(1)
    for _ in range(n):  # alternately, each on its own stream: their kernels overlap on the device
        _steps(a, ba, 1, sa)
        _steps(b, bb, 1, sb)
    torch.cuda.synchronize()
    pa, pb = a.last_episodes.buckets.plan, b.last_episodes.buckets.plan
    assert pa is not pb and pa.accumulators.data_ptr() != pb.accumulators.data_ptr() and pa.scratch.data_ptr() != pb.scratch.data_ptr()
    assert a._fused_tail().ticket.data_ptr() != b._fused_tail().ticket.data_ptr()
    for got, want in ((a, solo[0]), (b, solo[1])):
        for (k, p), w in zip(got.net.named_parameters(), want):
            assert torch.equal(p, w), f"{k}: a trainer stepping beside another one must train like it does alone"


def test_workspaces_go_with_their_trainer():
    import gc

    import rnad_hip

    tree = _tree()
    rn = _rnad(tree)
    from environment.episode import Buffer

    rn.use_graph = False
    rn.train_step(Buffer(1), alpha=0.3)
    torch.cuda.synchronize()
    owners = tree.handle().__dict__["_owner_plans"]
    assert len(owners) == 1
    with rnad_hip.workspace_owner(rn._workspace_token()):
        assert rnad_hip.bucket_plan(tree.handle(), rn.batch_size) is rn.last_episodes.buckets.plan
    assert rnad_hip.bucket_plan(tree.handle(), rn.batch_size) is not rn.last_episodes.buckets.plan, "outside: the shared default set"
    del rn
    gc.collect()
    assert len(owners) == 0


def test_a_sort_tile_the_scratch_was_not_sized_for_is_refused(monkeypatch):
    """r05 advisor: csrc/bucket.hip reads RNAD_SORT_TILE on every call while the scratch buffer was sized by rnad_bucket_plan once -- a knob
    that changed in between laid a larger histogram over the buffer.  Now the plan caches (C and Python) are keyed by the knobs, and the
    entry points that write the scratch refuse a combination nobody asked the sizes of."""
    import rnad_hip as hip
    from environment.episode import Episodes
    from nn.net import MLP

    tree = _tree()
    h = tree.handle()
    B, T = 1 << 13, 2 * h.max_depth
    monkeypatch.delenv("RNAD_SORT_TILE", raising=False)
    plan = hip.bucket_plan(h, B)
    torch.manual_seed(0)
    net = MLP(3, 64, device=DEV)
    ep = Episodes(tree, B, seed=5)
    ep.generate(net, bucketed=True)
    want = ep.indices[:, torch.argsort(ep.lane_ids.long())].clone()
    other = 2048 if plan.sort_tile != 2048 else 4096
    monkeypatch.setenv("RNAD_SORT_TILE", str(other))
    # the C entry point with the OLD plan's scratch: refused, loudly
    alive = torch.zeros((T + 1,), dtype=torch.int32, device=DEV)
    norm = torch.zeros((2,), dtype=torch.float64, device=DEV)
    with pytest.raises(hip.RnadHipError, match="rnad_bucket_plan"):
        hip._check(hip.lib().rnad_bucket_alive(h.ptr, T, B, hip._dp(plan.scratch, torch.int32, "scratch"), hip._dp(alive, torch.int32, "alive"),
                                               hip._dp(norm, torch.float64, "norm"), hip._stream()))
    # through the binding: a plan of its own for the new knob, sized for it -- and the same episodes
    again = hip.bucket_plan(h, B)
    assert again is not plan and again.sort_tile == other and plan.sort_tile != other
    ep2 = Episodes(tree, B, seed=5)
    ep2.generate(net, bucketed=True)
    assert torch.equal(ep2.indices[:, torch.argsort(ep2.lane_ids.long())], want)
