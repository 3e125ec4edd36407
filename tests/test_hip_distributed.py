"""The data-parallel update on REAL kernels: two ranks (sharing the one GPU of the test box, gloo transport) each roll out
half of the lanes and all-reduce normalisers + gradients; the result must equal the single-process full-batch step."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

B, SEED = 4096, 77


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _one_step(tmp, tag):
    from environment.episode import Buffer
    from environment.tree import Tree
    from learn.rnad import RNaD

    dev = torch.device("cuda:0")
    os.environ["RNAD_SAVE_DIR"] = os.path.join(tmp, tag)
    tree = Tree(device=dev, max_actions=3, max_transitions=2, depth_bound=3, transition_threshold=0.2)
    tree.generate_native(seed=4, prune=(1, 3))
    torch.manual_seed(SEED)  # same initial nets and the same rollout seed everywhere
    rn = RNaD(tree=tree, device=dev, directory_name="dp", batch_size=B, eta=0.2, b1_adam=0.0, lr=1e-3,
              net_params={"type": "MLP", "max_actions": 3, "width": 64})
    rn.initialize()
    log = {}
    rn.train_step(Buffer(1), alpha=0.4, log=log)
    torch.cuda.synchronize()
    return {k: v.detach().cpu().numpy() for k, v in rn.net.state_dict().items()}, log


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        params, log = _one_step(tmp, f"rank{rank}")
        np.savez(os.path.join(tmp, f"dp_{rank}.npz"), loss_v=log["loss_v"], loss_nerd=log["loss_nerd"], **params)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_equal_one_process(tmp_path):
    single, log = _one_step(str(tmp_path), "single")
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (np.load(tmp_path / f"dp_{r}.npz") for r in range(2))
    for k, want in single.items():
        np.testing.assert_array_equal(r0[k], r1[k])  # ranks stay in lock step
        np.testing.assert_allclose(r0[k], want, rtol=2e-5, atol=2e-7, err_msg=k)  # Adam normalises: compare loosely
    np.testing.assert_allclose(r0["loss_v"], log["loss_v"], rtol=1e-5)
    np.testing.assert_allclose(r0["loss_nerd"], log["loss_nerd"], rtol=1e-4, atol=1e-7)


def _nccl_single(rank, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                      NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="COLL", NCCL_DEBUG_FILE=os.path.join(tmp, "rccl_%p.log"))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    calls = {"all_reduce": 0, "broadcast": 0}
    real_ar, real_bc = dist.all_reduce, dist.broadcast

    def counting_all_reduce(t, *a, **kw):
        assert t.is_cuda  # the RCCL backend only takes device buffers
        calls["all_reduce"] += 1
        return real_ar(t, *a, **kw)

    def counting_broadcast(t, *a, **kw):
        assert t.is_cuda
        calls["broadcast"] += 1
        return real_bc(t, *a, **kw)

    dist.all_reduce, dist.broadcast = counting_all_reduce, counting_broadcast
    try:
        params, log = _one_step(tmp, "nccl1")
        np.savez(os.path.join(tmp, "nccl1.npz"), loss_v=log["loss_v"], loss_nerd=log["loss_nerd"], n_all_reduce=calls["all_reduce"],
                 n_broadcast=calls["broadcast"], **params)
    finally:
        dist.all_reduce, dist.broadcast = real_ar, real_bc
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_one_rank_rccl_group_runs_every_collective_of_the_update(tmp_path):
    """The box has one GPU, so RCCL cannot be given two ranks here; with a one-rank `nccl` group RNaD still takes its data-parallel
    branch (learn/rnad.py `_dist_on`), so every collective of the N-rank update goes through RCCL: the initial weight broadcasts,
    the int64 seed broadcast, the async f64 normaliser all-reduce, the fp32 gradient bucket and the logged losses.  Checked twice:
    by counting the calls, and in RCCL's own NCCL_DEBUG=INFO/COLL log."""
    single, log = _one_step(str(tmp_path), "single2")
    mp.spawn(_nccl_single, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    got = np.load(tmp_path / "nccl1.npz")
    for k, want in single.items():
        np.testing.assert_array_equal(got[k], want)  # a one-rank sum changes nothing
    assert abs(float(got["loss_v"]) - log["loss_v"]) < 1e-9  # the logged scalars are fp64 atomic sums: last bits vary
    assert int(got["n_all_reduce"]) >= 3, "normalisers, gradient bucket and logged losses must be all-reduced"
    assert int(got["n_broadcast"]) >= 9, "8 weight tensors + the rollout seed must be broadcast from rank 0"
    text = "".join(p.read_text(errors="replace") for p in tmp_path.glob("rccl_*.log"))
    n_ar = sum(1 for ln in text.splitlines() if "AllReduce" in ln and "opCount" in ln)
    n_bc = sum(1 for ln in text.splitlines() if "Broadcast" in ln and "opCount" in ln)
    assert n_ar >= 3 and n_bc >= 9, f"RCCL logged {n_ar} AllReduce / {n_bc} Broadcast operations:\n{text[-2000:]}"


def _nccl_graph(rank, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", RNAD_SAVE_DIR=os.path.join(tmp, "g"))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        from environment.episode import Buffer
        from environment.tree import Tree
        from learn.rnad import RNaD

        dev = torch.device("cuda:0")
        tree = Tree(device=dev, max_actions=3, max_transitions=1, depth_bound=4)
        tree.generate_native(seed=0)
        out = {"captured": True}
        # fuse: rollout and learner of the step in one launch (a data-parallel rank keeps two unless asked: RNaD._fuse_now) -- the counts
        # then come from k_bucket_alive_rep and the finish is the caller's, after the all-reduce of the normalisers
        for use_graph, fuse in ((False, False), (True, False), (False, True), (True, True)):
            torch.manual_seed(SEED)
            rn = RNaD(tree=tree, device=dev, directory_name=f"g{int(use_graph)}{int(fuse)}", batch_size=1 << 14, eta=0.2, b1_adam=0.0, lr=1e-3,
                      net_params={"type": "MLP", "max_actions": 3, "width": 64})
            rn.initialize()
            rn.use_graph = use_graph
            rn.fuse_rollout_learner = fuse
            buf = Buffer(1)
            for i in range(8):
                rn.train_step(buf, alpha=0.1 * i)
                rn.total_steps += 1
            torch.cuda.synchronize()
            assert (getattr(rn.last_episodes, "_compact", None) is not None), "the compact bucketed rollout must apply"
            out[(use_graph, fuse)] = [p.detach().cpu().numpy() for p in rn.net.parameters()]
            if use_graph:
                g = getattr(rn, "_graph", None)
                out["captured"] = out["captured"] and bool(g and g.get("graph") is not None and not g.get("failed"))
        np.savez(os.path.join(tmp, "nccl_graph.npz"), captured=out["captured"], **{f"e{i}": a for i, a in enumerate(out[(False, False)])},
                 **{f"g{i}": a for i, a in enumerate(out[(True, False)])}, **{f"ef{i}": a for i, a in enumerate(out[(False, True)])},
                 **{f"gf{i}": a for i, a in enumerate(out[(True, True)])})
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_graph_replay_with_rccl_collectives_inside(tmp_path):
    """Data-parallel steps are captured too: the normaliser and gradient all-reduces (RCCL) become nodes of the step's hipGraph.
    One rank here (one GPU on the box): the replayed steps must equal the eager ones bit for bit."""
    mp.spawn(_nccl_graph, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    got = np.load(tmp_path / "nccl_graph.npz")
    assert bool(got["captured"]), "the step with RCCL collectives inside must have been captured"
    for i in range(8):
        np.testing.assert_array_equal(got[f"e{i}"], got[f"g{i}"])
        # rollout + learner in one launch: the same per-row sums bit for bit, hence the same parameters
        np.testing.assert_array_equal(got[f"e{i}"], got[f"ef{i}"])
        np.testing.assert_array_equal(got[f"e{i}"], got[f"gf{i}"])


def _sharded_step(tmp, tag, shard):
    from environment.episode import Buffer
    from environment.tree import Tree
    from learn.rnad import RNaD

    dev = torch.device("cuda:0")
    os.environ["RNAD_SAVE_DIR"] = os.path.join(tmp, tag)
    tree = Tree(device=dev, max_actions=3, max_transitions=1, depth_bound=4)
    tree.generate_native(seed=0)
    torch.manual_seed(SEED)
    rn = RNaD(tree=tree, device=dev, directory_name="sh", batch_size=3 << 12,  # (lanes divisible by 2 and by 3 ranks)
              eta=0.2, b1_adam=0.0, lr=1e-3,
              net_params={"type": "MLP", "max_actions": 3, "width": 64})
    rn.initialize()
    rn.shard_rows = shard
    rn.keep_last_tables = True
    rn.group_sums_in_finish = False  # (the one-process step keeps its per-row tables: what the shards are compared with)
    rn.use_graph = False
    rn.train_step(Buffer(1), alpha=0.4)
    torch.cuda.synchronize()
    dlogit, dv, rows = rn.last_tables
    out = {k: v.detach().cpu().numpy() for k, v in rn.net.state_dict().items()}
    out["dlogit"], out["dv"] = dlogit.cpu().numpy(), dv.cpu().numpy()
    out["rows"] = rows.rows.cpu().numpy() if rows is not None else np.zeros(0, np.int32)
    return out


def _shard_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for shard in (True, False):
            np.savez(os.path.join(tmp, f"sh_{int(shard)}_{rank}.npz"), **_sharded_step(tmp, f"rank{rank}_{int(shard)}", shard))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", (2, 3))
def test_row_sharding_gives_the_one_process_row_sums(tmp_path, world):
    """RNaD.shard_rows: rank r evaluates the nets on its share of the 2S rows (all-gather of the record tables), the learner's 64-bit
    per-row sums are all-reduced, and the rank finishes and back-propagates its own rows.  The per-row gradient tables a rank ends up with
    are those of the ONE-process step on the whole batch bit for bit (integer sums; same rows, same records), the ranks' shares cover all
    rows (3 ranks: the shares are padded), and the update equals the replicated data-parallel one up to the order of the fp32 weight-gradient sums."""
    single = _sharded_step(str(tmp_path), "single", False)
    mp.spawn(_shard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    seen = np.zeros(single["dlogit"].shape[0], bool)
    for r in range(world):
        got = np.load(tmp_path / f"sh_1_{r}.npz")
        rows = got["rows"]
        assert rows.size > 0 and not seen[rows].any()
        seen[rows] = True
        n = single["dlogit"].shape[0]
        assert np.array_equal(got["dlogit"][:n][rows].view(np.uint32), single["dlogit"][rows].view(np.uint32)), f"dL/dlogit rows of rank {r}"
        assert np.array_equal(got["dv"][:n][rows].view(np.uint32), single["dv"][rows].view(np.uint32)), f"dL/dv rows of rank {r}"
        repl = np.load(tmp_path / f"sh_0_{r}.npz")
        for k in single:
            if k in ("dlogit", "dv", "rows"):
                continue
            np.testing.assert_array_equal(got[k], np.load(tmp_path / f"sh_1_0.npz")[k])  # the ranks stay in lock step
            np.testing.assert_allclose(got[k], repl[k], rtol=2e-5, atol=2e-7, err_msg=k)
            np.testing.assert_allclose(got[k], single[k], rtol=2e-5, atol=2e-7, err_msg=k)
    assert seen.all()


def _uniform_steps(tmp, tag, count=None):
    from environment.episode import Buffer
    from environment.tree import Tree
    from learn.rnad import RNaD

    dev = torch.device("cuda:0")
    os.environ["RNAD_SAVE_DIR"] = os.path.join(tmp, tag)
    tree = Tree(device=dev, max_actions=3, max_transitions=1, depth_bound=4)
    tree.generate_native(seed=0)
    assert tree.handle().uniform_length
    torch.manual_seed(SEED)
    rn = RNaD(tree=tree, device=dev, directory_name="un", batch_size=1 << 13, eta=0.2, b1_adam=0.0, lr=1e-3,
              net_params={"type": "MLP", "max_actions": 3, "width": 64})
    rn.initialize()
    rn.use_graph = False
    buf = Buffer(1)
    if count is not None:
        count["all_reduce"] = 0
    for i in range(3):
        rn.train_step(buf, alpha=0.2 * i)
        rn.total_steps += 1
    torch.cuda.synchronize()
    fused = getattr(rn.last_episodes, "_compact", None) is not None and rn._fuse_now()
    return {k: v.detach().cpu().numpy() for k, v in rn.net.state_dict().items()}, fused


def _uniform_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    count = {"all_reduce": 0}
    real = dist.all_reduce

    def counting(t, *a, **kw):
        count["all_reduce"] += 1
        return real(t, *a, **kw)

    dist.all_reduce = counting
    try:
        params, fused = _uniform_steps(tmp, f"u{rank}", count)
        np.savez(os.path.join(tmp, f"un_{rank}.npz"), n_all_reduce=count["all_reduce"], fused=fused, **params)
    finally:
        dist.all_reduce = real
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_on_a_uniform_tree_need_no_collective_for_the_normalisers(tmp_path):
    """Every episode of an unpruned tree lasts 2 * depth env steps, so the loss normalisers of the GLOBAL batch are batch_size * T / 2
    whatever the ranks played (RNaD._known_norm): the step all-reduces its 43 KB of gradients and nothing else, rollout and learner share
    one launch on every rank, and two ranks still train like one process."""
    single, _ = _uniform_steps(str(tmp_path), "single")
    mp.spawn(_uniform_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (np.load(tmp_path / f"un_{r}.npz") for r in range(2))
    assert int(r0["n_all_reduce"]) == 3 and int(r1["n_all_reduce"]) == 3, "one all-reduce per step: the gradient bucket"
    assert bool(r0["fused"]) and bool(r1["fused"])
    for k, want in single.items():
        np.testing.assert_array_equal(r0[k], r1[k])
        np.testing.assert_allclose(r0[k], want, rtol=2e-5, atol=2e-7, err_msg=k)  # (the ranks' gradient sums meet in another fp32 order)


# ------------------------------------------------------------------------------------------------ RCCL with more than one rank
def _rccl_ranks(rank, world, port, tmp, steps):
    """One process per GPU over `nccl` (= RCCL over xGMI): `steps` default steps on configs[1]'s shape two levels shallower, the first
    three enqueued eagerly, then captured with the collectives inside and replayed."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", RNAD_SAVE_DIR=os.path.join(tmp, f"r{rank}"),
                      NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,COLL", NCCL_DEBUG_FILE=os.path.join(tmp, "rccl2_%p.log"))
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from environment.episode import Buffer
        from environment.tree import Tree
        from learn.rnad import RNaD

        tree = Tree(device=dev, max_actions=3, max_transitions=1, depth_bound=4)
        tree.generate_native(seed=0)
        torch.manual_seed(SEED)
        rn = RNaD(tree=tree, device=dev, directory_name="rccl2", batch_size=1 << 14, eta=0.2, b1_adam=0.0, lr=1e-3,
                  net_params={"type": "MLP", "max_actions": 3, "width": 64})
        rn.initialize()
        rn.tabular_gate = 0
        buf = Buffer(1)
        for _ in range(steps):
            rn.train_step(buf, alpha=0.3)
            rn.total_steps += 1
        torch.cuda.synchronize()
        g = rn.__dict__.get("_graph") or {}
        np.savez(os.path.join(tmp, f"rccl2_{rank}.npz"), replayed=int(g.get("graph") is not None), world=dist.get_world_size(),
                 **{k: v.detach().cpu().numpy() for k, v in rn.net.state_dict().items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL with more than one rank needs at least two GPUs (the build loop's box has one)")
def test_two_rccl_ranks_train_like_one_process(tmp_path):
    """The first multi-GPU run's gate (BASELINE.json configs[2], [4]): 2 ranks over RCCL, eager steps then the captured step with the
    collectives inside, against the one-process step on the same global batch -- the bounds of the gloo test above (Adam normalises)."""
    from environment.episode import Buffer
    from environment.tree import Tree
    from learn.rnad import RNaD

    steps = 8
    dev = torch.device("cuda:0")
    os.environ["RNAD_SAVE_DIR"] = str(tmp_path / "single")
    tree = Tree(device=dev, max_actions=3, max_transitions=1, depth_bound=4)
    tree.generate_native(seed=0)
    torch.manual_seed(SEED)
    rn = RNaD(tree=tree, device=dev, directory_name="rccl2", batch_size=1 << 14, eta=0.2, b1_adam=0.0, lr=1e-3,
              net_params={"type": "MLP", "max_actions": 3, "width": 64})
    rn.initialize()
    rn.tabular_gate = 0
    buf = Buffer(1)
    for _ in range(steps):
        rn.train_step(buf, alpha=0.3)
        rn.total_steps += 1
    torch.cuda.synchronize()
    single = {k: v.detach().cpu().numpy() for k, v in rn.net.state_dict().items()}
    mp.spawn(_rccl_ranks, args=(2, _free_port(), str(tmp_path), steps), nprocs=2, join=True)
    r0, r1 = (np.load(tmp_path / f"rccl2_{r}.npz") for r in range(2))
    assert int(r0["world"]) == 2 and int(r0["replayed"]) == 1 and int(r1["replayed"]) == 1
    for k, want in single.items():
        np.testing.assert_array_equal(r0[k], r1[k])  # ranks stay in lock step
        np.testing.assert_allclose(r0[k], want, rtol=2e-4, atol=2e-6, err_msg=k)
    text = "".join(p.read_text(errors="replace") for p in tmp_path.glob("rccl2_*.log"))
    assert "nranks 2" in text or "nRanks 2" in text, "RCCL's own log must show a 2-rank communicator"
    assert sum(1 for ln in text.splitlines() if "AllReduce" in ln and "opCount" in ln) >= steps, "one gradient all-reduce per step and rank"
