"""RNaD.train_step replayed from a captured hipGraph is the eager step, bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


_BUFFERS = {}  # the replay buffer of each _run (the captured graph is bound to it)


def _run(tree, tmp_path, use_graph, steps, rotate_at=None, tag="", schedule=None, log_at=()):
    import os

    from environment.episode import Buffer
    from learn.rnad import RNaD

    os.environ["RNAD_SAVE_DIR"] = str(tmp_path)
    torch.manual_seed(7)
    rn = RNaD(tree=tree, device=DEV, directory_name=f"g{int(use_graph)}{tag}", batch_size=1 << 14, eta=0.2, b1_adam=0.0, lr=1e-3,
              net_params={"type": "MLP", "max_actions": tree.max_actions, "width": 64})
    rn.initialize()
    rn.use_graph = use_graph
    with torch.no_grad():
        for p in rn.net_reg_.parameters():
            p.mul_(1.01)
    buf = Buffer(1)
    _BUFFERS[id(rn)] = buf
    seeds = []
    for i in range(steps):
        if i == rotate_at:  # what __resume does between two regularisation updates (rnad.py:528-531)
            rn.net_reg_.load_state_dict(rn.net_reg.state_dict())
            rn.net_reg.load_state_dict(rn.net_target.state_dict())
        if schedule is not None:  # the caller announces its alphas, as RNaD.run does
            rn.alpha_ahead = lambda k, i=i: schedule(i + k)
        rn.train_step(buf, alpha=min(1.0, 0.15 * i) if schedule is None else schedule(i), log={} if i in log_at else None)
        rn.total_steps += 1
        seeds.append(rn.last_episodes.seed)
    torch.cuda.synchronize()
    nets = [p.detach().clone() for n in (rn.net, rn.net_target) for p in n.parameters()]
    return rn, nets, seeds


@pytest.mark.parametrize("name", ("ternary4", "pruned"))
def test_graph_replay_equals_eager_steps(name, tmp_path):
    from test_hip_bucket import TREES, _native_tree

    tree = _native_tree(**TREES[name])
    eager, nets_e, seeds_e = _run(tree, tmp_path, False, 10, rotate_at=6)
    graph, nets_g, seeds_g = _run(tree, tmp_path, True, 10, rotate_at=6)
    assert getattr(eager, "_graph", None) is None
    assert graph._graph["graph"] is not None and not graph._graph["failed"], "the step must have been captured"
    assert seeds_e == seeds_g
    for a, b in zip(nets_e, nets_g):
        assert torch.equal(a, b)
    # the trajectory of the last replay is a real one: same episodes as an eager rollout with that seed
    ep_e, ep_g = eager.last_episodes, graph.last_episodes
    assert torch.equal(ep_e.indices, ep_g.indices) and torch.equal(ep_e.policy, ep_g.policy) and torch.equal(ep_e.lane_ids, ep_g.lane_ids)
    assert np.isfinite(sum(float(p.abs().sum()) for p in nets_g))


def test_replayed_steps_take_their_scalars_from_the_queue(tmp_path):
    """The optimiser launch of a captured step moves a queue of (seed, alpha) on, so that a replay needs no launch before it -- as long as
    the step's scalars are the queued ones.  Announced schedule: one queue per RNAD_STEP_QUEUE steps; a logging (eager) step in between
    takes a seed of its own and the queue is set again; an alpha that was not announced: set again, every step.  Always the eager steps."""
    import rnad_hip as hip
    from test_hip_bucket import TREES, _native_tree

    tree = _native_tree(**TREES["ternary4"])
    steps = hip.STEP_QUEUE + 12
    schedule = lambda i: 1 if i > 15 else i * 2 / 30  # noqa: E731  (rnad.py:497 with delta_m = 30)
    eager, nets_e, seeds_e = _run(tree, tmp_path, False, steps, tag="q", schedule=schedule, log_at=(20,))
    graph, nets_g, seeds_g = _run(tree, tmp_path, True, steps, tag="q", schedule=schedule, log_at=(20,))
    g = graph._graph
    assert g["graph"] is not None and not g["failed"] and g["advances"]
    assert seeds_e == seeds_g
    for a, b in zip(nets_e, nets_g):
        assert torch.equal(a, b)
    replays = steps - graph._GRAPH_WARMUP - 1  # (the logging step ran eagerly)
    assert g["queue_sets"] == 2 + (replays - (20 - graph._GRAPH_WARMUP)) // hip.STEP_QUEUE, "at the capture, after the logging step, when it ran out"
    # no announcement and a moving alpha: every replay sets its own scalars -- still the eager steps (test_graph_replay_equals_eager_steps)
    moving, _, _ = _run(tree, tmp_path, True, 8, tag="m")
    assert moving._graph["queue_sets"] == 8 - moving._GRAPH_WARMUP


def test_many_replays_stay_finite(tmp_path):
    """150 replays of one captured step on the small golden tree (one group, no upper states, B = 512).  Regression: the loss sums
    and the overflow flag used to be cleared by hipMemsetAsync, and the memset node of the captured graph wrote garbage into them
    after ~57 replays (ROCm 7.2) -- a set overflow flag poisons the gradient tables with NaN.  They are cleared by a kernel now."""
    import os

    from _gpu import golden_tree
    from environment.episode import Buffer
    from learn.rnad import RNaD

    tree, _ = golden_tree("small")
    os.environ["RNAD_SAVE_DIR"] = str(tmp_path)
    torch.manual_seed(2000)
    rn = RNaD(tree=tree, device=DEV, directory_name="many", batch_size=512, eta=0.2, b1_adam=0.0, lr=5e-3,
              net_params={"type": "MLP", "max_actions": 3, "width": 256})
    rn.initialize()
    buf = Buffer(1)
    for i in range(150):
        rn.train_step(buf, alpha=min(1.0, i / 50))
        rn.total_steps += 1
    torch.cuda.synchronize()
    assert rn._graph["graph"] is not None and not rn._graph["failed"]
    assert all(torch.isfinite(p).all() for n in (rn.net, rn.net_target) for p in n.parameters())


def test_derived_episode_fields_follow_the_replayed_batch(tmp_path):
    """The captured step rewrites last_episodes' buffers in place: fields that are built on access (dense fields of the compact
    trajectory, observations, one-hot actions) must be rebuilt from the new batch after every replay."""
    from test_hip_bucket import TREES, _native_tree

    tree = _native_tree(**TREES["pruned"])
    rn, _, _ = _run(tree, tmp_path, True, 6, tag="inv")
    assert rn._graph["graph"] is not None
    ep = rn.last_episodes
    before = {k: getattr(ep, k).clone() for k in ("indices", "policy", "action_idx", "rewards", "observations", "actions", "masks")}
    # one more replay with another seed: a different batch in the same buffers
    rn.train_step(_BUFFERS[id(rn)], alpha=0.9)
    torch.cuda.synchronize()
    assert rn.last_episodes is ep
    assert not torch.equal(ep.indices, before["indices"]), "another seed must give another batch"
    A, S = tree.max_actions, tree.handle().S
    T = ep.t_eff + 1
    live = ep.indices != 0
    # the acting policy of every live slot is the row of the records the step used; rebuilt fields agree with the new indices
    rows = ep.indices.long() + (torch.arange(T, device=DEV) % 2).view(T, 1) * S
    rec = ep._compact[1]
    col = 3 * A + 3
    assert torch.equal(ep.policy[live], rec[rows[live]][:, col:col + A])
    assert torch.equal(ep.actions.argmax(-1)[live], ep.action_idx.long()[live])
    obs_now = ep.observations
    fresh = torch.empty_like(obs_now)
    import rnad_hip

    for t in range(T):
        rnad_hip.observe(tree.handle(), ep.indices[t], t & 1, obs=fresh[t])
    assert torch.equal(obs_now, fresh)


def test_logging_steps_and_mode_changes_leave_the_graph(tmp_path):
    """A logging step runs eagerly between replays; changing a baked-in hyper-parameter re-captures."""
    from environment.episode import Buffer
    from test_hip_bucket import TREES, _native_tree

    tree = _native_tree(**TREES["ternary4"])
    rn, _, _ = _run(tree, tmp_path, True, 5, tag="log")
    first = rn._graph["graph"]
    assert first is not None
    buf = rn.last_episodes and Buffer(1)
    log = {}
    rn.train_step(buf, alpha=0.5, log=log)  # eager (and a new buffer: a new capture will be needed)
    assert np.isfinite(log["loss_v"]) and np.isfinite(log["loss_nerd"])
    rn.eta = 0.3
    for _ in range(5):
        rn.train_step(buf, alpha=0.5)
    assert rn._graph["graph"] is not None and rn._graph["graph"] is not first
    torch.cuda.synchronize()
    assert all(torch.isfinite(p).all() for p in rn.net.parameters())


def test_fused_optimizer_tail_is_clip_adam_ema(tmp_path):
    """rnad_optimizer_step (clip + Adam + EMA target in one launch, on torch.optim.Adam's own state) against the torch sequence it
    replaces (rnad_clip_grad_norm / clip_grad_norm_, Adam.step, _foreach EMA), step after step from the same start."""
    import os

    from environment.episode import Buffer
    from learn.rnad import RNaD
    from test_hip_bucket import TREES, _native_tree

    tree = _native_tree(**TREES["ternary4"])
    os.environ["RNAD_SAVE_DIR"] = str(tmp_path)
    runs = {}
    for fused in (False, True):
        torch.manual_seed(3)
        rn = RNaD(tree=tree, device=DEV, directory_name=f"opt{int(fused)}", batch_size=1 << 13, eta=0.2, b1_adam=0.0, lr=1e-3, grad_clip=0.05,
                  gamma_averaging=0.01, net_params={"type": "MLP", "max_actions": 3, "width": 64})
        rn.initialize()
        rn.use_graph, rn.fused_optimizer = False, fused
        buf = Buffer(1)
        for i in range(6):
            rn.train_step(buf, alpha=0.2 * i)
            rn.total_steps += 1
        assert (rn._fused_tail() is not None) == fused
        st = rn.optimizer.state_dict()["state"]
        runs[fused] = ([p.detach().clone() for n in (rn.net, rn.net_target) for p in n.parameters()],
                       [st[k]["exp_avg_sq"].clone() for k in sorted(st)], [float(st[k]["step"]) for k in sorted(st)])
    assert runs[True][2] == runs[False][2] == [6.0] * 8
    for a, b in zip(runs[False][0] + runs[False][1], runs[True][0] + runs[True][1]):
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=2e-5, atol=1e-8)


def test_invalidate_tables_refreshes_the_captured_buffers_in_place(tmp_path):
    """invalidate_tables() after an edit autograd's version counters do not see -- of a regularisation net AND of the learner net
    (`.data` writes, as _sync_from_rank0's broadcast does): the regularisation tables and the packed weight images a captured step reads
    must be refreshed at the addresses the graph has baked in -- replays then equal eager steps from the same edit."""
    from test_hip_bucket import TREES, _native_tree

    tree = _native_tree(**TREES["ternary4"])
    out = {}
    for use_graph in (False, True):
        rn, _, _ = _run(tree, tmp_path, use_graph, 5, tag="inval")
        buf = _BUFFERS[id(rn)]
        ptrs = [t.data_ptr() for t in rn._reg_tables(tree.handle().observations_table())]
        images = [t.data_ptr() for t in rn._packed_images()]
        for p in rn.net_reg.parameters():
            p.data = p.data * 1.05  # a new tensor behind the same Parameter: no version bump on the old storage
        before = [p.detach().clone() for p in rn.net.parameters()]
        with torch.no_grad():
            for p in list(rn.net.parameters()) + list(rn.net_target.parameters()):
                p.data.mul_(0.97)  # in place through .data: same storage, same version counter -- invisible to _packed_images' key
        key = rn._packed_cache["layouts"][bool(rn._packed_cache["maintained"])]["key"]
        assert key == tuple((id(w), w.data_ptr(), w._version) for group in (rn.net._weights(), rn.net_target._weights()) for w in group)
        rn.invalidate_tables()
        assert [t.data_ptr() for t in rn._packed_images()] == images, "the weight images must be re-packed in place"
        assert all(not torch.equal(a, b) for a, b in zip(before, rn.net.parameters()))
        for i in range(3):
            rn.train_step(buf, alpha=0.5)
            rn.total_steps += 1
        torch.cuda.synchronize()
        assert [t.data_ptr() for t in rn._reg_tables(tree.handle().observations_table())] == ptrs, "the tables must be refreshed in place"
        if use_graph:
            assert rn._graph["graph"] is not None and not rn._graph["failed"]
        out[use_graph] = [p.detach().clone() for n in (rn.net, rn.net_target) for p in n.parameters()]
    for a, b in zip(out[False], out[True]):
        assert torch.equal(a, b)


def test_last_episodes_is_the_replayed_batch_after_an_eager_logging_step(tmp_path):
    """A logging step between two replays leaves its own Episodes in last_episodes / the buffer; the next replay must put the
    captured batch (the one it rewrites) back, labelled with that replay's seed."""
    from test_hip_bucket import TREES, _native_tree

    tree = _native_tree(**TREES["ternary4"])
    rn, _, _ = _run(tree, tmp_path, True, 6, tag="lastep")
    buf = _BUFFERS[id(rn)]
    captured = rn.last_episodes
    assert rn._graph["graph"] is not None and rn._graph["episodes"] is captured
    rn.train_step(buf, alpha=0.5, log={})
    assert rn.last_episodes is not captured
    rn.train_step(buf, alpha=0.5)
    torch.cuda.synchronize()
    assert rn.last_episodes is captured and buf.episodes_buffer[-1] is captured
    # the label matches the content: an eager rollout with that seed and these records' actor plays the same batch
    from environment.episode import Episodes

    again = Episodes(tree, captured.batch_size, seed=captured.seed)
    rec = captured._compact[1]
    import rnad_hip

    again.generate(rn.net, trim=False, store_values=False, tabular=True, bucketed=True, policy_table=(rec, rnad_hip.policy_column(tree.max_actions)),
                   compact=True)
    assert torch.equal(again.indices, captured.indices) and torch.equal(again.lane_ids, captured.lane_ids)


def test_changing_the_learning_rate_recaptures_the_step(tmp_path):
    """lr / betas / eps travel by value in the captured optimiser launch: editing param_groups must not be silently ignored."""
    from test_hip_bucket import TREES, _native_tree

    tree = _native_tree(**TREES["ternary4"])
    rn, _, _ = _run(tree, tmp_path, True, 5, tag="lr")
    buf = _BUFFERS[id(rn)]
    first = rn._graph["graph"]
    assert first is not None
    before = [p.detach().clone() for p in rn.net.parameters()]
    rn.optimizer.param_groups[0]["lr"] = 0.0
    for _ in range(5):
        rn.train_step(buf, alpha=0.5)
    torch.cuda.synchronize()
    assert rn._graph["graph"] is not None and rn._graph["graph"] is not first
    for a, b in zip(before, rn.net.parameters()):
        assert torch.equal(a, b), "with lr = 0 Adam must leave the parameters where they were"


def test_resume_from_a_reference_style_checkpoint_keeps_the_graph_and_the_fused_tail(tmp_path, monkeypatch):
    """optimizer.load_state_dict brings the checkpoint's param_groups -- the reference writes capturable=False, fused=None and CPU
    step counters (rnad.py:232-237, :318; tests/golden/ref_run_ckpt_1_0 is one, at a width the fused MLP does not take) -- after
    _resume_from the captured step and the one-launch optimiser tail must still be in use."""
    from environment.episode import Buffer
    from learn.rnad import RNaD
    from test_hip_bucket import TREES, _native_tree

    monkeypatch.setenv("RNAD_SAVE_DIR", str(tmp_path))
    tree = _native_tree(**TREES["ternary4"])
    kw = dict(tree=tree, device=DEV, directory_name="refstyle", batch_size=1 << 13, eta=0.2, b1_adam=0.0, lr=1e-3, bounds=[4], delta_m=[3],
              net_params={"type": "MLP", "max_actions": 3, "width": 64})
    torch.manual_seed(4)
    first = RNaD(**kw)
    first.run(max_updates=1, checkpoint_mod=1, expl_mod=10**9, log_mod=10**9)  # m = 0: checkpoints 0/0, 0/1, 0/2
    ck_path = tmp_path / "saved_runs" / "refstyle" / "0" / "2"
    ck = torch.load(ck_path, weights_only=False)
    for grp in ck["optimizer"]["param_groups"]:  # what torch 2.0's Adam(...) of the reference saves
        grp.update(capturable=False, fused=None, foreach=None)
    for st in ck["optimizer"]["state"].values():
        st["step"] = st["step"].detach().cpu()
    torch.save(ck, ck_path)
    rn = RNaD(**kw)
    rn.initialize()
    assert (rn.m, rn.n) == (0, 2)
    grp = rn.optimizer.param_groups[0]
    assert grp["capturable"] and grp["fused"]
    assert all(st["step"].is_cuda and st["step"].dtype == torch.float32 and float(st["step"]) == 2.0 for st in rn.optimizer.state.values())
    buf = Buffer(1)
    for i in range(6):
        rn.train_step(buf, alpha=0.5)
        rn.total_steps += 1
    torch.cuda.synchronize()
    assert rn._fused_tail() is not None
    assert rn._graph["graph"] is not None and not rn._graph["failed"]
    assert all(float(st["step"]) == 8.0 for st in rn.optimizer.state.values())
    assert all(torch.isfinite(p).all() for p in rn.net.parameters())


@pytest.mark.parametrize("use_graph", (False, True))
def test_packed_weight_images_follow_the_optimiser(use_graph, tmp_path):
    """No pack launch per step: rnad_optimizer_step writes every new weight (and EMA target weight) into the packed images the MLP
    kernels read.  After a run of steps the two images must be exactly what rnad_mlp_pack makes of the tensors -- also after an
    edit of the weights behind the trainer's back (picked up through the version counters before the next step / replay)."""
    import rnad_hip
    from test_hip_bucket import TREES, _native_tree

    tree = _native_tree(**TREES["ternary4"])
    rn, _, _ = _run(tree, tmp_path, use_graph, 7, tag="img")
    buf = _BUFFERS[id(rn)]
    A = tree.max_actions

    fold = rn._packed_cache["maintained"]
    assert fold is True, "this tree's legal planes are uniform: the FOLD layout must be the one in use"

    def check():
        torch.cuda.synchronize()
        fresh = rnad_hip.mlp_pack_many([rn.net._weights(), rn.net_target._weights()], A, fold=fold)
        held = rn._packed_cache["layouts"][fold]["images"]
        assert torch.equal(held[0], fresh[0]) and torch.equal(held[1], fresh[1])

    check()
    ptrs = [t.data_ptr() for t in rn._packed_cache["layouts"][fold]["images"]]
    with torch.no_grad():
        for p in rn.net.parameters():
            p.mul_(0.5)  # bumps the version counters
    for i in range(3):
        rn.train_step(buf, alpha=0.5)
        rn.total_steps += 1
    check()
    assert [t.data_ptr() for t in rn._packed_cache["layouts"][fold]["images"]] == ptrs, "the images are re-packed in place (a captured step reads them)"
    if use_graph:
        assert rn._graph["graph"] is not None and not rn._graph["failed"]


def test_the_learner_on_distinct_trajectories_trains_like_the_per_lane_learner(tmp_path, monkeypatch):
    """RNaD.distinct_trajectories (rollout + learner in one launch, the learner half once per distinct trajectory of a work item): the
    per-row sums are integer sums of the same addends, so the trainer ends up with the same parameters bit for bit -- forced on from the
    first step, and switched on by itself after DISTINCT_AFTER updates (a new graph is captured at the switch)."""
    from learn.rnad import RNaD
    from test_hip_bucket import TREES, _native_tree

    tree = _native_tree(**TREES["ternary4"])
    monkeypatch.setattr(RNaD, "distinct_trajectories", False, raising=False)
    off, nets_off, seeds_off = _run(tree, tmp_path, True, 12, tag="d0")
    assert off._fuse_now() and not off._distinct_now()
    assert getattr(off.last_episodes, "_compact", None) is not None and off._graph["graph"] is not None
    monkeypatch.setattr(RNaD, "distinct_trajectories", True, raising=False)
    on, nets_on, seeds_on = _run(tree, tmp_path, True, 12, tag="d1")
    assert on._distinct_now() and on._graph["graph"] is not None and not on._graph["failed"]
    monkeypatch.setattr(RNaD, "distinct_trajectories", None, raising=False)
    monkeypatch.setattr(RNaD, "DISTINCT_AFTER", 6)
    auto, nets_auto, seeds_auto = _run(tree, tmp_path, True, 12, tag="d2")
    assert auto._distinct_now() and auto._graph["graph"] is not None and not auto._graph["failed"]
    assert seeds_off == seeds_on == seeds_auto
    for a, b, c in zip(nets_off, nets_on, nets_auto):
        assert torch.equal(a, b) and torch.equal(a, c)


def test_every_automatic_switch_in_one_run_trains_like_the_plain_learner(tmp_path, monkeypatch):
    """The step picks its learner by itself -- the tree's leaf paths, the one launch per lane, the one launch on the distinct trajectories of a
    work item after DISTINCT_AFTER updates -- and leaves the captured graph for logging steps.  Each pair of modes is tested on its own; here
    they all switch inside ONE run (leaf paths -> a logging step -> replays -> leaf paths forced off: per lane -> a rotation of the
    regularisation nets -> distinct trajectories, a re-capture -> another logging step -> replays), against a trainer that takes the per-lane
    learner eagerly throughout.  Integer per-row sums in every learner, the same nets on the same distinct observations in both runs: the
    same parameters, bit for bit."""
    import os

    from environment.episode import Buffer
    from learn.rnad import RNaD
    from test_hip_bucket import TREES, _native_tree

    tree = _native_tree(**TREES["ternary4"])
    h = tree.handle()
    B, steps = 1 << 14, 18
    monkeypatch.setattr(RNaD, "DISTINCT_AFTER", 11)

    def run(switching, tag):
        os.environ["RNAD_SAVE_DIR"] = str(tmp_path)
        torch.manual_seed(7)
        rn = RNaD(tree=tree, device=DEV, directory_name=f"soup{tag}", batch_size=B, eta=0.2, b1_adam=0.0, lr=1e-3,
                  net_params={"type": "MLP", "max_actions": tree.max_actions, "width": 64})
        rn.initialize()
        rn.tabular_gate = 0
        rn.use_graph = switching
        if not switching:
            rn.leaf_paths, rn.distinct_trajectories = False, False
        with torch.no_grad():
            for p in rn.net_reg_.parameters():
                p.mul_(1.01)
        buf = Buffer(1)
        seen = []
        for i in range(steps):
            if i == 8:  # rnad.py:528-531
                rn.net_reg_.load_state_dict(rn.net_reg.state_dict())
                rn.net_reg.load_state_dict(rn.net_target.state_dict())
            if switching and i == 6:
                rn.leaf_paths = False
            log = {} if i in (4, 13) else None
            rn.train_step(buf, alpha=min(1.0, 0.1 * i), log=log)
            g = rn.__dict__.get("_graph") or {}
            seen.append((rn._leaf_now(h, B, 2 * h.max_depth) is not None, bool(rn._distinct_now()), g.get("graph") is not None and log is None))
            rn.total_steps += 1
        torch.cuda.synchronize()
        return rn, [p.detach().clone() for n in (rn.net, rn.net_target) for p in n.parameters()], seen

    auto, nets_auto, seen = run(True, "a")
    plain, nets_plain, _ = run(False, "p")
    leaf, distinct, replayed = zip(*seen)
    assert all(leaf[:6]) and not any(leaf[6:]), leaf                      # the leaf-path learner, then the one launch per lane
    assert not any(distinct[:10]) and all(distinct[11:]), distinct        # ... then the distinct trajectories of a work item
    assert replayed[3] and not replayed[4] and replayed[9] and replayed[-1], replayed  # replays around the eager logging steps and the re-captures
    for a, b in zip(nets_auto, nets_plain):
        assert torch.equal(a, b)
