"""RNaD.train_step replayed from a captured hipGraph is the eager step, bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


_BUFFERS = {}  # the replay buffer of each _run (the captured graph is bound to it)


def _run(tree, tmp_path, use_graph, steps, rotate_at=None, tag=""):
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
        rn.train_step(buf, alpha=min(1.0, 0.15 * i))
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
