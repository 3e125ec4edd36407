"""RNaD.train_step replayed from a captured hipGraph is the eager step, bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


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
