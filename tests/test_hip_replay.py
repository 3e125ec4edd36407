"""Replay / off-policy path against the reference's own Buffer.sample -> Episodes.collate -> RNaD.__learn (tests/golden/replay_small.npz,
made by tests/golden/make_replay.py): two rollouts by two different actors, one of them shorter (collate pads it in time) and
smaller than its bucket (sample clips), a learner that is neither actor (V-trace importance ratios != 1)."""
import numpy as np
import pytest
import torch

from _util import assert_bits_equal, load

pytestmark = pytest.mark.gpu


def _episodes(G, tree, g, prefix):
    ro = {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}
    return G.episodes_from_golden(tree, ro)


def _collated(G, tree, g):
    from environment.episode import Episodes

    eps = [_episodes(G, tree, g, "e0_"), _episodes(G, tree, g, "e1_")]
    parts = [eps[i].sample(int(g["bucket_sizes"][i]), selected=g[f"selected{i}"]) for i in range(2)]
    return eps, parts, Episodes.collate(parts)


def test_sample_and_collate_reproduce_the_reference_batch():
    import _gpu as G

    g = load("replay_small")
    tree, _ = G.golden_tree("small")
    eps, parts, batch = _collated(G, tree, g)
    assert parts[0].batch_size == min(int(g["bucket_sizes"][0]), eps[0].batch_size) == 40  # sample clips to the source batch (episode.py:245)
    T, B = g["c_indices"].shape
    assert (batch.t_eff + 1, batch.batch_size) == (T, B) and eps[0].t_eff < eps[1].t_eff
    valid = g["c_indices"] != 0
    np.testing.assert_array_equal(G.cpu(batch.indices), g["c_indices"])
    assert_bits_equal(G.cpu(batch.observations), g["c_observations"], "observations")  # zero-padded in time like the reference
    assert_bits_equal(G.cpu(batch.policy), g["c_policy"], "policy")
    assert_bits_equal(G.cpu(batch.rewards), g["c_rewards"], "rewards")
    assert_bits_equal(G.cpu(batch.values), g["c_values"], "values")
    np.testing.assert_array_equal(G.cpu(batch.actions)[valid], g["c_actions"][valid])
    np.testing.assert_array_equal(G.cpu(batch.masks)[valid], g["c_masks"][valid])
    np.testing.assert_array_equal(G.cpu(batch.turns)[valid], g["c_turns"][valid])  # the reference pads `turns` with 0: invalid slots only
    np.testing.assert_array_equal(G.cpu(batch.alive)[:T], valid.sum(1))
    assert float(g["padded_fraction"]) > 0.3


def test_buffer_sample_draws_bucket_sizes_like_the_reference():
    """Buffer.sample makes the reference's numpy call (episode.py:321), so the same numpy seed gives the same bucket sizes."""
    import _gpu as G
    from environment.episode import Buffer

    g = load("replay_small")
    tree, _ = G.golden_tree("small")
    buf = Buffer(2)
    buf.append(_episodes(G, tree, g, "e0_"))
    buf.append(_episodes(G, tree, g, "e1_"))
    np.random.seed(int(g["seed0"]) + 2)
    seen = {}
    real = np.random.multinomial
    np.random.multinomial = lambda *a, **k: seen.setdefault("sizes", real(*a, **k))
    try:
        batch = buf.sample(int(g["batch"]))
    finally:
        np.random.multinomial = real
    np.testing.assert_array_equal(seen["sizes"], g["bucket_sizes"])
    assert batch.batch_size == g["c_indices"].shape[1]


@pytest.mark.parametrize("mode", (False, "forward", True))
def test_learn_on_the_collated_batch_gives_the_reference_gradients(mode):
    import _gpu as G
    from learn.rnad import RNaD

    g = load("replay_small")
    tree, _ = G.golden_tree("small")
    _, _, batch = _collated(G, tree, g)
    A = tree.max_actions
    rn = RNaD.__new__(RNaD)
    rn.tree, rn.device = tree, G.DEV
    rn.net, rn.net_target = G.mlp_from(g, A, "w_net_"), G.mlp_from(g, A, "w_target_")
    rn.net_reg, rn.net_reg_ = G.mlp_from(g, A, "w_reg_"), G.mlp_from(g, A, "w_reg__")
    rn.eta, rn.c_bar, rn.roh_bar, rn.vtrace_gamma = float(g["eta"]), 1, 1, 1
    rn.neurd_clip, rn.beta, rn.grad_clip = 10**3, 2, 10**3
    rn.value_weight, rn.neurd_weight, rn.epsilon_threshold, rn.n_discrete = 1, 1, 0.03, 32
    rn.tabular, rn.tabular_gate = mode, 0
    log = {}
    rn._RNaD__learn(batch, float(g["alpha"]), log=log)
    # the actors differ from the learner: importance ratios are not 1 on this batch
    pi = g["pi"]
    ratio = (g["c_actions"] * pi).sum(-1) / np.maximum((g["c_actions"] * g["c_policy"]).sum(-1), 1e-12)
    assert np.abs(ratio[g["c_indices"] != 0] - 1).max() > 0.05
    for k, p in rn.net.named_parameters():
        want = g["g_net_" + k.replace(".", "_")]
        scale = np.abs(want).max() + 1e-12
        np.testing.assert_allclose(G.cpu(p.grad), want, rtol=1e-4, atol=2e-6 * scale, err_msg=k)
    np.testing.assert_allclose(log["loss_v"], g["loss_v"], rtol=2e-5)
    np.testing.assert_allclose(log["loss_nerd"], g["loss_nerd"], rtol=1e-4, atol=1e-6)
