"""End to end: K steps of the step bench.py times -- RNaD.train_step in its default configuration (distinct observations, compact
bucketed rollout, rollout + learner in one launch, folded backward, one-launch optimiser tail; 3 eager steps, the capture, replays of the
hipGraph) -- against the CPU port of reference learn/rnad.py:495-526 (oracle/port.py::CpuTrainer: Episodes.generate -> __learn -> clip ->
Adam -> EMA target, the C oracle around a PyTorch-CPU MLP) from the SAME initial weights, seeds and alphas.

What is compared, per step:  the alive counts; the trajectory (states, actions, the episode's reward) of every lane, bit for bit, except
on lanes whose draw flipped -- the two sides evaluate the policy head with different fp32 summation orders (MFMA tiles vs sgemm), so a
running sum of the inverse CDF can fall on the other side of a lane's uniform: probability ~ |delta p| ~ 1e-7 per decision; the flips are
COUNTED and bounded;  and after the K steps the learner's parameters and the EMA target.

Tolerances (measured, profiles/r05_parity_errors.json): Adam with beta1 = 0 turns a gradient entry into lr * g / (sqrt(v_hat) + eps), i.e.
the SIGN of g on the first step: an entry whose gradient is below the fp32 summation noise of ~1e5 addends moves by +-lr whichever side
computes it, so the parameter comparison is `|a - b| <= atol + rtol * |b|` with atol = 2e-6 on all but a counted handful of entries,
every one of which must lie within the K * 2 * lr such a sign flip can produce.
"""
import os
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")
LR = 5e-5  # the reference's default (rnad.py:45)


def _arrays(tree):
    return dict(index=tree.index_tensor.cpu().numpy(), value=tree.value_tensor.cpu().numpy(), chance=tree.chance_tensor.cpu().numpy(),
                expected_value=tree.expected_value_tensor.cpu().numpy(), legal=tree.legal_tensor.cpu().numpy(),
                depth_bound=tree.depth_bound)


def _trainer(tree, B, width, lazy, leaf=None):
    from learn.rnad import RNaD

    os.environ["RNAD_SAVE_DIR"] = tempfile.mkdtemp(prefix="rnad_e2e_")
    torch.manual_seed(11)
    rn = RNaD(tree=tree, device=DEV, directory_name="e2e", batch_size=B, eta=0.2, b1_adam=0.0, lr=LR,
              net_params={"type": "MLP", "max_actions": tree.max_actions, "width": width})
    rn.initialize()
    rn.lazy_rows = lazy
    rn.leaf_paths = leaf
    rn.tabular_gate = 0  # the per-row mode whatever the ratio of tree to batch (the test's batches are small next to configs[1]'s)
    # four different nets, as in the middle of a run (rnad.py:528-531 rotates them): every term of log_policy_reg (:382) is live
    g = torch.Generator(device="cpu")
    g.manual_seed(5)
    with torch.no_grad():
        for k, module in enumerate((rn.net_target, rn.net_reg, rn.net_reg_)):
            for p in module.parameters():
                p.add_((0.02 * (k + 1) * torch.randn(p.shape, generator=g)).to(DEV))
    return rn


# depth: the configs[1] tree (A = 3, C = 1, terminal values +-1) at that depth; a name: a tree of tests/test_hip_bucket.py::TREES -- "pruned"
# (A = 3, C = 2, ragged episode lengths) and "a5c4" (the configs[3] shape: A = 5, C = 4, pruned) take RNaD's lazy-rows step with the staged
# actor, the default on trees that are large next to the batch
# r06, the last two: the BENCHMARKED size -- configs[1]'s tree at 2^20 lanes, the very cut, sort tile and kernel instantiations bench.py
# times (3 eager steps, then the capture and its first replay) -- and an 8-GPU rank's 2^19 lanes with the leaf-path learner forced (the learner that
# carries the N = 1 point of configs[2]).  They run last (the CPU port takes ~20 s per step at that size, its MLP
# in row chunks so that the host's memory stays bounded).
@pytest.mark.parametrize("depth,log2_B,width,lazy,K,leaf",
                         ((4, 14, 64, False, 8, None), (4, 16, 256, False, 8, None), (6, 16, 256, False, 8, None), (6, 18, 256, False, 8, None),
                          (6, 16, 256, None, 8, None), ("pruned", 14, 64, True, 8, None), ("a5c4", 14, 64, True, 8, None),
                          ("pruned", 14, 64, False, 8, None), ("a5c4", 14, 256, True, 8, None),  # (width 256: the split-precision value heads of configs[3])
                          pytest.param(6, 19, 256, False, 4, True, marks=pytest.mark.last),
                          pytest.param(6, 20, 256, False, 4, None, marks=pytest.mark.last)))
def test_default_step_trains_like_the_cpu_port(depth, log2_B, width, lazy, K, leaf):
    from environment.episode import Buffer
    from oracle.port import CpuTrainer
    from test_hip_bucket import TREES, _native_tree

    regular = not isinstance(depth, str)
    tree = _native_tree(A=3, C=1, depth=depth, seed=0) if regular else _native_tree(**TREES[depth])
    B = 1 << log2_B
    rn = _trainer(tree, B, width, lazy, leaf)
    h = tree.handle()
    T = 2 * h.max_depth
    cpu = CpuTrainer(_arrays(tree), width=width, lr=LR, eta=rn.eta, gamma_averaging=rn.gamma_averaging, n_discrete=rn.n_discrete,
                     epsilon_threshold=rn.epsilon_threshold, neurd_clip=rn.neurd_clip, logit_clip=rn.beta, grad_clip=rn.grad_clip,
                     state_dicts=[m.state_dict() for m in (rn.net, rn.net_target, rn.net_reg, rn.net_reg_)], keep=True,
                     chunk_rows=1 << 14 if log2_B >= 18 else None)  # (cache-sized chunks: 4.6x faster than whole-batch matrices)
    torch.set_num_threads(min(16, os.cpu_count() or 8) if log2_B >= 18 else 8)  # (measured on the 256-thread host: 16 threads 5.4 s per 2^18-lane step, 64 threads 19.7 s)
    buf = Buffer(1)
    delta_m = 64
    rn.alpha_ahead = lambda k: rn.alpha_of(rn.total_steps + k, delta_m)
    flipped_total, decisions = 0, 0
    modes = []
    for step in range(K):
        alpha = rn.alpha_of(step, delta_m)  # rnad.py:497: 0, 1/32, 2/32, ...
        rn.train_step(buf, alpha)
        ep = rn.last_episodes
        g = rn.__dict__.get("_graph") or {}
        modes.append("replay" if g.get("graph") is not None else "eager")
        seed = int(ep.seed)
        cpu.step(B, seed, alpha=alpha)
        ro = cpu.last["rollout"]
        Tc = ro["T"]  # (the port stops once every lane is absorbed, episode.py:194; the GPU step plays the whole window and masks)
        assert Tc == T if regular else Tc <= T
        # ---- the batch the GPU played against the batch the port played: same lanes, same seed, (nearly) the same policy bits
        lanes = ep.lane_ids.long().cpu().numpy()
        assert np.array_equal(np.sort(lanes), np.arange(B))
        idx_all = ep.indices.cpu().numpy().astype(np.int64)  # [T, B] in bucket order: column j is lane lanes[j]
        assert not idx_all[Tc:].any() or flipped_total, "beyond the port's last step every lane is absorbed"
        idx = idx_all[:Tc]
        want_idx = ro["indices"][:, lanes]
        same = (idx == want_idx).all(0)
        act = ep.action_idx.cpu().numpy().astype(np.int64)[:Tc]
        live = want_idx != 0
        same &= ((act == ro["actions"][:, lanes]) | ~live).all(0)
        rew = ep.rewards.cpu().numpy()[:Tc]
        same &= (rew.view(np.uint32) == ro["rewards"][:, lanes].view(np.uint32)).all(0)
        flipped = int((~same).sum())
        flipped_total += flipped
        decisions += int(live.sum()) * 3 // 2
        # ~1e-7 per decision: a handful per million lanes at most
        assert flipped <= max(2, B * T // 250_000), f"step {step}: {flipped} of {B} lanes played another episode than the port"
        if flipped == 0:
            np.testing.assert_array_equal(ep.alive.cpu().numpy()[:Tc], live.sum(1))
        assert ep._compact is not None, "the compact bucketed rollout ran"
        rn.total_steps += 1
    torch.cuda.synchronize()
    assert modes[:3] == ["eager"] * 3 and modes[-1] == "replay", modes
    g = rn._graph
    assert g["graph"] is not None and not g["failed"], "the step was captured and replayed"
    if leaf:
        assert rn._leaf_now(h, B, T) is not None and rn._fuse_now(), "the leaf-path learner ran"
    if lazy is False and regular:
        assert rn._dedup_now(h, None, False, False, rn._fold()) is not None, "distinct observations are on for this tree"
    if lazy is not False and not regular:
        assert rn.last_rows is not None and getattr(rn.last_episodes, "staged_rows", None) is not None, "lazy rows with the staged actor ran"
    # ---- parameters and EMA target after K updates
    outliers, n, worst = 0, 0, 0.0
    for name, got_module, want_module in (("net", rn.net, cpu.net), ("target", rn.net_target, cpu.net_target)):
        want_sd = want_module.state_dict()
        for k, p in got_module.state_dict().items():
            a, b = p.detach().cpu().numpy().astype(np.float64).ravel(), want_sd[k].numpy().astype(np.float64).ravel()
            err = np.abs(a - b)
            bad = err > 2e-6 + 1e-4 * np.abs(b)
            n += a.size
            worst = max(worst, float(err.max()))
            if name == "net":
                outliers += int(bad.sum())
                # a gradient entry inside the summation noise: Adam (beta1 = 0) moves it by +-lr per step on either side
                assert err.max() <= K * 2 * LR * 1.01, f"{name}.{k}: {err.max():.3e}"
                np.testing.assert_allclose(a[~bad], b[~bad], rtol=1e-4, atol=2e-6, err_msg=f"{name}.{k}")
            else:  # the EMA target has moved by gamma_averaging of the net's steps: everything within the tolerance
                np.testing.assert_allclose(a, b, rtol=1e-4, atol=2e-6, err_msg=f"{name}.{k}")
    torch.set_num_threads(8)
    stats = dict(depth=depth, log2_B=log2_B, width=width, lazy=lazy, leaf=leaf, steps=K, flipped_lanes=flipped_total, decisions=decisions,
                 flipped_per_1e7_decisions=flipped_total * 1e7 / max(decisions, 1), parameters=n, parameters_outside_tolerance=outliers,
                 largest_parameter_error=worst, modes=modes)
    print("e2e", stats)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        import json

        with open(os.path.join(out, "e2e_stats.jsonl"), "a") as f:
            f.write(json.dumps(stats) + "\n")
    # the entries outside the tolerance are sign flips of noise-level gradient entries: a counted handful
    assert outliers <= max(8, n // 2000), f"{outliers} of {n} parameters outside atol 2e-6 + rtol 1e-4"
