#!/usr/bin/env python3
"""Golden fixture of the reference's replay / off-policy path, made by IMPORTING the
reference like make_golden.py (build container only; only the .npz travels):

    two `Episodes.generate` batches played by two DIFFERENT actor nets on the pruned `small` tree (ragged lengths, so
    `collate` pads) -> reference `Buffer(2)` -> `Buffer.sample(batch)` = numpy multinomial bucket sizes + `random.sample`
    lane subsets + `Episodes.collate` (episode.py:243-333) -> reference `RNaD.__learn` on the collated batch with a learner
    net that is neither actor, so the V-trace importance ratios differ from 1 (rnad.py:502-510 with n_batches_per_buffer = 2).

Recorded: both source trajectories, the draws (bucket sizes, selected lanes), the collated batch, every intermediate of
`__learn` that make_golden.make_learn records, and the learner's parameter gradients.

Usage:  python tests/golden/make_replay.py      (writes tests/golden/replay_small.npz)
"""
import random

import numpy as np

import make_golden as mg

ref_episode, ref_net, ref_vtrace = mg.ref_episode, mg.ref_net, mg.ref_vtrace


def play(tree, batch, seed, width):
    mg.seed_all(seed)
    net = ref_net.MLP(tree.max_actions, width)
    ep = ref_episode.Episodes(tree, batch)
    ep.generate(net)
    return ep, net


def ep_arrays(ep, prefix):
    return {prefix + k: getattr(ep, k) for k in ("indices", "turns", "observations", "masks", "policy", "actions", "rewards", "values")} | {
        prefix + "t_eff": ep.t_eff}


def main():
    mg.seed_all(mg.TREE_SPECS["small"]["seed"])
    tree = mg.make_tree("small")  # same seed -> the committed tree_small.npz again (rewritten identically)
    width, batch = 32, 120
    # two rollouts by two different (seeded) actors; different batch sizes and, on this pruned tree, different lengths
    # (the first seed >= 400 whose small batch happens to end early, so that `collate` has to pad it in time, episode.py:271-282)
    seed0 = 400
    while True:
        ep0, _ = play(tree, 40, seed=seed0, width=width)
        if ep0.t_eff < 2 * tree.depth_bound - 1:
            break
        seed0 += 1
    ep1, _ = play(tree, 128, seed=seed0 + 1, width=width)
    assert ep1.t_eff > ep0.t_eff
    buf = ref_episode.Buffer(2)
    buf.append(ep0)
    buf.append(ep1)

    draws = {}
    real_multinomial, real_sample = np.random.multinomial, random.sample

    def rec_multinomial(n, pvals, *a, **k):
        r = real_multinomial(n, pvals, *a, **k)
        draws["bucket_sizes"] = np.asarray(r).copy()
        return r

    def rec_sample(population, k, *a, **kw):
        r = real_sample(population, k, *a, **kw)
        draws.setdefault("selected", []).append(np.asarray(r, dtype=np.int64))
        return r

    mg.seed_all(seed0 + 2)
    np.random.multinomial, random.sample = rec_multinomial, rec_sample
    try:
        collated = buf.sample(batch)
    finally:
        np.random.multinomial, random.sample = real_multinomial, real_sample
    assert len(draws["selected"]) == 2 and sum(draws["bucket_sizes"]) == batch
    assert collated.t_eff == max(ep0.t_eff, ep1.t_eff)

    # the reference's __learn on the collated batch; learner / target / reg / reg_ are four fresh nets (seed 403)
    import types

    holder = {}
    real_save = mg.save
    mg.save = lambda name, **arrays: holder.update(arrays)  # capture instead of writing learn_<name>.npz
    try:
        mg.make_learn("replay", tree, collated, eta=0.2, alpha=0.45, seed=seed0 + 3, width=width)
    finally:
        mg.save = real_save
    assert isinstance(holder, dict) and "g_net_value_fc0_weight" in holder
    del types
    valid = np.asarray(holder["valid"])
    real_save(
        "replay_small",
        **ep_arrays(ep0, "e0_"), **ep_arrays(ep1, "e1_"), **ep_arrays(collated, "c_"),
        bucket_sizes=draws["bucket_sizes"], selected0=draws["selected"][0], selected1=draws["selected"][1],
        batch=batch, seed0=seed0, padded_fraction=float(1.0 - valid.mean()),
        **holder,
    )
    print("t_eff", ep0.t_eff, ep1.t_eff, "->", collated.t_eff, "bucket sizes", draws["bucket_sizes"], "valid fraction", valid.mean())


if __name__ == "__main__":
    main()
