#!/usr/bin/env python3
"""Golden fixtures of the reference's ON-POLICY step (learn/rnad.py:502-510 with the default one-batch buffer), made by IMPORTING
the reference like make_golden.py (build container only; only the .npz files travel):

    `Episodes(tree, B).generate(net)` played by the LEARNER net itself, then the reference's `RNaD.__learn` on that very batch
    with three other (seeded) nets as target / reg / reg_ -- so the acting policy of every slot IS the learner's pi, which is the
    premise of the compact bucketed learner (csrc/bucket.hip: k_bucket_learn<A, true, .> on row-precomputed operands).

Recorded per tree (c1, small, ragged, a5): the four nets' weights, the trajectory the reference recorded (indices, one-hot actions
as indices, rewards, acting policy, masks), the hyper-parameters, the two losses and the learner's parameter gradients.

Usage:  python tests/golden/make_onpolicy.py      (writes tests/golden/onpolicy_<tree>.npz)
"""
import numpy as np

import make_golden as mg

ref_episode, ref_net, ref_tree, ref_vtrace = mg.ref_episode, mg.ref_net, mg.ref_tree, mg.ref_vtrace


def tree_like_fixture(name):
    """The reference tree of tests/golden/tree_<name>.npz, regenerated from its seeds (and checked against the fixture)."""
    spec = mg.TREE_SPECS[name]
    mg.seed_all(spec["seed"])
    tree = ref_tree.Tree(**spec["kw"], **mg._lambdas(spec))
    tree.generate()
    with np.load(mg.os.path.join(mg.HERE, f"tree_{name}.npz")) as z:
        for key, t in (("index", tree.index_tensor), ("value", tree.value_tensor), ("chance", tree.chance_tensor),
                       ("expected_value", tree.expected_value_tensor), ("legal", tree.legal_tensor)):
            assert np.array_equal(z[key], t.numpy()), f"tree_{name}.npz no longer matches the regenerated tree ({key})"
    return tree


def make(name, batch, seed, eta, alpha, width=32, **over):
    tree = tree_like_fixture(name)
    mg.seed_all(seed)
    nets = [ref_net.MLP(tree.max_actions, width) for _ in range(4)]
    ep = ref_episode.Episodes(tree, batch)
    ep.generate(nets[0])  # the learner plays
    rn = mg._bare_rnad(tree, nets, eta, **over)
    rec = {}
    orig_lv, orig_ln = ref_vtrace.get_loss_v, ref_vtrace.get_loss_nerd

    def lv(*a):
        r = orig_lv(*a)
        rec["loss_v"] = r.detach().clone()
        return r

    def ln(*a, **kw):
        r = orig_ln(*a, **kw)
        rec["loss_nerd"] = r.detach().clone()
        return r

    ref_vtrace.get_loss_v, ref_vtrace.get_loss_nerd = lv, ln
    try:
        rn._RNaD__learn(ep, alpha)
    finally:
        ref_vtrace.get_loss_v, ref_vtrace.get_loss_nerd = orig_lv, orig_ln
    arrays = dict(eta=eta, alpha=alpha, width=width, batch=batch, t_eff=ep.t_eff,
                  indices=ep.indices, actions=ep.actions.argmax(-1), rewards=ep.rewards, policy=ep.policy, masks=ep.masks,
                  loss_v=rec["loss_v"], loss_nerd=rec["loss_nerd"])
    for i, tag in enumerate(("net", "target", "reg", "reg_")):
        arrays.update(mg.state_dict_np(nets[i], f"w_{tag}_"))
    for k, p_ in nets[0].named_parameters():
        arrays["g_net_" + k.replace(".", "_")] = p_.grad
    for k, v_ in over.items():
        arrays["hp_" + k] = v_
    mg.save("onpolicy_" + name, **arrays)


def main():
    make("c1", batch=192, seed=500, eta=0.2, alpha=0.35)
    make("small", batch=384, seed=501, eta=0.2, alpha=0.6)
    make("ragged", batch=320, seed=502, eta=0.5, alpha=0.0, c_bar=0.9, roh_bar=1.2, vtrace_gamma=0.97, beta=1.0, neurd_clip=0.7)
    make("a5", batch=256, seed=503, eta=0.2, alpha=1.0)


if __name__ == "__main__":
    main()
