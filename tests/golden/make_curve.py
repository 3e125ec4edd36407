#!/usr/bin/env python3
"""NashConv-vs-update curves of the REFERENCE on the golden `small` tree (reference main.py:55-81 hyper-parameters).

Runs only in the build container (imports /root/reference like make_golden.py).  Writes tests/golden/curve_small.npz:
nashconv[seed, m] for m = 0..M (m = 0 is the untrained net), for a few seeds -- the band the GPU run is compared with
(tests/test_hip_curve.py).  Takes a few minutes of CPU.
"""
import os
import sys

HERE = os.path.dirname(os.path.realpath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (sets up the reference import, stubs and seeding helpers)
import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(8)
torch.multinomial = mg._orig_multinomial  # plain reference sampling here: no recording needed

M, DELTA, B, SEEDS = 12, 100, 512, (0, 1, 2)


def main():
    g = np.load(os.path.join(HERE, "tree_small.npz"))
    tree = mg.ref_tree.Tree(max_actions=3, max_transitions=2, depth_bound=4)
    for key, attr in (("index", "index_tensor"), ("value", "value_tensor"), ("chance", "chance_tensor"),
                      ("expected_value", "expected_value_tensor"), ("legal", "legal_tensor"), ("root_value", "root_value_tensor"),
                      ("solution", "solution_tensor")):
        setattr(tree, attr, torch.tensor(g[key]))
    tree.hash = 1234
    curves = []
    for seed in SEEDS:
        mg.seed_all(1000 + seed)
        rn = mg.ref_rnad.RNaD(tree=tree, device=torch.device("cpu"), directory_name=f"curve{seed}", wandb=False, eta=0.2, bounds=[M],
                              delta_m=[DELTA], lr=1e-3, gamma_averaging=0.01, batch_size=B, logit_clip=2, b1_adam=0.0,
                              net_params={"type": "MLP", "max_actions": 3, "width": 2**8})
        ncs = []
        orig = rn._RNaD__nashconv

        def rec():
            v = orig()
            ncs.append(v)
            return v

        rn._RNaD__nashconv = rec
        rn._RNaD__initialize()
        rn._RNaD__nashconv()  # untrained net (the reference logs from m = 1 on)
        rn._RNaD__resume(checkpoint_mod=10**9, expl_mod=1, log_mod=10**9)
        rn._RNaD__nashconv()
        curves.append(list(ncs))
        print(seed, [round(x, 3) for x in ncs], flush=True)
        assert len(ncs) == M + 1
    np.savez_compressed(os.path.join(HERE, "curve_small.npz"), nashconv=np.array(curves), M=M, delta_m=DELTA, batch=B, eta=0.2, lr=1e-3,
                        gamma_averaging=0.01, seeds=np.array(SEEDS))


if __name__ == "__main__":
    main()
