"""Minimal driver, the counterpart of the reference's main.py:9-81: generate a small 3x3 stochastic tree, save it, then run
R-NaD for a few values of eta on the MI355X.  Lives next to the `environment / learn / nn / util` packages, exactly like the
reference's script lives next to its own, and uses only their reference-compatible API.

    python r-nad_amd/main.py [--updates 8] [--steps 100] [--batch 512]
"""
import argparse
import logging
from random import random
from time import time

import torch

from environment.tree import Tree
from learn.rnad import RNaD

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--updates", type=int, default=64, help="regularisation updates m (reference: bounds=[64])")
    ap.add_argument("--steps", type=int, default=100, help="learner steps per update (reference: delta_m=[100])")
    ap.add_argument("--batch", type=int, default=2**9)
    ap.add_argument("--etas", type=float, nargs="*", default=[0, 0.2, 0.5, 1])
    args = ap.parse_args()
    logging.basicConfig(level=logging.INFO)
    if not torch.cuda.is_available():
        raise SystemExit("this build runs the R-NaD hot path on an MI355X only (no CPU fallback)")

    tree = Tree(
        device=torch.device("cuda"),
        max_actions=3,
        max_transitions=2,
        transition_threshold=0.3,
        depth_bound=4,
        depth_bound_lambda=lambda tree: tree.depth_bound - 1 - 2 * (random() < 0.5),
        desc="3x3 stochastic tree, with depth up to 4",
    )
    tree.generate()
    tree.assert_index_is_tree()
    tree.save("small_tree")
    # tree.load("small_tree")  # instead of generate(), to reuse a tree

    timestamp = str(int(time()))
    for idx, eta in enumerate(args.etas):
        same_init_net = None if idx == 0 else f"{timestamp}-eta={args.etas[0]}"  # compare etas from one initial net
        trial = RNaD(
            use_same_init_net_as=same_init_net,
            tree=tree,
            directory_name=f"{timestamp}-eta={eta}",
            device=tree.device,
            wandb=False,
            eta=eta,
            bounds=[args.updates],
            delta_m=[args.steps],
            lr=1 * 10**-3,
            gamma_averaging=0.01,
            batch_size=args.batch,
            logit_clip=2,
            net_params={"type": "MLP", "max_actions": tree.max_actions, "width": 2**8},
        )
        trial.run(log_mod=10, expl_mod=1, checkpoint_mod=args.steps)
        print(f"eta={eta}: NashConv by update:", [round(v, 3) for _, _, v in trial.nashconv_history])
