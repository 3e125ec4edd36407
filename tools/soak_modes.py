#!/usr/bin/env python3
"""Train on the configs[1] tree (depth-6 ternary, S = 66 431) in the three net-evaluation modes from the same seed and print
NashConv of the target net along the way: the modes must track each other (forward == dense exactly; True to rounding)."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "r-nad_amd"))
import torch  # noqa: E402

from environment.episode import Buffer  # noqa: E402
from environment.tree import Tree  # noqa: E402
from learn.rnad import RNaD  # noqa: E402

dev = torch.device("cuda:0")
tree = Tree(device=dev, max_actions=3, max_transitions=1, depth_bound=6)
tree.generate_native(seed=0)
os.environ["RNAD_SAVE_DIR"] = tempfile.mkdtemp()
B, steps, every = 1 << int(os.environ.get("LG", 16)), int(os.environ.get("STEPS", 600)), 100
for mode in (False, "forward", True):
    torch.manual_seed(11)
    rn = RNaD(tree=tree, device=dev, directory_name=f"soak-{mode}", batch_size=B, eta=0.2, b1_adam=0.0, lr=5e-4, delta_m=[200],
              bounds=[10**6], net_params={"type": "MLP", "max_actions": 3, "width": 256})
    rn.initialize()
    rn.tabular = mode
    buf = Buffer(1)
    out, t0 = [], time.perf_counter()
    for i in range(steps):
        m, n = divmod(i, 200)
        if n == 0 and i:
            rn.net_reg_.load_state_dict(rn.net_reg.state_dict())
            rn.net_reg.load_state_dict(rn.net_target.state_dict())
        alpha = 1 if n > 100 else n * 2 / 200
        rn.m = m
        rn.train_step(buf, alpha)
        rn.total_steps += 1
        if (i + 1) % every == 0:
            out.append(round(rn._RNaD__nashconv(), 4))
    torch.cuda.synchronize()
    print(f"tabular={mode!s:8} {steps} steps of 2^{B.bit_length() - 1} episodes in {time.perf_counter() - t0:6.2f} s  NashConv every {every}: {out}", flush=True)
