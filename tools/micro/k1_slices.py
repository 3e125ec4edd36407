#!/usr/bin/env python3
"""K1 (rnad_observe) as bench.py times it: every launch writes its own [B, 2, A, A] slice of a [T, B, 2, A, A] buffer (906 MB at T = 12:
beyond the Infinity Cache), hipEvents around 3 x T launches.   python tools/micro/k1_slices.py   (RNAD_HIP_SO selects a variant)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "r-nad_amd"))
import torch  # noqa: E402

import rnad_hip  # noqa: E402
from environment.tree import Tree  # noqa: E402

dev = torch.device("cuda:0")
B, T, A = 1 << 20, 12, 3
tree = Tree(device=dev, max_actions=A, max_transitions=1, depth_bound=6)
tree.generate_native(seed=0)
h = tree.handle()
g = torch.Generator(device=dev)
g.manual_seed(0)
idx = torch.randint(1, h.S, (T, B), device=dev, generator=g, dtype=torch.int32)
obs = torch.empty((T, B, 2, A, A), device=dev)
bits = torch.empty((T, B), dtype=torch.uint8, device=dev)


def run():
    for t in range(T):
        rnad_hip.observe(h, idx[t], t & 1, obs=obs[t], mask_bits=bits[t])


run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / (5 * T)
written = B * (2 * A * A * 4 + 1)
print(f"{us:.2f} us per launch; {written / us / 1e6:.2f} TB/s written ({written / us / 1e6 / 8:.2f} of 8 TB/s); with the 4 B index read {(written + 4 * B) / us / 1e6:.2f} TB/s")
