#!/usr/bin/env python3
"""K1 (rnad_observe) alone on the c2 tree at B = 2^20, once per env step of a real rollout, plus a calibration copy of a
known byte count -- run under `rocprofv3 --pmc FETCH_SIZE` and, separately, `--pmc WRITE_SIZE` (tools/k1_pmc.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "r-nad_amd"))
import torch  # noqa: E402

import rnad_hip  # noqa: E402
from environment.episode import Episodes  # noqa: E402
from environment.tree import Tree  # noqa: E402
from nn.net import MLP  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
B = 1 << 20
tree = Tree(device=dev, max_actions=3, max_transitions=1, depth_bound=6)
tree.generate_native(seed=0)
ep = Episodes(tree, B, seed=1)
ep.generate(MLP(3, 256, device=dev))
T = ep.t_eff + 1
# every launch writes its own [B, 2, A, A] slice of a [T, B, 2, A, A] buffer (906 MB at T = 12: larger than the 256 MiB Infinity Cache, so the
# fabric counters cannot be served from it -- one 75 MB buffer rewritten 36 times could be)
obs_all = torch.empty((T, B, 2, 3, 3), device=dev)
bits_all = torch.empty((T, B), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
# calibration: a 160 MiB float4 streaming copy = 167 772 160 B read + 167 772 160 B written (same bytes as one K1 launch's model)
src = torch.randn((B * 40,), device=dev)
dst = torch.empty_like(src)
for _ in range(3):
    dst.copy_(src)
torch.cuda.synchronize()
for rep in range(3):
    for t in range(T):
        rnad_hip.observe(tree.handle(), ep.indices[t], t & 1, obs=obs_all[t], mask_bits=bits_all[t])
torch.cuda.synchronize()
print("done", T)
