#!/usr/bin/env python3
"""How fast the CPU port (oracle/port.py) steps on this host for a few thread / chunk settings (sizing tests/test_hip_e2e.py's full-size cases)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "r-nad_amd"))
import torch
from oracle.port import CpuTrainer
from environment.tree import Tree

t = Tree(device=torch.device("cpu"), max_actions=3, max_transitions=1, depth_bound=6)
t.generate_native(seed=0)
arr = dict(index=t.index_tensor.numpy(), value=t.value_tensor.numpy(), chance=t.chance_tensor.numpy(), expected_value=t.expected_value_tensor.numpy(),
           legal=t.legal_tensor.numpy(), depth_bound=t.depth_bound)
B = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
for threads in (16, 32, 64, 128):
    for chunk in (1 << 14, 1 << 16):
        torch.set_num_threads(threads)
        tr = CpuTrainer(arr, width=256, seed=1, chunk_rows=chunk)
        tr.step(4096, 1)
        t0 = time.perf_counter()
        T, r, u = tr.step(B, 2)
        print(f"B=2^{B.bit_length()-1} torch_threads={threads} chunk={chunk} rollout {r:.2f} s update {u:.2f} s total {time.perf_counter()-t0:.2f} s", flush=True)
