"""Scratch helper (not a test): time the CPU port at several thread counts on the current host."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
CODE = r'''
import sys, time
sys.path[:0]=[%r+"/r-nad_amd", %r]
import torch, numpy as np, os
torch.set_num_threads(int(os.environ["OMP_NUM_THREADS"]))
import rnad_hip
from oracle.port import CpuTrainer
tr = rnad_hip.tree_generate(3,1,6,seed=0)
arr = {k: v.numpy() for k,v in tr.items()}; arr["depth_bound"]=6
ct = CpuTrainer(arr); B=2**15
ct.step(4096, seed=0)
t=time.time(); T, a, b = ct.step(B, seed=1); T, a2, b2 = ct.step(B, seed=2); dt=time.time()-t
print(os.environ["OMP_NUM_THREADS"], "threads: env-steps/s %%.0f rollout %%.2fs update %%.2fs" %% (2*B*T/dt, a+a2, b+b2))
''' % (ROOT, ROOT)
for n in sys.argv[1:]:
    env = dict(os.environ, OMP_NUM_THREADS=n)
    subprocess.run([sys.executable, "-c", CODE], env=env)
