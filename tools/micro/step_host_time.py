"""Host-enqueue time vs wall time of RNaD.train_step at several batch sizes (is the step launch-bound at 2^17 lanes?)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.realpath(__file__)), "..", "..", "r-nad_amd"))
import torch
from environment.episode import Buffer
from environment.tree import Tree
from learn.rnad import RNaD
dev = torch.device("cuda:0")
tree = Tree(device=dev, max_actions=3, max_transitions=1, depth_bound=6)
tree.generate_native(seed=0)
os.environ["RNAD_SAVE_DIR"] = tempfile.mkdtemp()
for lg in (17, 18, 20):
    rn = RNaD(tree=tree, device=dev, directory_name=f"x{lg}", batch_size=1 << lg, eta=0.2, b1_adam=0.0, net_params={"type": "MLP", "max_actions": 3, "width": 256})
    rn.initialize()
    buf = Buffer(1)
    for i in range(3):
        rn.train_step(buf, 0.1); rn.total_steps += 1
    torch.cuda.synchronize()
    n = 20
    t = time.perf_counter(); host = 0.0
    for i in range(n):
        t0 = time.perf_counter()
        rn.train_step(buf, 0.1); rn.total_steps += 1
        host += time.perf_counter() - t0
    torch.cuda.synchronize()
    tot = time.perf_counter() - t
    print(f"batch 2^{lg}: host enqueue {1e3*host/n:.2f} ms/step, wall {1e3*tot/n:.2f} ms/step", flush=True)
