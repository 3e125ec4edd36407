import os, sys, time, tempfile
sys.path.insert(0, "/root/repo/r-nad_amd")
import torch
import rnad_hip
from environment.episode import Buffer
from environment.tree import Tree
from learn.rnad import RNaD
dev = torch.device("cuda:0")
tree = Tree(device=dev, max_actions=5, max_transitions=4, depth_bound=8, transition_threshold=0.1)
tree.generate_native(seed=0, prune=(7, 8))
os.environ["RNAD_SAVE_DIR"] = tempfile.mkdtemp()
rn = RNaD(tree=tree, device=dev, directory_name="x", batch_size=1 << 20, eta=0.2, b1_adam=0.0, net_params={"type": "MLP", "max_actions": 5, "width": 256})
rn.initialize()
buf = Buffer(1)
for skip in (True, False, True):
    rn.skip_absorbed = skip
    for i in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        rn.train_step(buf, 0.1); rn.total_steps += 1
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"skip={skip} step {i}: host {1e3*(t1-t):.1f} ms, total {1e3*(t2-t):.1f} ms", flush=True)
