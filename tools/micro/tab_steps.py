"""A few RNaD.tabular steps on the c2 tree (for rocprofv3 --kernel-trace --stats)."""
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
rn = RNaD(tree=tree, device=dev, directory_name="x", batch_size=1 << int(os.environ.get("LG", 20)), eta=0.2, b1_adam=0.0,
          net_params={"type": "MLP", "max_actions": 3, "width": 256})
rn.initialize()
with torch.no_grad():
    for p in rn.net_reg_.parameters():
        p.mul_(1.001)
rn.tabular = {"full": True, "forward": "forward", "off": False}[os.environ.get("TAB", "full")]
buf = Buffer(1)
for i in range(3):
    rn.train_step(buf, 0.1); rn.total_steps += 1
torch.cuda.synchronize()
n = 10
t = time.perf_counter()
for i in range(n):
    rn.train_step(buf, 0.1); rn.total_steps += 1
torch.cuda.synchronize()
print(f"tabular={rn.tabular} step: {1e3 * (time.perf_counter() - t) / n:.3f} ms")
