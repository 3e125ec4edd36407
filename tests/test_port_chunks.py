"""oracle/port.py::CpuTrainer with its MLPs evaluated in row chunks (what the full-size cases of tests/test_hip_e2e.py and bench.py's
cpu_baseline use: cache-sized chunks are ~4.6x faster than whole-batch matrices) steps like the unchunked port: the same rollouts bit
for bit, the same parameters up to the order in which the weight gradient is accumulated."""
import numpy as np
import torch


def test_chunked_port_steps_like_the_unchunked_one():
    from environment.tree import Tree
    from oracle.port import CpuTrainer

    tree = Tree(device=torch.device("cpu"), max_actions=3, max_transitions=2, depth_bound=3, transition_threshold=0.2)
    tree.generate_native(seed=1)
    arrays = dict(index=tree.index_tensor.numpy(), value=tree.value_tensor.numpy(), chance=tree.chance_tensor.numpy(),
                  expected_value=tree.expected_value_tensor.numpy(), legal=tree.legal_tensor.numpy(), depth_bound=tree.depth_bound)
    whole = CpuTrainer(arrays, width=32, seed=1, keep=True)
    chunked = CpuTrainer(arrays, width=32, seed=1, keep=True, chunk_rows=777)  # (not a divisor of T * B: a ragged last chunk)
    for k in range(3):
        whole.step(2048, 7 + k, alpha=0.25 * k)
        chunked.step(2048, 7 + k, alpha=0.25 * k)
        for key in ("indices", "actions", "rewards"):
            assert np.array_equal(whole.last["rollout"][key], chunked.last["rollout"][key]), key
    for (name, p), (_, q) in zip(whole.net.named_parameters(), chunked.net.named_parameters()):
        np.testing.assert_allclose(q.detach().numpy(), p.detach().numpy(), rtol=1e-5, atol=1e-7, err_msg=name)
    for (name, p), (_, q) in zip(whole.net_target.named_parameters(), chunked.net_target.named_parameters()):
        np.testing.assert_allclose(q.detach().numpy(), p.detach().numpy(), rtol=1e-5, atol=1e-7, err_msg=name)
