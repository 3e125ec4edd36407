"""Data-parallel protocol of RNaD.__learn on 2 ranks (gloo, CPU).

The HIP kernels cannot run here, so THIS TEST swaps `rnad_hip.learn_fused` for an oracle-backed stand-in (test
infrastructure; the product never does that) and checks what the distributed code is responsible for: each rank sees
half of the episodes, the two loss normalisers are all-reduced before gradients are scaled, the parameter gradients are
all-reduced in one bucket -- and the result equals the reference's full-batch gradients from the golden fixture.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _util import load, load_tree, mlp_weights

NAME = "small_eta0.2"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_learn_fused(indices, mask_bits, actions, rewards, mu, logit, v, v_target_net, logit_reg, logit_reg_, norm, hp, want_aux=False):
    """CPU stand-in with the signature of rnad_hip.learn_fused, built from the oracle's single functions."""
    from oracle import oracle

    T, B, A = mu.shape
    n = lambda t: t.detach().cpu().numpy()  # noqa: E731
    masks = ((n(mask_bits)[..., None] >> np.arange(A)) & 1).astype(np.float32)
    pi, log_pi = oracle.policy_head(n(logit).reshape(T, B, A), masks)
    _, log_r = oracle.policy_head(n(logit_reg).reshape(T, B, A), masks)
    _, log_r_ = oracle.policy_head(n(logit_reg_).reshape(T, B, A), masks)
    pip = oracle.process_policy(pi, masks, hp.n_disc, hp.eps_threshold)
    lpol = log_pi - (np.float32(hp.alpha) * log_r + np.float32(hp.one_minus_alpha) * log_r_)
    valid = (n(indices) != 0).astype(np.float32)
    turns = np.broadcast_to((np.arange(T) % 2)[:, None], (T, B)).astype(np.int64)
    a_oh = np.eye(A, dtype=np.float32)[n(actions)]
    vts, hps, qs = [], [], []
    for p in range(2):
        rew = n(rewards) if p == 0 else -n(rewards)
        vt, hpd, q = oracle.vtrace(n(v_target_net).reshape(T, B, 1), valid, turns, n(mu), pip, lpol, a_oh, rew, p, hp.eta, hp.lambda_,
                                   hp.c, hp.rho, hp.gamma)
        vts.append(vt); hps.append(hpd); qs.append(q)  # noqa: E702
    dv = np.zeros((T, B), np.float32)
    dl = np.zeros((T, B, A), np.float32)
    g_norm = n(norm)
    for p in range(2):  # per player: local normaliser -> global normaliser (gradients are linear in 1 / norm)
        zeros_m = np.zeros_like(hps[p])
        m0, m1 = (hps[p], zeros_m) if p == 0 else (zeros_m, hps[p])
        local = max(float(hps[p].sum()), 1.0)
        _, dvp = oracle.loss_v(n(v).reshape(T, B, 1), vts[0], vts[1], m0, m1)
        q0, q1 = (qs[0], np.zeros_like(qs[1])) if p == 0 else (np.zeros_like(qs[0]), qs[1])
        only_p = valid * (turns == p)
        _, dlp = oracle.loss_nerd(n(logit).reshape(T, B, A), pip, q0, q1, only_p, np.full_like(turns, p), masks, hp.clip, hp.threshold)
        scale = np.float32(local / max(float(g_norm[p]), 1.0))
        dv += dvp.reshape(T, B) * scale * hp.w_v
        dl += dlp * scale * hp.w_n
    return torch.from_numpy(dl), torch.from_numpy(dv), torch.zeros(2, dtype=torch.float64), None, None, None


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import rnad_hip
        from _gpu import mask_bits_of  # pure numpy helper
        from environment.episode import Episodes
        from environment.tree import Tree
        from learn.rnad import RNaD
        from nn.net import MLP
        from oracle import oracle

        rnad_hip.learn_fused = _oracle_learn_fused
        g, ro = load("learn_" + NAME), load("rollout_small")
        tg = load_tree("small")
        A = tg["index"].shape[-1]
        np.random.seed(0)
        tree = Tree(max_actions=A, max_transitions=tg["index"].shape[1], depth_bound=4)

        def net(prefix):
            w = mlp_weights(g, prefix)
            m = MLP(A, w[0].shape[0], device=torch.device("cpu"))
            m.load_state_dict(dict(zip(oracle.MLP_KEYS, [torch.as_tensor(x) for x in w])))
            return m

        rn = RNaD.__new__(RNaD)
        rn.tree, rn.device = tree, torch.device("cpu")
        rn.net, rn.net_target, rn.net_reg, rn.net_reg_ = net("w_net_"), net("w_target_"), net("w_reg_"), net("w_reg__")
        rn.eta, rn.c_bar, rn.roh_bar, rn.vtrace_gamma = float(g["eta"]), 1, 1, 1
        rn.neurd_clip, rn.beta, rn.grad_clip = 10**3, 2, 10**3
        rn.value_weight, rn.neurd_weight, rn.epsilon_threshold, rn.n_discrete = 1, 1, 0.03, 32

        T, B = ro["indices"].shape
        half = B // world
        sl = slice(rank * half, (rank + 1) * half)
        ep = Episodes.__new__(Episodes)
        ep.tree, ep.batch_size, ep.t_eff, ep._lazy = tree, half, T - 1, {}
        ep.indices = torch.as_tensor(ro["indices"][:, sl].astype(np.int32))
        ep.observations = torch.as_tensor(ro["observations"][:, sl].copy())
        ep.mask_bits = torch.as_tensor(mask_bits_of(ro["masks"][:, sl]))
        ep.policy = torch.as_tensor(ro["policy"][:, sl].copy())
        ep.action_idx = torch.as_tensor(ro["actions"][:, sl].argmax(-1).astype(np.int32))
        ep.rewards = torch.as_tensor(ro["rewards"][:, sl].copy())
        alive = np.zeros(T + 1, np.int32)
        alive[:T] = (ro["indices"][:, sl] != 0).sum(1)
        ep.alive = torch.as_tensor(alive)

        rn._RNaD__learn(ep, float(g["alpha"]))
        grads = {k: p.grad.numpy().copy() for k, p in rn.net.named_parameters()}
        np.savez(os.path.join(out_dir, f"grads_{rank}.npz"), **grads)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradients_equal_reference_full_batch(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = load("learn_" + NAME)
    r0, r1 = (np.load(tmp_path / f"grads_{r}.npz") for r in range(world))
    for k in r0.files:
        np.testing.assert_array_equal(r0[k], r1[k])  # all-reduced: identical on every rank
        want = g["g_net_" + k.replace(".", "_")]
        scale = np.abs(want).max() + 1e-12
        np.testing.assert_allclose(r0[k], want, rtol=1e-4, atol=2e-6 * scale, err_msg=k)


# ---------------------------------------------------------------------------------------- shared run directory
def _init_worker(rank, world, port, root):
    """Two ranks, ONE saved_runs directory: fresh start, then a resume from a later checkpoint that only rank 0 wrote."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RNAD_SAVE_DIR=root)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from environment.tree import Tree
        from learn.rnad import RNaD

        tg = load_tree("c1")
        tree = Tree(max_actions=2, max_transitions=1, depth_bound=3)
        tree.hash = 1234
        torch.manual_seed(100 + rank)  # different initial nets per rank unless rank 0's are broadcast

        def make():
            return RNaD(tree=tree, device=torch.device("cpu"), directory_name="shared", batch_size=64, eta=0.2, b1_adam=0.0,
                        net_params={"type": "MLP", "max_actions": 2, "width": 16})

        rn = make()
        rn.initialize()
        assert (rn.m, rn.n) == (0, 0)
        store_dir = os.path.join(root, "saved_runs", "shared")
        assert os.path.exists(os.path.join(store_dir, "params")) and os.path.exists(os.path.join(store_dir, "0", "0"))  # behind the barrier
        w = torch.cat([p.detach().reshape(-1) for p in rn.net.parameters()])
        both = [torch.zeros_like(w) for _ in range(world)]
        dist.all_gather(both, w)
        assert torch.equal(both[0], both[1]), "ranks must start from rank 0's weights"
        # rank 0 trains on (pretend) and checkpoints at (m, n) = (1, 2); then everybody restarts
        rn.m, rn.n, rn.total_steps = 1, 2, 7
        with torch.no_grad():
            for p in rn.net.parameters():
                p.add_(1.0)
        rn._RNaD__save_checkpoint()
        dist.barrier()
        rn2 = make()
        rn2.lr = 123.0  # overwritten by the stored params on resume
        rn2.initialize()
        assert (rn2.m, rn2.n, rn2.total_steps) == (1, 2, 7) and rn2.lr == rn.lr
        w2 = torch.cat([p.detach().reshape(-1) for p in rn2.net.parameters()])
        assert torch.equal(w2, both[0] + 1.0)
        assert sorted(os.listdir(store_dir)) == ["0", "1", "params"], os.listdir(store_dir)  # no temporary files left behind
        del tg
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_share_one_run_directory(tmp_path):
    """every rank used to scan the directory on its own while rank 0 was writing into it."""
    mp.spawn(_init_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
