"""CPU port of one full R-NaD iteration (rollout + update) for bench.py's `cpu_baseline` leg.

TEST INFRASTRUCTURE ONLY (see oracle/rnad_oracle.c).  Everything around the MLP is the C oracle; the MLP forward /
backward, Adam and the EMA are PyTorch on the CPU -- exactly the library the reference itself runs on when no GPU is
present.  The sequence mirrors reference learn/rnad.py:495-526: Episodes.generate -> __learn -> Adam -> EMA target.
"""
import time

import numpy as np
import torch

from oracle import oracle


class CpuMLP(torch.nn.Module):
    """reference nn/net.py:18-35 (parameters and layer layout only)."""

    def __init__(self, A, width):
        super().__init__()
        self.value_fc0 = torch.nn.Linear(2 * A * A, width)
        self.value_fc1 = torch.nn.Linear(width, 1)
        self.policy_fc0 = torch.nn.Linear(2 * A * A, width)
        self.policy_fc1 = torch.nn.Linear(width, A)

    def logits(self, x):
        return self.policy_fc1(torch.relu(self.policy_fc0(x))), self.value_fc1(torch.relu(self.value_fc0(x)))

    def weights(self):
        sd = self.state_dict()
        return [sd[k].detach().numpy() for k in oracle.MLP_KEYS]


class CpuTrainer:
    def __init__(self, tree_arrays, width=256, lr=5e-5, eta=0.2, gamma_averaging=0.001, seed=0):
        torch.manual_seed(seed)
        self.tree = tree_arrays
        self.A = tree_arrays["index"].shape[-1]
        self.net = CpuMLP(self.A, width)
        self.net_target, self.net_reg, self.net_reg_ = (CpuMLP(self.A, width) for _ in range(3))
        for n in (self.net_target, self.net_reg, self.net_reg_):
            n.load_state_dict(self.net.state_dict())
        self.opt = torch.optim.Adam(self.net.parameters(), lr=lr, betas=(0.0, 0.999), eps=1e-8)
        self.eta, self.gamma_averaging = eta, gamma_averaging
        self.T_cap = 2 * int(tree_arrays["depth_bound"])

    def step(self, B, seed, alpha=0.5):
        """One iteration; returns (T, rollout seconds, update seconds)."""
        A = self.A
        t0 = time.perf_counter()
        ro = oracle.rollout(self.tree, self.net.weights(), B, self.T_cap, seed)  # episode.py:175-230
        t1 = time.perf_counter()
        T = ro["T"]
        obs = torch.from_numpy(ro["observations"].reshape(T * B, 2 * A * A))
        logit, v = self.net.logits(obs)  # rnad.py:373
        with torch.no_grad():
            _, v_t = self.net_target.logits(obs)
            lr, _ = self.net_reg.logits(obs)
            lr_, _ = self.net_reg_.logits(obs)
        masks = ro["masks"].reshape(T * B, A)
        pi, log_pi = oracle.policy_head(logit.detach().numpy(), masks)
        _, log_r = oracle.policy_head(lr.numpy(), masks)
        _, log_r_ = oracle.policy_head(lr_.numpy(), masks)
        pip = oracle.process_policy(pi, masks, 32, 0.03)  # rnad.py:374
        lpol = (log_pi - (np.float32(alpha) * log_r + np.float32(1 - alpha) * log_r_)).reshape(T, B, A)  # :382
        valid = (ro["indices"] != 0).astype(np.float32)
        turns = np.broadcast_to((np.arange(T) % 2)[:, None], (T, B)).astype(np.int64)
        a_oh = np.eye(A, dtype=np.float32)[ro["actions"]]
        vts, hps, qs = [], [], []
        for p in range(2):  # rnad.py:384-406
            rew = ro["rewards"] if p == 0 else -ro["rewards"]
            vt, hp, q = oracle.vtrace(v_t.numpy().reshape(T, B, 1), valid, turns, ro["policy"], pip.reshape(T, B, A), lpol, a_oh, rew,
                                      p, self.eta, 1.0, 1.0, 1.0, 1.0)
            vts.append(vt); hps.append(hp); qs.append(q)  # noqa: E702
        _, dv = oracle.loss_v(v.detach().numpy(), vts[0], vts[1], hps[0], hps[1])  # rnad.py:407
        _, dl = oracle.loss_nerd(logit.detach().numpy(), pip, qs[0], qs[1], valid, turns, masks, 1e3, 2.0)  # :412-422
        torch.autograd.backward([logit, v], [torch.from_numpy(dl).view_as(logit), torch.from_numpy(dv).view_as(v)])  # :425
        torch.nn.utils.clip_grad_norm_(self.net.parameters(), 1e3)
        self.opt.step()
        self.opt.zero_grad()
        with torch.no_grad():  # EMA target, rnad.py:516-523
            for p_t, p_n in zip(self.net_target.parameters(), self.net.parameters()):
                p_t.copy_(self.gamma_averaging * p_n + (1 - self.gamma_averaging) * p_t)
        t2 = time.perf_counter()
        return T, t1 - t0, t2 - t1
