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
    """The hyper-parameters are the reference's defaults (rnad.py:40-64); `state_dicts` (net, net_target, net_reg, net_reg_: the four
    state_dicts of a trainer, reference key names) starts from given weights instead of fresh ones -- how tests/test_hip_e2e.py runs this
    port in lock step with the GPU trainer.  keep=True: the last step's rollout and gradients stay in self.last."""

    def __init__(self, tree_arrays, width=256, lr=5e-5, eta=0.2, gamma_averaging=0.001, seed=0, state_dicts=None, n_discrete=32,
                 epsilon_threshold=0.03, neurd_clip=1e3, logit_clip=2.0, grad_clip=1e3, betas=(0.0, 0.999), eps=1e-8, keep=False, chunk_rows=None):
        torch.manual_seed(seed)
        self.tree = tree_arrays
        self.A = tree_arrays["index"].shape[-1]
        self.net = CpuMLP(self.A, width)
        self.net_target, self.net_reg, self.net_reg_ = (CpuMLP(self.A, width) for _ in range(3))
        for n in (self.net_target, self.net_reg, self.net_reg_):
            n.load_state_dict(self.net.state_dict())
        if state_dicts is not None:
            for n, sd in zip((self.net, self.net_target, self.net_reg, self.net_reg_), state_dicts):
                n.load_state_dict({k: v.detach().to("cpu", torch.float32).clone() for k, v in sd.items()})
        self.opt = torch.optim.Adam(self.net.parameters(), lr=lr, betas=(float(betas[0]), float(betas[1])), eps=eps)
        self.eta, self.gamma_averaging = eta, gamma_averaging
        self.n_discrete, self.epsilon_threshold = int(n_discrete), float(epsilon_threshold)
        self.neurd_clip, self.logit_clip, self.grad_clip = float(neurd_clip), float(logit_clip), float(grad_clip)
        self.T_cap = 2 * int(tree_arrays["depth_bound"])
        self.keep, self.last = bool(keep), None
        # chunk_rows: evaluate / differentiate the MLPs on this many (t, b) rows at a time (a 2^20-lane batch is 12.6 M rows: 13 GB per
        # hidden activation otherwise).  Same per-row outputs; the weight gradient is accumulated chunk by chunk (another fp32 order).
        self.chunk_rows = chunk_rows

    def step(self, B, seed, alpha=0.5, lane0=0):
        """One iteration; returns (T, rollout seconds, update seconds)."""
        A = self.A
        t0 = time.perf_counter()
        ro = oracle.rollout(self.tree, self.net.weights(), B, self.T_cap, seed, lane0=lane0)  # episode.py:175-230
        t1 = time.perf_counter()
        T = ro["T"]
        obs = torch.from_numpy(ro["observations"].reshape(T * B, 2 * A * A))
        chunks = None
        if self.chunk_rows and T * B > self.chunk_rows:
            chunks = [(i, min(i + self.chunk_rows, T * B)) for i in range(0, T * B, self.chunk_rows)]

            def whole(net):
                with torch.no_grad():
                    parts = [net.logits(obs[i:j]) for i, j in chunks]
                return torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])

            logit, v = whole(self.net)  # rnad.py:373 (values now; the graph for the backward is rebuilt per chunk below)
            _, v_t = whole(self.net_target)
            lr, _ = whole(self.net_reg)
            lr_, _ = whole(self.net_reg_)
        else:
            logit, v = self.net.logits(obs)  # rnad.py:373
            with torch.no_grad():
                _, v_t = self.net_target.logits(obs)
                lr, _ = self.net_reg.logits(obs)
                lr_, _ = self.net_reg_.logits(obs)
        masks = ro["masks"].reshape(T * B, A)
        pi, log_pi = oracle.policy_head(logit.detach().numpy(), masks)
        _, log_r = oracle.policy_head(lr.numpy(), masks)
        _, log_r_ = oracle.policy_head(lr_.numpy(), masks)
        pip = oracle.process_policy(pi, masks, self.n_discrete, self.epsilon_threshold)  # rnad.py:374
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
        _, dl = oracle.loss_nerd(logit.detach().numpy(), pip, qs[0], qs[1], valid, turns, masks, self.neurd_clip, self.logit_clip)  # :412-422
        if chunks is not None:
            dl_t, dv_t = torch.from_numpy(dl).view_as(logit), torch.from_numpy(dv).view_as(v)
            for i, j in chunks:  # :425, chunk by chunk (parameter .grad accumulates)
                lg_c, v_c = self.net.logits(obs[i:j])
                torch.autograd.backward([lg_c, v_c], [dl_t[i:j], dv_t[i:j]])
        else:
            torch.autograd.backward([logit, v], [torch.from_numpy(dl).view_as(logit), torch.from_numpy(dv).view_as(v)])  # :425
        torch.nn.utils.clip_grad_norm_(self.net.parameters(), self.grad_clip)  # :456
        if self.keep:
            self.last = dict(rollout=ro, grads={k: p.grad.detach().clone() for k, p in self.net.named_parameters()})
        self.opt.step()
        self.opt.zero_grad()
        with torch.no_grad():  # EMA target, rnad.py:516-523
            for p_t, p_n in zip(self.net_target.parameters(), self.net.parameters()):
                p_t.copy_(self.gamma_averaging * p_n + (1 - self.gamma_averaging) * p_t)
        t2 = time.perf_counter()
        return T, t1 - t0, t2 - t1
