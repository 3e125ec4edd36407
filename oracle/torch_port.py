"""A pure-PyTorch-CPU transcription of one R-NaD iteration (rollout + update), for bench.py's `cpu_baseline.torch_cpu` leg.

TEST INFRASTRUCTURE ONLY (see oracle/rnad_oracle.c): imported by tests/ and by bench.py's cpu_baseline leg, never by the product.

SURVEY.md section 8(d) asks for the reference's op sequence timed on the host cores in the library the reference itself runs on.  The
reference cannot travel to the GPU box, so this file restates that sequence with torch tensor ops -- own code, same operations in the
same order:
  rollout   environment/episode.py:194-212 (observe -> net.forward -> States.step) with environment/episode.py:62-68,96-125
  update    learn/rnad.py:365-425 (four nets' forward_batch, process_policy, v_trace per player, the two losses, backward),
            learn/vtrace.py:24-55,199-352,356-431, then clip_grad_norm_, Adam, the EMA target (rnad.py:456,514-523)
Pinned by tests/test_oracle_golden.py::test_torch_transcription_gives_the_reference_gradients against the on-policy fixtures
(the reference's own parameter gradients and losses).
"""
import time

import torch


class TorchMLP(torch.nn.Module):
    """nn/net.py:18-35: two heads of Linear(2 A^2, W) -> relu -> Linear(W, .)."""

    def __init__(self, A, width):
        super().__init__()
        self.A = A
        self.value_fc0 = torch.nn.Linear(2 * A * A, width)
        self.value_fc1 = torch.nn.Linear(width, 1)
        self.policy_fc0 = torch.nn.Linear(2 * A * A, width)
        self.policy_fc1 = torch.nn.Linear(width, A)

    def heads(self, obs):
        """logits [N, A], value [N, 1], policy [N, A], log-policy [N, A] under the mover's legal mask (net.py:40-49,70-79)."""
        legal = obs[:, 1, :, 0] != 0
        x = obs.reshape(-1, 2 * self.A * self.A)
        value = self.value_fc1(torch.relu(self.value_fc0(x)))
        logits = self.policy_fc1(torch.relu(self.policy_fc0(x)))
        e = torch.where(legal, torch.exp(logits), torch.zeros_like(logits))
        total = e.sum(-1, keepdim=True)
        policy = e / total.clamp_min(1e-12)
        log_policy = torch.where(legal, logits - torch.log(total), torch.zeros_like(logits))
        return logits, value, policy, log_policy


def observe(tree, idx, player):
    """[B, 2, A, A]: expected values and legal mask from the mover's point of view (episode.py:62-68: the column player sees the negated
    transpose)."""
    ev = tree["expected_value"].index_select(0, idx)[:, 0]
    legal = tree["legal"].index_select(0, idx)[:, 0]
    if player == 1:
        ev, legal = -ev.transpose(1, 2), legal.transpose(1, 2)
    return torch.stack([ev, legal], dim=1)


def play(tree, net, B, T_cap):
    """Episodes.generate: the batch played to the end with `net` as the actor (torch.multinomial draws, as the reference)."""
    A = net.A
    idx = torch.ones((B,), dtype=torch.long)
    rows = torch.arange(B)
    out = {k: [] for k in ("indices", "observations", "policy", "actions", "rewards", "masks")}
    row_action = None
    with torch.no_grad():
        for t in range(T_cap):
            if bool((idx == 0).all()):
                break
            player = t & 1
            obs = observe(tree, idx, player)
            _, _, policy, _ = net.heads(obs)
            action = torch.multinomial(policy, 1).squeeze(-1)
            out["indices"].append(idx.clone())
            out["observations"].append(obs)
            out["policy"].append(policy)
            out["actions"].append(torch.nn.functional.one_hot(action, A).to(torch.float32))
            out["masks"].append(obs[:, 1, :, 0])
            if player == 0:
                row_action = action
                out["rewards"].append(torch.zeros((B,)))
                continue
            chance = tree["chance"].index_select(0, idx)[rows, :, row_action, action]
            nxt = tree["index"].index_select(0, idx)[rows, :, row_action, action]
            val = tree["value"].index_select(0, idx)[rows, :, row_action, action]
            c = torch.multinomial(chance, 1).squeeze(-1)
            idx = nxt[rows, c]
            out["rewards"].append(val[rows, c] * (idx == 0))
    return {k: torch.stack(v) for k, v in out.items()}


def process_policy(pi, mask, n_disc, eps):
    """vtrace.py:24-55: threshold at eps (unless every probability is below it), renormalise, hand out n_disc blocks in descending order."""
    T, B, A = pi.shape
    p, m = pi.reshape(-1, A), mask.reshape(-1, A)
    m = m * ((p >= eps) | (p.max(-1, keepdim=True).values < eps))
    p = m * p / (m * p).sum(-1, keepdim=True)
    blocks = torch.ceil(n_disc * p).to(torch.int32)
    left = torch.full((p.shape[0],), float(n_disc))
    res = torch.zeros_like(p)
    order = torch.argsort(p, descending=True)
    rows = torch.arange(p.shape[0])
    for i in range(A):
        take = torch.minimum(left, blocks[rows, order[:, i]].to(torch.float32))
        left = left - take
        res[rows, order[:, i]] += take
    return (res / n_disc).view(T, B, A)


def v_trace(v, valid, turn, mu, pi, log_pi_reg, a_oh, reward, player, eta, lam, c_bar, rho_bar, gamma):
    """vtrace.py:207-352 for one player: (v_target [T,B,1], has_played [T,B], q [T,B,A]); the backward scan over T with the three-way
    select (our step / the other player's step / an absorbed slot)."""
    T = v.shape[0]
    mine = turn == player
    others = ((2 * mine.to(torch.float32) - 1) * valid).unsqueeze(-1)  # _player_others

    def picked(x):  # probability of the action taken (1 on absorbed slots)
        return (a_oh * x).sum(-1) * valid + (1 - valid)

    ratio = picked(pi) / picked(mu)
    inv_mu = picked(torch.ones_like(pi)) / picked(mu)
    ent = -eta * (pi * log_pi_reg).sum(-1) * others.squeeze(-1)
    elp = -eta * log_pi_reg * others
    zB, oB = torch.zeros_like(reward[0]), torch.ones_like(reward[0])
    c_r, c_ru, c_nv, c_nvt, c_is = zB, zB, torch.zeros_like(v[0]), torch.zeros_like(v[0]), oB
    v_target, q = [None] * T, [None] * T
    has_played = [None] * T
    for t in range(T - 1, -1, -1):
        ru = reward[t] + gamma * c_ru + ent[t]
        dr = reward[t] + gamma * c_r
        w = ratio[t] * c_is
        ours_vt = (v[t] + torch.clamp(w, max=rho_bar).unsqueeze(-1) * (ru.unsqueeze(-1) + gamma * c_nv - v[t])
                   + lam * torch.clamp(w, max=c_bar).unsqueeze(-1) * gamma * (c_nvt - c_nv))
        ours_q = v[t] + elp[t] + a_oh[t] * inv_mu[t].unsqueeze(-1) * (dr.unsqueeze(-1) + gamma * c_is.unsqueeze(-1) * c_nvt - v[t])
        live, me = valid[t] != 0, mine[t]
        sel = lambda ours, opp, reset: torch.where(live, torch.where(me, ours, opp), reset)  # noqa: E731
        sel1 = lambda ours, opp, reset: torch.where(live.unsqueeze(-1), torch.where(me.unsqueeze(-1), ours, opp), reset)  # noqa: E731
        v_target[t] = sel1(ours_vt, torch.zeros_like(ours_vt), torch.zeros_like(ours_vt))
        q[t] = sel1(ours_q, torch.zeros_like(ours_q), torch.zeros_like(ours_q))
        c_r, c_ru, c_nv, c_nvt, c_is = (sel(zB, ent[t] + ratio[t] * dr, zB), sel(zB, ru, zB), sel1(v[t], gamma * c_nv, torch.zeros_like(c_nv)),
                                        sel1(ours_vt, gamma * c_nvt, torch.zeros_like(c_nvt)), sel(oB, w, oB))
        # _has_played (vtrace.py:140-175): its carry is never set, so the mask is "a valid step of this player"
        has_played[t] = valid[t] * mine[t].to(torch.float32)
    return torch.stack(v_target), torch.stack(has_played), torch.stack(q)


def learn_losses(nets, ep, alpha, hp):
    """rnad.py:365-424: (loss, loss_v, loss_nerd) with autograd through the learner net only."""
    net, target, reg, reg_ = nets
    T, B = ep["indices"].shape
    A = net.A
    obs = ep["observations"].reshape(T * B, 2, A, A)
    valid = (ep["indices"] != 0).to(torch.float32)
    turn = (torch.arange(T) % 2).view(T, 1).expand(T, B)
    masks = ep["masks"]
    logit, v, pi, log_pi = (x.view(T, B, -1) for x in net.heads(obs))
    with torch.no_grad():
        pip = process_policy(pi, masks, hp["n_disc"], hp["eps"])
        v_tgt = target.heads(obs)[1].view(T, B, 1)
        log_r = reg.heads(obs)[3].view(T, B, A)
        log_r_ = reg_.heads(obs)[3].view(T, B, A)
        lpol = log_pi - (alpha * log_r + (1 - alpha) * log_r_)
        parts = []
        for player in range(2):
            rew = ep["rewards"] if player == 0 else -ep["rewards"]
            parts.append(v_trace(v_tgt, valid, turn, ep["policy"], pip, lpol, ep["actions"], rew, player, hp["eta"], 1.0, hp["c"], hp["rho"],
                                 hp["gamma"]))
    loss_v = 0
    for vt, played, _ in parts:  # get_loss_v
        n = played.sum()
        loss_v = loss_v + (played.unsqueeze(-1) * (v - vt) ** 2).sum() / (n + (n == 0))
    loss_n = 0
    for player, (_, _, q) in enumerate(parts):  # get_loss_nerd
        adv = torch.clamp(q - (pip * q).sum(-1, keepdim=True), -hp["clip"], hp["clip"])
        centred = logit - (logit * masks).mean(-1, keepdim=True)
        force = (centred > -hp["threshold"]) * torch.clamp(adv, max=0.0) + (centred < hp["threshold"]) * torch.clamp(adv, min=0.0)
        per_slot = (masks * centred * force.detach()).sum(-1)
        m = valid * (turn == player)
        n = m.sum()
        loss_n = loss_n - (per_slot * m).sum() / (n + (n == 0))
    return hp["w_v"] * loss_v + hp["w_n"] * loss_n, loss_v, loss_n


HP = dict(eta=0.2, c=1.0, rho=1.0, gamma=1.0, clip=10_000.0, threshold=2.0, n_disc=32, eps=0.03, w_v=1.0, w_n=1.0)


class TorchCpuTrainer:
    """One process, PyTorch on the host cores: what the reference is when no GPU is present."""

    def __init__(self, tree_arrays, width=256, lr=5e-5, eta=0.2, gamma_averaging=0.001, seed=0):
        torch.manual_seed(seed)
        self.tree = {k: torch.as_tensor(tree_arrays[k]) for k in ("index", "value", "chance", "expected_value", "legal")}
        self.tree["index"] = self.tree["index"].long()
        A = self.tree["index"].shape[-1]
        self.nets = [TorchMLP(A, width) for _ in range(4)]  # learner, target, reg, reg_ (rnad.py:226-231: all start equal)
        for n in self.nets[1:]:
            n.load_state_dict(self.nets[0].state_dict())
        self.opt = torch.optim.Adam(self.nets[0].parameters(), lr=lr, betas=(0.0, 0.999), eps=1e-8)
        self.hp = dict(HP, eta=eta)
        self.gamma_averaging = gamma_averaging
        self.T_cap = 2 * int(tree_arrays["depth_bound"])

    def step(self, B, seed, alpha=0.5):
        """One iteration of rnad.py:495-526; returns (T, rollout seconds, update seconds)."""
        torch.manual_seed(seed)
        t0 = time.perf_counter()
        ep = play(self.tree, self.nets[0], B, self.T_cap)
        t1 = time.perf_counter()
        loss, _, _ = learn_losses(self.nets, ep, alpha, self.hp)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(self.nets[0].parameters(), 10_000.0)
        self.opt.step()
        self.opt.zero_grad()
        with torch.no_grad():
            for p_t, p_n in zip(self.nets[1].parameters(), self.nets[0].parameters()):
                p_t.copy_(self.gamma_averaging * p_n + (1 - self.gamma_averaging) * p_t)
        return ep["indices"].shape[0], t1 - t0, time.perf_counter() - t1
