#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference (baskuit/R-NaD).

Runs only in the build container (needs /root/reference); the GPU box never runs it and never
sees the reference -- only the .npz data files written here travel.  Nothing from the
reference's source text is stored: fixtures hold inputs and the outputs the reference computed.

How the reference is made importable:
  * the reference tree is copied to a throw-away temp dir at run time (its `RNaD.__init__`
    mkdirs next to its own source, reference learn/rnad.py:156-163, and /root/reference is
    read-only); the copy is deleted on exit,
  * `pygambit` (tree.py:5) and `wandb` (rnad.py:16) are not installed: a Shapley-Snow solver
    (`_pygambit_stub.py`) and an empty module are injected into `sys.modules`,
  * `torch.multinomial` is wrapped by a recorder.  On CPU torch implements
    `multinomial(p, 1)` as `argmax(p / q)` with `q ~ Exp(1)` drawn by `empty_like(p).exponential_(1)`;
    the wrapper draws q the same way from the same generator state, asserts that the result equals
    the UNPATCHED `torch.multinomial` on that state, and records q.  The fixtures are therefore
    the unmodified reference's outputs, with the noise it consumed stored beside them.

Usage:  python tests/golden/make_golden.py      (rewrites tests/golden/*.npz)
"""
import atexit
import json
import os
import random
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.realpath(__file__))
REF_SRC = "/root/reference"

# --------------------------------------------------------------------------- import the reference
_tmp = tempfile.mkdtemp(prefix="rnad_ref_")
atexit.register(shutil.rmtree, _tmp, True)
REF = os.path.join(_tmp, "ref")
shutil.copytree(REF_SRC, REF, ignore=shutil.ignore_patterns("__pycache__", "*.png"))
sys.path.insert(0, REF)
sys.path.insert(0, HERE)
import _pygambit_stub  # noqa: E402

sys.modules["pygambit"] = _pygambit_stub
sys.modules["wandb"] = types.ModuleType("wandb")

import environment.episode as ref_episode  # noqa: E402
import environment.tree as ref_tree  # noqa: E402
import learn.rnad as ref_rnad  # noqa: E402
import learn.vtrace as ref_vtrace  # noqa: E402
import nn.net as ref_net  # noqa: E402
import util.metric as ref_metric  # noqa: E402

torch.set_num_threads(1)

# --------------------------------------------------------------------------- multinomial recorder
_orig_multinomial = torch.multinomial
NOISE = []  # list of np arrays, one per multinomial call, in call order


def _recording_multinomial(p, num_samples=1, replacement=False, *, generator=None, out=None):
    assert num_samples == 1 and generator is None and out is None
    state = torch.get_rng_state()
    want = _orig_multinomial(p, 1)
    torch.set_rng_state(state)
    q = torch.empty_like(p).exponential_(1)
    got = torch.argmax(p / q, dim=-1, keepdim=True)
    assert torch.equal(want, got), "torch CPU multinomial is no longer argmax(p / Exp(1))"
    NOISE.append(q.numpy().copy())
    return got


torch.multinomial = _recording_multinomial


def seed_all(s):
    torch.manual_seed(s)
    np.random.seed(s)
    random.seed(s)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    out = {}
    for k, v in arrays.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(path, **out)
    print(f"wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB)")


def state_dict_np(net, prefix):
    return {prefix + k.replace(".", "_"): v.detach().numpy().copy() for k, v in net.state_dict().items()}


# --------------------------------------------------------------------------- 1. trees
# The lambdas below are re-declared verbatim in tests/test_tree.py (they consume python `random`).
TREE_SPECS = {
    # BASELINE.json configs[0]: depth-3 binary tree, no chance
    "c1": dict(seed=0, kw=dict(max_actions=2, max_transitions=1, depth_bound=3)),
    # reference main.py:31-39 shape: 3x3, 2 chance outcomes, threshold .3, randomly pruned depth <= 4
    "small": dict(
        seed=1,
        kw=dict(max_actions=3, max_transitions=2, transition_threshold=0.3, depth_bound=4),
        depth_lambda="prune",
    ),
    # ragged legality + 3 chance outcomes (threshold < 1/C so a profile can never be emptied)
    "ragged": dict(
        seed=2,
        kw=dict(max_actions=4, max_transitions=3, transition_threshold=0.2, depth_bound=3, row_actions=3, col_actions=4),
        row_lambda="ragged",
        col_lambda="ragged",
    ),
    # 5 actions: exercises A=5 reductions
    "a5": dict(seed=3, kw=dict(max_actions=5, max_transitions=2, transition_threshold=0.1, depth_bound=2)),
}


def _lambdas(spec):
    kw = {}
    if spec.get("depth_lambda") == "prune":
        kw["depth_bound_lambda"] = lambda tree: tree.depth_bound - 1 - 2 * (random.random() < 0.5)
    if spec.get("row_lambda") == "ragged":
        kw["row_actions_lambda"] = lambda tree: random.randint(1, tree.max_actions)
    if spec.get("col_lambda") == "ragged":
        kw["col_actions_lambda"] = lambda tree: random.randint(1, tree.max_actions)
    return kw


def make_tree(name):
    spec = TREE_SPECS[name]
    seed_all(spec["seed"])
    tree = ref_tree.Tree(**spec["kw"], **_lambdas(spec))
    tree.generate()
    tree.assert_index_is_tree()
    meta = dict(seed=spec["seed"], kw=spec["kw"], depth_lambda=spec.get("depth_lambda"),
                row_lambda=spec.get("row_lambda"), col_lambda=spec.get("col_lambda"),
                hash=tree.hash, size=int(tree.value_tensor.shape[0]))
    if name == "c1":
        # the reference's own on-disk format (Tree.save, tree.py:385-415): kept as a data fixture for the loader test
        tree.save("golden_c1")
        shutil.copy(os.path.join(REF, "saved_trees", "golden_c1", "tree.tar"), os.path.join(HERE, "ref_tree_c1.tar"))
    save(
        "tree_" + name,
        index=tree.index_tensor, value=tree.value_tensor, chance=tree.chance_tensor,
        expected_value=tree.expected_value_tensor, legal=tree.legal_tensor,
        root_value=tree.root_value_tensor, solution=tree.solution_tensor,
        meta=json.dumps(meta),
    )
    return tree


# --------------------------------------------------------------------------- 2. rollouts
def make_rollout(name, tree, batch, seed, width=32):
    """Unmodified `Episodes.generate` (episode.py:175-230) with a seeded reference MLP."""
    seed_all(seed)
    net = ref_net.MLP(tree.max_actions, width)
    logits_rec = []
    fwd = net.forward

    def rec_forward(obs):
        out = fwd(obs)
        logits_rec.append(out[0].detach().numpy().copy())
        return out

    net.forward = rec_forward
    NOISE.clear()
    ep = ref_episode.Episodes(tree, batch)
    ep.generate(net)
    net.forward = fwd
    T = ep.t_eff + 1
    A, C = tree.max_actions, tree.max_transitions
    # call order per loop iteration: action draw [B,A]; on col turns additionally chance draw [B,C]
    noise_action = np.zeros((T, batch, A), np.float32)
    noise_chance = np.zeros((T, batch, C), np.float32)  # rows for even t stay unused (zeros)
    it = iter(NOISE)
    for t in range(T):
        noise_action[t] = next(it)
        if t % 2 == 1:
            noise_chance[t] = next(it)
    assert next(it, None) is None
    assert T % 2 == 0
    save(
        "rollout_" + name,
        **state_dict_np(net, "w_"),
        width=width, batch=batch, t_eff=ep.t_eff,
        indices=ep.indices, turns=ep.turns, observations=ep.observations, masks=ep.masks,
        policy=ep.policy, actions=ep.actions, rewards=ep.rewards, values=ep.values,
        logits=np.stack(logits_rec), noise_action=noise_action, noise_chance=noise_chance,
    )
    return ep, net


# --------------------------------------------------------------------------- 3. process_policy
def make_process_policy():
    seed_all(7)
    A = 4
    rows = []
    masks = []

    def add(p, m):
        rows.append(p)
        masks.append(m)

    # random rows with random masks (renormalised onto the mask like the net's policy)
    for _ in range(200):
        m = (np.random.rand(A) < 0.7).astype(np.float32)
        if m.sum() == 0:
            m[np.random.randint(A)] = 1
        p = np.random.dirichlet(np.ones(A) * np.random.choice([0.1, 0.5, 2.0])).astype(np.float32) * m
        p = p / p.sum() if p.sum() > 0 else m / m.sum()
        add(p.astype(np.float32), m)
    one = np.float32(1.0)
    add(np.array([0.25, 0.25, 0.25, 0.25], np.float32), np.ones(A, np.float32))  # exact 4-way tie
    add(np.array([0.5, 0.5, 0, 0], np.float32), np.array([1, 1, 0, 0], np.float32))  # tie, 2 legal
    add(np.array([0, 0, 1, 0], np.float32), np.array([0, 0, 1, 0], np.float32))  # single legal action
    add(np.array([0.97, 0.01, 0.01, 0.01], np.float32), np.ones(A, np.float32))  # three below eps
    add(np.array([0.03, 0.03, 0.47, 0.47], np.float32), np.ones(A, np.float32))  # exactly eps
    add(np.array([1 / 3, 1 / 3, 1 / 3, 0], np.float32), np.array([1, 1, 1, 0], np.float32))
    add(np.array([0.4, 0.3, 0.2, 0.1], np.float32), np.ones(A, np.float32))
    add(np.array([1, 0, 0, 0], np.float32) * one, np.array([1, 0, 0, 0], np.float32))  # absorbing-state row
    n = len(rows)
    T = 2
    while len(rows) % T:
        add(rows[0], masks[0])
    pol = torch.tensor(np.stack(rows)).view(T, -1, A)
    msk = torch.tensor(np.stack(masks)).view(T, -1, A)
    out = {}
    for n_disc, eps in ((32, 0.03), (16, 0.1), (32, 0.6)):  # eps=.6: "all below threshold" branch
        out[f"out_n{n_disc}_e{eps}"] = ref_vtrace.process_policy(pol.clone(), msk.clone(), n_disc, eps)
    save("process_policy", policy=pol, mask=msk, n_rows=n, **out)


# --------------------------------------------------------------------------- 4. learn step pieces
def _bare_rnad(tree, nets, eta, **over):
    """An RNaD object without running __init__ (which touches the filesystem)."""
    o = object.__new__(ref_rnad.RNaD)
    o.tree = tree
    o.net, o.net_target, o.net_reg, o.net_reg_ = nets
    o.eta = eta
    o.n_discrete, o.epsilon_threshold = 32, 0.03
    o.c_bar, o.roh_bar = 1, 1
    o.vtrace_gamma = 1
    o.neurd_clip, o.beta, o.grad_clip = 10**3, 2, 10**3
    o.value_weight, o.neurd_weight = 1, 1
    o.wandb = False
    for k, v in over.items():
        setattr(o, k, v)
    return o


def make_learn(name, tree, ep, eta, alpha, seed, width=32, **over):
    """The reference's own `RNaD.__learn` (rnad.py:353-456) with every intermediate recorded."""
    seed_all(seed)
    nets = [ref_net.MLP(tree.max_actions, width) for _ in range(4)]
    rn = _bare_rnad(tree, nets, eta, **over)
    rec = {}
    orig = dict(pp=ref_vtrace.process_policy, vt=ref_vtrace.v_trace, lv=ref_vtrace.get_loss_v, ln=ref_vtrace.get_loss_nerd)
    vt_calls = []

    def pp(policy, mask, n_disc, eps):
        r = orig["pp"](policy, mask, n_disc, eps)
        rec["pi"] = policy.detach().clone()
        rec["pi_processed"] = r.detach().clone()
        return r

    def vt(v, valid, player_id, mu, pi, logpi, others, a_oh, reward, player, **kw):
        r = orig["vt"](v, valid, player_id, mu, pi, logpi, others, a_oh, reward, player, **kw)
        vt_calls.append(dict(v=v, valid=valid, logpi=logpi, reward=reward, out=r, kw=kw))
        return r

    def lv(*a):
        r = orig["lv"](*a)
        rec["loss_v"] = r.detach().clone()
        return r

    def ln(*a, **kw):
        r = orig["ln"](*a, **kw)
        rec["loss_nerd"] = r.detach().clone()
        return r

    ref_vtrace.process_policy, ref_vtrace.v_trace, ref_vtrace.get_loss_v, ref_vtrace.get_loss_nerd = pp, vt, lv, ln
    fb = nets[0].forward_batch

    def fb_rec(episodes):
        outs = fb(episodes)
        for o in (outs[0], outs[3]):
            o.retain_grad()
        rec["fb"] = outs
        return outs

    nets[0].forward_batch = fb_rec
    try:
        rn._RNaD__learn(ep, alpha)
    finally:
        ref_vtrace.process_policy, ref_vtrace.v_trace = orig["pp"], orig["vt"]
        ref_vtrace.get_loss_v, ref_vtrace.get_loss_nerd = orig["lv"], orig["ln"]
        nets[0].forward_batch = fb
    logit, log_pi, pi, v = rec["fb"]
    arrays = dict(
        eta=eta, alpha=alpha, width=width,
        logit=logit, log_pi=log_pi, pi=pi, v=v,
        pi_processed=rec["pi_processed"],
        v_target_net=vt_calls[0]["v"], valid=vt_calls[0]["valid"], log_policy_reg=vt_calls[0]["logpi"],
        loss_v=rec["loss_v"], loss_nerd=rec["loss_nerd"],
        dlogit=logit.grad, dv=v.grad,
    )
    for p in range(2):
        vt_out, hp, lo = vt_calls[p]["out"]
        arrays[f"v_target_p{p}"] = vt_out
        arrays[f"has_played_p{p}"] = hp
        arrays[f"q_p{p}"] = lo
    for i, tag in enumerate(("net", "target", "reg", "reg_")):
        arrays.update(state_dict_np(nets[i], f"w_{tag}_"))
    for k, p_ in nets[0].named_parameters():
        arrays["g_net_" + k.replace(".", "_")] = p_.grad
    for k, v_ in over.items():
        arrays["hp_" + k] = v_
    save("learn_" + name, **arrays)


# --------------------------------------------------------------------------- 4b. synthetic off-policy v_trace
def make_vtrace_synth():
    """v_trace / losses on hand-made trajectories where pi != mu and rho, c, gamma, lambda != 1."""
    seed_all(11)
    T, B, A = 10, 96, 3
    lengths = np.random.randint(0, T + 1, size=B)
    lengths[:4] = (0, 1, T, T - 1)
    valid = (np.arange(T)[:, None] < lengths[None, :]).astype(np.float32)
    player_id = np.random.randint(0, 2, size=(T, B)).astype(np.int64)  # NOT alternating: general case
    player_id[:, : B // 2] = (np.arange(T) % 2)[:, None]
    mask = (np.random.rand(T, B, A) < 0.8).astype(np.float32)
    mask[..., 0] = 1

    def pol():
        x = np.random.dirichlet(np.ones(A), size=(T, B)).astype(np.float32) * mask
        return (x / x.sum(-1, keepdims=True)).astype(np.float32)

    mu, pi = pol(), pol()
    logpi_reg = (np.random.randn(T, B, A) * 0.3).astype(np.float32) * mask
    act = np.zeros((T, B), np.int64)
    for t in range(T):
        for b in range(B):
            act[t, b] = np.random.choice(A, p=mu[t, b] / mu[t, b].sum())
    a_oh = np.eye(A, dtype=np.float32)[act]
    v = np.random.randn(T, B, 1).astype(np.float32)
    reward = (np.random.randn(T, B) * (np.random.rand(T, B) < 0.3)).astype(np.float32)
    tt = lambda x: torch.tensor(x)  # noqa: E731
    arrays = dict(valid=valid, player_id=player_id, mask=mask, mu=mu, pi=pi, logpi_reg=logpi_reg,
                  actions=act, v=v, reward=reward)
    for tag, hp in (("a", dict(eta=0.2, lambda_=1.0, c=1.0, rho=1.0, gamma=1.0)),
                    ("b", dict(eta=0.5, lambda_=0.9, c=0.8, rho=1.3, gamma=0.95))):
        for p in range(2):
            rew = tt(reward) if p == 0 else -tt(reward)
            others = ref_vtrace._player_others(tt(player_id), tt(valid), p)
            vt, hpd, lo = ref_vtrace.v_trace(tt(v), tt(valid), tt(player_id), tt(mu), tt(pi), tt(logpi_reg),
                                            others, tt(a_oh), rew, p, **hp)
            arrays[f"{tag}_v_target_p{p}"] = vt
            arrays[f"{tag}_has_played_p{p}"] = hpd
            arrays[f"{tag}_q_p{p}"] = lo
        arrays[f"{tag}_hp"] = json.dumps(hp)
    # losses + autograd grads on the 'b' outputs with a non-trivial clip / threshold
    logit = torch.tensor((np.random.randn(T, B, A) * 1.5).astype(np.float32), requires_grad=True)
    vv = torch.tensor(v, requires_grad=True)
    vts = [tt(arrays[f"b_v_target_p{p}"].numpy()) for p in range(2)]
    hps = [arrays[f"b_has_played_p{p}"] for p in range(2)]
    qs = [arrays[f"b_q_p{p}"] for p in range(2)]
    loss_v = ref_vtrace.get_loss_v([vv] * 2, vts, hps)
    isc = [torch.ones(T, B, 1)] * 2
    loss_n = ref_vtrace.get_loss_nerd([logit] * 2, [tt(pi)] * 2, qs, tt(valid), tt(player_id), tt(mask), isc,
                                     clip=1.5, threshold=2.0)
    (loss_v + loss_n).backward()
    arrays.update(logit=logit.detach(), loss_v=loss_v.detach(), loss_nerd=loss_n.detach(), dlogit=logit.grad,
                  dv=vv.grad, nerd_clip=1.5, nerd_threshold=2.0)
    save("vtrace_synth", **arrays)


# --------------------------------------------------------------------------- 6. NashConv
def make_nashconv(name, tree, seed, width=32):
    seed_all(seed)
    net = ref_net.MLP(tree.max_actions, width)
    net.device = torch.device("cpu")
    data = ref_metric.NashConvData(tree)
    data.get_nashconv_from_net(tree, net)
    means = data.mean_nashconv_by_depth()
    arrays = dict(
        **state_dict_np(net, "w_"), width=width,
        joint_policy=data.joint_policy, row_best=data.row_best, col_best=data.col_best,
        reach_probability=data.reach_probability, depth=data.depth,
        nashconv=(data.row_best[1] + data.col_best[1]).item(),
        mean_depths=np.array(sorted(means)), mean_values=np.array([means[k] for k in sorted(means)]),
    )
    # the tree's own solution as the joint policy (the invariant tests/test_nashconv.py wanted to pin;
    # pre-filling data.joint_policy is what get_nashconv_from_net does, metric.py:72-86)
    sol = ref_metric.NashConvData(tree)
    sol.joint_policy = tree.solution_tensor.clone()
    sol.get_nashconv(tree, sol.joint_policy)
    arrays.update(sol_row_best=sol.row_best, sol_col_best=sol.col_best, sol_reach=sol.reach_probability,
                  sol_depth=sol.depth, sol_nashconv=(sol.row_best[1] + sol.col_best[1]).item())
    save("nashconv_" + name, **arrays)


# --------------------------------------------------------------------------- 5/7. a short real RNaD.run
def make_run(tree):
    """`RNaD(...).run()` for 2 regularisation updates x 3 steps on the c1 tree; per-step snapshots."""
    seed_all(21)
    B = 64
    rn = ref_rnad.RNaD(tree=tree, device=torch.device("cpu"), directory_name="golden", batch_size=B, eta=0.2,
                       bounds=[2], delta_m=[3], lr=1e-2, gamma_averaging=0.1, b1_adam=0.0, wandb=False,
                       net_params={"type": "MLP", "max_actions": tree.max_actions, "width": 16})
    steps = []
    learn = rn._RNaD__learn
    gen = ref_episode.Episodes.generate
    samp = ref_episode.Episodes.sample
    cur = {}

    def gen_rec(self, net):
        NOISE.clear()
        cur["w_actor"] = {k: v.detach().numpy().copy() for k, v in net.state_dict().items()}
        gen(self, net)
        T = self.t_eff + 1
        it = iter(NOISE)
        na = np.zeros((T, B, tree.max_actions), np.float32)
        nc = np.zeros((T, B, tree.max_transitions), np.float32)
        for t in range(T):
            na[t] = next(it)
            if t % 2:
                nc[t] = next(it)
        cur.update(noise_action=na, noise_chance=nc, indices=self.indices.numpy().copy(),
                   actions=self.actions.numpy().copy(), rewards=self.rewards.numpy().copy())

    def samp_rec(self, n):
        r = samp(self, n)
        return r

    def learn_rec(episodes, alpha, log=None):
        cur["alpha"] = alpha
        cur["m"], cur["n"] = rn.m, rn.n
        # Episodes.sample permutes lanes (episode.py:246); store the permuted trajectory the loss saw
        cur["sampled_indices"] = episodes.indices.numpy().copy()
        learn(episodes, alpha, log=log)
        cur["grads"] = {k: p.grad.detach().numpy().copy() for k, p in rn.net.named_parameters()}

    ref_episode.Episodes.generate = gen_rec
    ref_episode.Episodes.sample = samp_rec
    rn._RNaD__learn = learn_rec
    opt_step_holder = {}

    # snapshot after the EMA update of every step: wrap load_state_dict of net_target (rnad.py:523)
    def install_snap():
        tgt_load = rn.net_target.load_state_dict

        def snap(sd, *a, **k):
            r = tgt_load(sd, *a, **k)
            if "alpha" in cur:
                st = dict(cur)
                st["net"] = {k2: v.detach().numpy().copy() for k2, v in rn.net.state_dict().items()}
                st["net_target"] = {k2: v.detach().numpy().copy() for k2, v in rn.net_target.state_dict().items()}
                st["net_reg"] = {k2: v.detach().numpy().copy() for k2, v in rn.net_reg.state_dict().items()}
                st["net_reg_"] = {k2: v.detach().numpy().copy() for k2, v in rn.net_reg_.state_dict().items()}
                steps.append(st)
                cur.clear()
            return r

        rn.net_target.load_state_dict = snap

    init = rn._RNaD__initialize

    def init_then_hook():
        init()
        opt_step_holder["w0"] = {k: v.detach().numpy().copy() for k, v in rn.net.state_dict().items()}
        install_snap()

    rn._RNaD__initialize = init_then_hook
    try:
        rn.run(checkpoint_mod=3, expl_mod=10**9, log_mod=10**9)
    finally:
        ref_episode.Episodes.generate = gen
        ref_episode.Episodes.sample = samp
    # the reference's checkpoint files of this run (rnad.py:208-209, :307-319) as data fixtures for the resume test
    run_dir = os.path.join(REF, "saved_runs", "golden")
    shutil.copy(os.path.join(run_dir, "params"), os.path.join(HERE, "ref_run_params"))
    shutil.copy(os.path.join(run_dir, "1", "0"), os.path.join(HERE, "ref_run_ckpt_1_0"))
    final = {f"final_{tag}_{k.replace('.', '_')}": v.detach().numpy().copy()
             for tag, net in (("net", rn.net), ("target", rn.net_target), ("reg", rn.net_reg), ("reg_", rn.net_reg_))
             for k, v in net.state_dict().items()}
    arrays = dict(n_steps=len(steps), batch=B, eta=0.2, lr=1e-2, gamma_averaging=0.1, width=16,
                  bounds=np.array([2]), delta_m=np.array([3]), **final)
    for k, v in opt_step_holder["w0"].items():
        arrays["w0_" + k.replace(".", "_")] = v
    for i, st in enumerate(steps):
        arrays[f"s{i}_alpha"] = st["alpha"]
        arrays[f"s{i}_mn"] = np.array([st["m"], st["n"]])
        for key in ("noise_action", "noise_chance", "indices", "actions", "rewards", "sampled_indices"):
            arrays[f"s{i}_{key}"] = st[key]
        for tag in ("w_actor", "grads", "net", "net_target", "net_reg", "net_reg_"):
            for k, v in st[tag].items():
                arrays[f"s{i}_{tag}_{k.replace('.', '_')}"] = v
    save("run_c1", **arrays)


def main():
    trees = {name: make_tree(name) for name in TREE_SPECS}
    eps = {}
    eps["c1"], _ = make_rollout("c1", trees["c1"], batch=128, seed=100)
    eps["small"], _ = make_rollout("small", trees["small"], batch=256, seed=101)
    eps["ragged"], _ = make_rollout("ragged", trees["ragged"], batch=192, seed=102)
    eps["a5"], _ = make_rollout("a5", trees["a5"], batch=64, seed=103)
    make_process_policy()
    make_learn("c1_eta0.2", trees["c1"], eps["c1"], eta=0.2, alpha=0.35, seed=200)
    make_learn("small_eta0", trees["small"], eps["small"], eta=0.0, alpha=1.0, seed=201)
    make_learn("small_eta0.2", trees["small"], eps["small"], eta=0.2, alpha=0.6, seed=202)
    make_learn("ragged_eta0.5", trees["ragged"], eps["ragged"], eta=0.5, alpha=0.0, seed=203,
               c_bar=0.9, roh_bar=1.2, vtrace_gamma=0.97, beta=1.0, neurd_clip=0.7)
    make_learn("a5_eta0.2", trees["a5"], eps["a5"], eta=0.2, alpha=0.5, seed=204)
    make_vtrace_synth()
    for name in TREE_SPECS:
        make_nashconv(name, trees[name], seed=300)
    make_run(trees["c1"])


if __name__ == "__main__":
    main()
