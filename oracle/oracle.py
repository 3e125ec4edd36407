"""ctypes loader for the CPU oracle (oracle/rnad_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg.  The product package (r-nad_amd/) must never import this module.

All wrappers take and return numpy arrays in the REFERENCE's layouts.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.realpath(__file__))
_SO = os.path.join(_HERE, "_build", "librnad_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "rnad_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "rnad_rng.h")
    stale = not os.path.exists(_SO) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_SO) for p in (src, hdr)
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.oracle_loss_v.restype = C.c_double
        _lib.oracle_loss_nerd.restype = C.c_double
        _lib.oracle_neg_log_u.restype = C.c_float
        _lib.oracle_uniform.restype = C.c_float
        _lib.oracle_rollout.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def observe(ev, legal, idx, player):
    ev, legal, idx, player = _f32(ev), _f32(legal), _i64(idx), _i64(player)
    A = ev.shape[-1]
    B = idx.shape[0]
    obs = np.empty((B, 2, A, A), np.float32)
    mask = np.empty((B, A), np.float32)
    lib().oracle_observe(C.c_int64(B), A, _p(ev), _p(legal), _p(idx), _p(player), _p(obs), _p(mask))
    return obs, mask


def sample(policy, noise):
    policy, noise = _f32(policy), _f32(noise)
    B, A = policy.shape
    out = np.empty((B,), np.int64)
    lib().oracle_sample(C.c_int64(B), A, _p(policy), _p(noise), _p(out))
    return out


def noise(B, n, seed, lane0, t, stream):
    out = np.empty((B, n), np.float32)
    lib().oracle_noise(C.c_int64(B), n, C.c_uint64(seed), C.c_int64(lane0), t, stream, _p(out))
    return out


def uniforms(B, seed, lane0, t):
    """[B, 3]: the seeded uniforms of lanes lane0 .. lane0 + B - 1 for the game transition of env step t (include/rnad_rng.h):
    column 0 the row player's action draw, 1 the column player's, 2 the chance draw."""
    out = np.empty((B, 3), np.float32)
    lib().oracle_uniforms(C.c_int64(B), C.c_uint64(seed), C.c_int64(lane0), t, _p(out))
    return out


def action_uniform(B, seed, lane0, t):
    """[B]: the uniform that decides the action of env step t."""
    return np.ascontiguousarray(uniforms(B, seed, lane0, t)[:, t & 1])


def chance_uniform(B, seed, lane0, t):
    """[B]: the uniform that decides the chance outcome of the transition env step t belongs to."""
    return np.ascontiguousarray(uniforms(B, seed, lane0, t)[:, 2])


def pick(policy, u):
    """The seeded draw: inverse CDF of u over the rows of policy."""
    policy, u = _f32(policy), _f32(u)
    B, A = policy.shape
    assert u.shape == (B,)
    out = np.empty((B,), np.int64)
    lib().oracle_pick(C.c_int64(B), A, _p(policy), _p(u), _p(out))
    return out


def transition(index, chance, value, idx, row_a, col_a, noise_c):
    """noise_c [B, C]: explicit Exp(1) noise (torch's race); [B]: the seeded chance uniform (inverse CDF)."""
    index, chance, value = _i64(index), _f32(chance), _f32(value)
    idx, row_a, col_a, noise_c = _i64(idx), _i64(row_a), _i64(col_a), _f32(noise_c)
    _, Cc, A, _ = index.shape
    B = idx.shape[0]
    out = np.empty((B,), np.int64)
    rew = np.empty((B,), np.float32)
    fn = lib().oracle_transition_pick if noise_c.ndim == 1 else lib().oracle_transition
    assert noise_c.shape == ((B,) if noise_c.ndim == 1 else (B, Cc))
    fn(C.c_int64(B), A, Cc, _p(index), _p(chance), _p(value), _p(idx), _p(row_a), _p(col_a), _p(noise_c), _p(out), _p(rew))
    return out, rew


def policy_head(logits, mask, want_log=True):
    logits, mask = _f32(logits), _f32(mask)
    A = logits.shape[-1]
    N = logits.size // A
    pol = np.empty_like(logits)
    logp = np.empty_like(logits) if want_log else None
    lib().oracle_policy_head(C.c_int64(N), A, _p(logits), _p(mask), _p(pol), _p(logp))
    return pol, logp


MLP_KEYS = ("value_fc0.weight", "value_fc0.bias", "value_fc1.weight", "value_fc1.bias",
            "policy_fc0.weight", "policy_fc0.bias", "policy_fc1.weight", "policy_fc1.bias")


def mlp_forward(weights, x, A):
    """weights: the 8 arrays in MLP_KEYS order; x [N, 2*A*A]."""
    w = [_f32(a) for a in weights]
    x = _f32(x).reshape(-1, 2 * A * A)
    N = x.shape[0]
    W = w[0].shape[0]
    logits = np.empty((N, A), np.float32)
    value = np.empty((N,), np.float32)
    lib().oracle_mlp_forward(C.c_int64(N), A, W, *[_p(a) for a in w], _p(x), _p(logits), _p(value))
    return logits, value


def process_policy(policy, mask, n_disc, eps):
    policy, mask = _f32(policy), _f32(mask)
    A = policy.shape[-1]
    out = np.empty_like(policy)
    lib().oracle_process_policy(C.c_int64(policy.size // A), A, _p(policy), _p(mask), int(n_disc), C.c_float(eps), _p(out))
    return out


def vtrace(v, valid, player_id, mu, pi, logpi, a_oh, reward, player, eta, lambda_, c, rho, gamma):
    v, valid, mu, pi, logpi, a_oh, reward = map(_f32, (v, valid, mu, pi, logpi, a_oh, reward))
    player_id = _i64(player_id)
    T, B, A = mu.shape
    vt = np.empty((T, B, 1), np.float32)
    hp = np.empty((T, B), np.int64)
    q = np.empty((T, B, A), np.float32)
    lib().oracle_vtrace(T, C.c_int64(B), A, _p(v), _p(valid), _p(player_id), _p(mu), _p(pi), _p(logpi), _p(a_oh),
                        _p(reward), int(player), C.c_float(eta), C.c_float(lambda_), C.c_float(c), C.c_float(rho),
                        C.c_float(gamma), _p(vt), _p(hp), _p(q))
    return vt, hp, q


def loss_v(v, vt0, vt1, m0, m1, scale=1.0, want_grad=True):
    v, vt0, vt1 = map(_f32, (v, vt0, vt1))
    m0, m1 = _i64(m0), _i64(m1)
    dv = np.zeros_like(v) if want_grad else None
    loss = lib().oracle_loss_v(C.c_int64(v.size), _p(v), _p(vt0), _p(vt1), _p(m0), _p(m1), C.c_float(scale), _p(dv))
    return loss, dv


def loss_nerd(logit, pi, q0, q1, valid, player_id, legal, clip, thr, scale=1.0, want_grad=True):
    logit, pi, q0, q1, valid, legal = map(_f32, (logit, pi, q0, q1, valid, legal))
    player_id = _i64(player_id)
    A = logit.shape[-1]
    dl = np.zeros_like(logit) if want_grad else None
    loss = lib().oracle_loss_nerd(C.c_int64(logit.size // A), A, _p(logit), _p(pi), _p(q0), _p(q1), _p(valid),
                                  _p(player_id), _p(legal), C.c_float(clip), C.c_float(thr), C.c_float(scale), _p(dl))
    return loss, dl


def nashconv(index, value, chance, legal, root_policy, table_policy, state_index=1, reach=1.0):
    index, value, chance, legal = _i64(index), _f32(value), _f32(chance), _f32(legal)
    root_policy, table_policy = _f32(root_policy), _f32(table_policy)
    S, Cc, A, _ = index.shape
    row_best = np.zeros((S,), np.float32)
    col_best = np.zeros((S,), np.float32)
    reach_o = np.zeros((S,), np.float32)
    depth = np.zeros((S,), np.int32)
    lib().oracle_nashconv(A, Cc, _p(index), _p(value), _p(chance), _p(legal), _p(root_policy), _p(table_policy),
                          C.c_int64(state_index), C.c_float(reach), _p(row_best), _p(col_best), _p(reach_o), _p(depth))
    return row_best, col_best, reach_o, depth


def rollout(tree, weights, B, T_cap, seed, lane0=0, want_logits=False):
    """tree: dict with index/value/chance/expected_value/legal in reference layout."""
    index, value, chance = _i64(tree["index"]), _f32(tree["value"]), _f32(tree["chance"])
    ev, legal = _f32(tree["expected_value"]), _f32(tree["legal"])
    _, Cc, A, _ = index.shape
    w = [_f32(a) for a in weights]
    W = w[0].shape[0]
    wp = (C.c_void_p * 8)(*[a.ctypes.data for a in w])
    out = dict(
        indices=np.zeros((T_cap, B), np.int64), observations=np.zeros((T_cap, B, 2, A, A), np.float32),
        masks=np.zeros((T_cap, B, A), np.float32), policy=np.zeros((T_cap, B, A), np.float32),
        actions=np.zeros((T_cap, B), np.int64), rewards=np.zeros((T_cap, B), np.float32),
        values=np.zeros((T_cap, B), np.float32),
    )
    logits = np.zeros((T_cap, B, A), np.float32) if want_logits else None
    T = lib().oracle_rollout(C.c_int64(B), A, Cc, W, T_cap, _p(index), _p(value), _p(chance), _p(ev), _p(legal), wp,
                             C.c_uint64(seed), C.c_int64(lane0), _p(out["indices"]), _p(out["observations"]),
                             _p(out["masks"]), _p(out["policy"]), _p(out["actions"]), _p(out["rewards"]),
                             _p(out["values"]), _p(logits))
    out = {k: v[:T] for k, v in out.items()}
    if want_logits:
        out["logits"] = logits[:T]
    out["T"] = T
    return out


def philox(ctr, k0, k1):
    c = np.array(ctr, dtype=np.uint32)
    lib().oracle_philox(_p(c), C.c_uint32(k0), C.c_uint32(k1))
    return c


def neg_log_u(x):
    return float(lib().oracle_neg_log_u(C.c_uint32(x)))
