"""The seeded-draw contract written out in plain numpy -- independent of include/rnad_rng.h, which the kernels and
the C oracle both compile: philox4x32-10 (Random123), the uniform of a decision slot, the inverse-CDF pick."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter: 4 arrays of uint32 (same shape), key: 2 ints -> 4 arrays of uint32."""
    c = [np.asarray(x, dtype=np.uint64) & MASK for x in counter]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return [x.astype(np.uint32) for x in c]


def decision_uniforms(n, seed, lane0, t):
    """[n, 3] float32: the uniforms of lanes lane0 .. lane0 + n - 1 for the game transition that env step t belongs to -- row player's
    action, column player's action, chance outcome: u = (2 (x >> 9) + 1) 2^-24 of philox({lane_lo, lane_hi, t_even | 2 << 24, 0}, seed)."""
    lane = np.uint64(lane0) + np.arange(n, dtype=np.uint64)
    zero = np.zeros(n, np.uint64)
    x = philox4x32_10([lane & MASK, lane >> np.uint64(32), zero + np.uint64((t & ~1) | (2 << 24)), zero], (seed & 0xFFFFFFFF, seed >> 32))
    return np.stack([((2 * (w.astype(np.int64) >> 9) + 1) * 2.0**-24).astype(np.float32) for w in x[:3]], axis=1)


def slot_uniform(n, seed, lane0, step, stream_id):
    """The uniform rnad_sample(seed=, step=, stream_id=) consumes: stream 0 = the mover's action at env step `step`, 1 = the chance draw."""
    u = decision_uniforms(n, seed, lane0, step)
    return np.ascontiguousarray(u[:, 2] if stream_id else u[:, step & 1])


def pick(p, u):
    """Inverse CDF in fp32: the number of running sums p_0 + ... + p_a (index order, a < n - 1) that are <= u * sum(p)."""
    p = np.asarray(p, np.float32)
    c = np.zeros(len(p), np.float32)
    k = np.zeros(len(p), np.int64)
    s = np.zeros(len(p), np.float32)
    for a in range(p.shape[1]):
        s = (s + p[:, a]).astype(np.float32)
    target = (np.asarray(u, np.float32) * s).astype(np.float32)
    for a in range(p.shape[1] - 1):
        c = (c + p[:, a]).astype(np.float32)
        k += c <= target
    return k


def adversarial_weights(rng, B, n):
    """Rows with exact zeros anywhere, tiny and huge weights, ties, denormal-adjacent values, all mass on the last category."""
    p = rng.dirichlet(np.ones(n), size=B).astype(np.float32)
    p[rng.random((B, n)) < 0.3] = 0
    p[np.arange(B), rng.integers(0, n, B)] += np.float32(1e-3)  # at least one positive weight, anywhere
    k = B // 8
    p[:k] *= np.float32(1e-30)                       # sums far below 1
    p[k:2 * k] *= np.float32(1e30)                   # and far above
    p[2 * k:3 * k] = np.float32(1.0) / np.float32(n)  # exact ties of the running sums with u * total on the 2^-24 grid
    p[3 * k:4 * k, :-1] = 0
    p[3 * k:4 * k, -1] = 1                            # only the last category is positive
    return p
