"""Known-answer and distribution tests for the seeded-draw contract (include/rnad_rng.h), via the oracle build."""
import numpy as np

from oracle import oracle
from tests import _numpy_rng as nprng

# Random123 kat_vectors for philox4x32-10: (counter, key) -> output
KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF,) * 4, (0xFFFFFFFF, 0xFFFFFFFF), (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
     (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


def test_philox_known_answers():
    for ctr, key, want in KAT:
        got = oracle.philox(ctr, *key)
        assert tuple(int(x) for x in got) == want


def test_neg_log_matches_libm():
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.integers(0, 2**32, 20000, dtype=np.uint64), [0, 1, 2**32 - 1, 2**31, 2**9, 2**9 - 1]])
    for x in xs:
        u = (2 * (int(x) >> 9) + 1) * 2.0**-24
        got = oracle.neg_log_u(int(x))
        assert got > 0
        assert abs(got - (-np.log(u))) < 1e-7 + 2e-7 * abs(np.log(u))


def test_noise_is_exponential_and_lane_addressed():
    a = oracle.noise(100000, 3, seed=7, lane0=0, t=4, stream=0)
    assert abs(a.mean() - 1.0) < 0.01 and abs(a.var() - 1.0) < 0.03
    # sharding invariance: lanes [1000, 1100) drawn with an offset equal the slice of the full draw
    b = oracle.noise(100, 3, seed=7, lane0=1000, t=4, stream=0)
    np.testing.assert_array_equal(a[1000:1100], b)
    # different step / stream / seed give different noise
    for kw in (dict(t=5, stream=0, seed=7), dict(t=4, stream=1, seed=7), dict(t=4, stream=0, seed=8)):
        assert not np.array_equal(oracle.noise(100, 3, lane0=0, **kw), a[:100])
    # 5 slots = two philox blocks
    c = oracle.noise(16, 5, seed=7, lane0=0, t=4, stream=0)
    np.testing.assert_array_equal(c[:, :3], a[:16])


def test_decision_uniforms_are_the_documented_philox_words():
    """u[i] = (2 (x[i] >> 9) + 1) 2^-24 of philox(counter {lane_lo, lane_hi, t_even | 2 << 24, 0}, key {seed_lo, seed_hi})."""
    seed, lane0 = 2**40 + 17, 2**33 + 5
    for t in (0, 1, 6, 7, 31):
        u = oracle.uniforms(4, seed, lane0, t)
        for b in range(4):
            lane = lane0 + b
            x = oracle.philox((lane & 0xFFFFFFFF, lane >> 32, (t & ~1) | (2 << 24), 0), seed & 0xFFFFFFFF, seed >> 32)
            want = [np.float32((2 * (int(w) >> 9) + 1) * 2.0**-24) for w in x[:3]]
            assert [float(v) for v in u[b]] == [float(v) for v in want]
    # both steps of a transition read the same block; lanes are addressed globally (sharding invariance)
    np.testing.assert_array_equal(oracle.uniforms(8, 3, 100, 4), oracle.uniforms(8, 3, 100, 5))
    np.testing.assert_array_equal(oracle.uniforms(200, 3, 0, 4)[100:108], oracle.uniforms(8, 3, 100, 4))
    assert not np.array_equal(oracle.uniforms(8, 3, 100, 4), oracle.uniforms(8, 3, 100, 6))
    assert not np.array_equal(oracle.uniforms(8, 3, 100, 4), oracle.uniforms(8, 4, 100, 4))
    u = oracle.uniforms(1 << 18, 1, 0, 0)
    assert u.min() > 0 and u.max() < 1 and abs(u.mean() - 0.5) < 2e-3
    assert abs(np.corrcoef(u[:, 0], u[:, 1])[0, 1]) < 0.01 and abs(np.corrcoef(u[:, 1], u[:, 2])[0, 1]) < 0.01


_pick_reference = nprng.pick  # the definition in plain numpy fp32 (tests/_numpy_rng.py)


def test_numpy_restatement_of_the_contract_is_pinned():
    """tests/_numpy_rng.py (what the device draws are compared with, independent of include/rnad_rng.h): philox known answers, and the
    uniforms / picks of the header through the oracle build."""
    for ctr, key, want in KAT:
        got = nprng.philox4x32_10([np.array([c], np.uint64) for c in ctr], key)
        assert tuple(int(x[0]) for x in got) == want
    for seed, lane0, t in ((1, 0, 0), (2**40 + 17, 2**33 + 5, 7), (99, 123456, 31)):
        np.testing.assert_array_equal(nprng.decision_uniforms(1000, seed, lane0, t), oracle.uniforms(1000, seed, lane0, t))
        np.testing.assert_array_equal(nprng.slot_uniform(1000, seed, lane0, t, 0), oracle.action_uniform(1000, seed, lane0, t))
        np.testing.assert_array_equal(nprng.slot_uniform(1000, seed, lane0, t, 1), oracle.chance_uniform(1000, seed, lane0, t))
    rng = np.random.default_rng(11)
    for n in (1, 2, 3, 5, 8):
        p = nprng.adversarial_weights(rng, 40_000, n)
        u = nprng.slot_uniform(len(p), 5, 0, 2, 0)
        got = oracle.pick(p, u)
        np.testing.assert_array_equal(got, nprng.pick(p, u))
        assert (p[np.arange(len(p)), got] > 0).all()


def test_pick_is_the_inverse_cdf_and_never_draws_a_zero_weight():
    rng = np.random.default_rng(4)
    for n in (1, 2, 3, 5, 8):
        B = 200_000
        p = rng.dirichlet(np.ones(n), size=B).astype(np.float32)
        p[rng.random((B, n)) < 0.3] = 0
        p[np.arange(B), rng.integers(0, n, B)] += np.float32(1e-3)  # at least one positive weight, anywhere
        for u in (rng.random(B).astype(np.float32).clip(2.0**-24, 1 - 2.0**-24), np.full(B, 1 - 2.0**-24, np.float32),
                  np.full(B, 2.0**-24, np.float32)):
            got = oracle.pick(p, u)
            np.testing.assert_array_equal(got, _pick_reference(p, u))
            assert (p[np.arange(B), got] > 0).all()
    # unnormalised weights (the chance tensor of a pruned tree is renormalised, a policy row sums to 1 +- ulps): same law
    w = np.array([[0.5, 0.0, 1.5, 2.0]], np.float32)
    got = oracle.pick(np.repeat(w, 1 << 20, 0), oracle.action_uniform(1 << 20, 9, 0, 2))
    freq = np.bincount(got, minlength=4) / float(1 << 20)
    np.testing.assert_allclose(freq, w[0] / w.sum(), atol=2e-3)


def test_seeded_draws_follow_the_policy():
    """Chi-square of the drawn categories against the weights, for both action uniforms and the chance uniform."""
    n = 1 << 20
    p = np.array([0.03, 0.17, 0.0, 0.45, 0.35], np.float32)
    for which, t in ((oracle.action_uniform, 0), (oracle.action_uniform, 1), (oracle.chance_uniform, 1)):
        got = oracle.pick(np.repeat(p[None], n, 0), which(n, 123, 0, t))
        cnt = np.bincount(got, minlength=5).astype(np.float64)
        assert cnt[2] == 0
        live = p > 0
        chi2 = (((cnt - n * p) ** 2)[live] / (n * p[live])).sum()
        assert chi2 < 21.1, chi2  # 3 degrees of freedom: P(chi2 > 21.1) ~ 1e-4
