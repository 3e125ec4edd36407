"""Known-answer tests for the seeded-noise contract (include/rnad_rng.h), via the oracle build."""
import numpy as np

from oracle import oracle

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
