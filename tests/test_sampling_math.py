"""race_beats (r-nad_amd/csrc/rollout_math.hpp): the sampler decides RN(pa/qa) > RN(pb/qb) from the cross products whenever they are
not within 2^-20 of each other and divides otherwise.  Every operation it uses is an IEEE one (fp32 multiply, fused multiply-add, max,
compare), so the rule is replayed here in numpy on adversarial inputs -- near-ties a few ulps either side of equality, zeros, tiny
probabilities, the extreme noise values -- and its fast verdicts are checked against the two fp32 divisions the reference's
multinomial race makes (torch CPU multinomial == argmax(p / q), first maximum wins).  The kernels themselves are compared with the
oracle's rollout, which divides, in the -m gpu tests."""
import numpy as np

F = np.float32
MARGIN, FLOOR = F(2.0 ** -20), F(2.0 ** -100)
Q_MIN, Q_MAX = 5.9604641e-08, 16.635532  # rnad_neg_log_u at u = 1 - 2^-24 and u = 2^-24


def _fma32(a, b, c):
    """fp32 fused multiply-add of fp32 arrays: a * b + c is exact in double when b is a power of two."""
    return (a.astype(np.float64) * np.float64(b) + c.astype(np.float64)).astype(F)


def _fast_verdict(pa, qa, pb, qb):
    """(decided, win) as race_beats computes them before it falls back to the divisions."""
    x, y = pa * qb, pb * qa
    win = x > np.maximum(_fma32(y, MARGIN, y), FLOOR)
    lose = (y > np.maximum(_fma32(x, MARGIN, x), FLOOR)) | ~(pa > 0)
    assert not (win & lose).any()
    return win | lose, win


def _check(pa, qa, pb, qb):
    pa, qa, pb, qb = (np.asarray(v, F) for v in (pa, qa, pb, qb))
    with np.errstate(under="ignore"):
        decided, win = _fast_verdict(pa, qa, pb, qb)
        want = (pa / qa) > (pb / qb)
    bad = decided & (win != want)
    assert not bad.any(), (pa[bad][:4], qa[bad][:4], pb[bad][:4], qb[bad][:4])
    return decided.mean()


def _noise(rng, n):
    """q as rnad_neg_log_u produces it, with the extremes over-represented."""
    u = (2.0 * rng.integers(0, 1 << 23, n) + 1.0) * 2.0 ** -24
    q = (-np.log(u)).astype(F)
    edge = rng.random(n)
    q = np.where(edge < 0.05, F(Q_MIN), np.where(edge > 0.95, F(Q_MAX), q))
    return np.clip(q, F(Q_MIN), F(Q_MAX)).astype(F)


def test_fast_verdicts_on_random_draws_are_the_divisions_and_almost_always_decide():
    rng = np.random.default_rng(1)
    n = 4_000_000
    pa, pb = rng.random(n).astype(F), rng.random(n).astype(F)
    frac = _check(pa, _noise(rng, n), pb, _noise(rng, n))
    assert frac > 0.9999


def test_fast_verdicts_on_near_ties():
    """pa chosen so that pa/qa sits within a few ulps of pb/qb, on both sides and exactly on it."""
    rng = np.random.default_rng(2)
    n = 2_000_000
    for scale in (1.0, 1e-6, 1e-12, 2.0 ** -58, 2.0 ** -70, 1e-30, 1e-37):
        pb = (rng.random(n) * scale).astype(F)
        qa, qb = _noise(rng, n), _noise(rng, n)
        tie = (pb.astype(np.float64) * qa.astype(np.float64) / qb.astype(np.float64)).astype(F)
        for span in (4, 64, 1 << 10, 1 << 14):  # the last two straddle the 2^-20 margin itself
            k = rng.integers(-span, span + 1, n).astype(np.int32)
            with np.errstate(over="ignore", invalid="ignore"):
                pa = (tie.view(np.int32) + k).view(F)
            pa = np.where(np.isfinite(pa) & (pa >= 0), pa, tie).astype(F)
            _check(pa, qa, pb, qb)


def test_fast_verdicts_with_zero_and_equal_probabilities():
    rng = np.random.default_rng(3)
    n = 500_000
    qa, qb = _noise(rng, n), _noise(rng, n)
    z, p = np.zeros(n, F), rng.random(n).astype(F)
    assert _check(z, qa, p, qb) == 1.0   # an illegal action never beats anything, without dividing
    assert _check(z, qa, z, qb) == 1.0
    assert _check(p, qa, z, qb) > 0.999  # a legal one beats an illegal incumbent (unless p is denormal-small)
    _check(p, qa, p, qa)                 # the same pair twice: a tie, first maximum wins -> not greater
    tiny = (rng.random(n) * 1e-44).astype(F)
    _check(tiny, qa, z, qb)
    _check(tiny, qa, tiny[::-1].copy(), qb)
    _check(np.full(n, 2.0, F), np.full(n, Q_MIN, F), p, qb)  # the largest quotient the race can see
