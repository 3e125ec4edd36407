// rollout_math.hpp -- per-lane device arithmetic of the rollout, shared by rollout.hip and bucket.hip (gfx950).
//
// nn/net.py:45-49 (masked exp-normalise policy head, multinomial as an Exp(1) race) and environment/episode.py:106-121 (the
// chance draw and transition of States.step) for one lane.  Citations are baskuit/R-NaD file:line.
#pragma once

#include "common.hpp"

namespace rnad {
namespace dev {

// nn/net.py:45-46: exp_logits = where(legal, exp(logits), 0); policy = normalize(exp_logits, p=1) (eps 1e-12).
// :76-77: log_policy = where(legal, logits - log(sum(exp_logits)), 0).
template <int A>
__device__ __forceinline__ void policy_head_ptr(const float *logit, uint32_t legal_bits, float *policy, float *log_policy) {
    float ex[A];
    float s = 0.0f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        ex[a] = ((legal_bits >> a) & 1) ? expf(logit[a]) : 0.0f;
        s += fabsf(ex[a]);
    }
    const float d = fmaxf(s, 1e-12f);
#pragma unroll
    for (int a = 0; a < A; ++a) policy[a] = ex[a] / d;
    if (log_policy) {
        float s2 = 0.0f;
#pragma unroll
        for (int a = 0; a < A; ++a) s2 += ex[a];
        const float ls = logf(s2);
#pragma unroll
        for (int a = 0; a < A; ++a) log_policy[a] = ((legal_bits >> a) & 1) ? logit[a] - ls : 0.0f;
    }
}

// torch CPU multinomial(p, 1) == argmax(p / q), q ~ Exp(1), first maximum wins (Distributions.cpp, n_sample == 1).
template <int N>
__device__ __forceinline__ int race_argmax(const float *p, const float *q) {
    int best = 0;
    float bv = p[0] / q[0];
#pragma unroll
    for (int a = 1; a < N; ++a) {
        const float r = p[a] / q[a];
        if (r > bv) {
            bv = r;
            best = a;
        }
    }
    return best;
}

// RN(pa / qa) > RN(pb / qb), the comparison race_argmax makes, decided without the two IEEE divisions (11 VALU instructions each)
// whenever the cross products are not within 2^-20 of each other -- all but ~2^-19 of the draws; the rest divide, so the verdict is
// the reference's in every case.  For noise out of rnad_neg_log_u (q in [2^-24.0, 16.7]) and probabilities (0 <= p <= 2):
//   x = RN(pa qb), y = RN(pb qa).  x > y (1 + 2^-20) and x >= 2^-100 (x normal, so |x - pa qb| <= 2^-24 x; y's error is relative
//   2^-24 or absolute 2^-150) give pa/qa > (pb/qb)(1 + 2^-21); pa/qa = x' / (qa qb) >= 2^-100 / 2^9 is a normal number, so its
//   rounding loses at most 2^-24 of it and RN(pb/qb) gains at most 2^-24 of pb/qb (or 2^-150): strictly greater.
//   y > x (1 + 2^-20) and y >= 2^-100 give pa/qa < pb/qb, and rounding is monotone: not greater.  pa == 0: RN = 0, never greater.
// (tests/test_sampling_math.py replays this on the host over adversarial near-ties.)
__device__ __forceinline__ bool race_beats(float pa, float qa, float pb, float qb) {
#ifdef RNAD_NO_OPT_RACE
    return pa / qa > pb / qb;
#else
    constexpr float kMargin = 0x1p-20f, kFloor = 0x1p-100f;
    const float x = pa * qb, y = pb * qa;
    const bool win = x > fmaxf(fmaf(y, kMargin, y), kFloor);
    const bool lose = y > fmaxf(fmaf(x, kMargin, x), kFloor) || !(pa > 0.0f);
    if (win || lose) return win;  // (never both)
    return pa / qa > pb / qb;
#endif
}

// race_argmax for noise drawn by rnad_exp_noise (the range race_beats assumes): same result, no divisions on the common path.
template <int N>
__device__ __forceinline__ int race_argmax_drawn(const float *p, const float *q) {
    int best = 0;
    float pb = p[0], qb = q[0];
#pragma unroll
    for (int a = 1; a < N; ++a) {
        if (race_beats(p[a], q[a], pb, qb)) {
            pb = p[a];
            qb = q[a];
            best = a;
        }
    }
    return best;
}

// Runtime category count n <= NMAX without runtime-indexed arrays (those would live in scratch): fully
// unrolled, predicated on k < n.
template <int NMAX>
__device__ __forceinline__ void exp_noise_n(uint64_t seed, uint64_t lane, uint32_t step, uint32_t stream, int n,
                                            float (&q)[NMAX]) {
#pragma unroll
    for (int j = 0; j < NMAX; j += 4) {
        if (j < n) {
            uint32_t c[4] = {(uint32_t)lane, (uint32_t)(lane >> 32), step | (stream << 24), (uint32_t)(j >> 2)};
            rnad_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (j + i < NMAX) q[j + i] = rnad_neg_log_u(c[i]);
        }
    }
}

template <int NMAX>
__device__ __forceinline__ void load_n(const float *__restrict__ src, int n, float (&dst)[NMAX]) {
#pragma unroll
    for (int k = 0; k < NMAX; ++k)
        if (k < n) dst[k] = src[k];
}

template <int NMAX>
__device__ __forceinline__ int race_argmax_n(int n, const float (&p)[NMAX], const float (&q)[NMAX]) {
    int best = 0;
    float bv = p[0] / q[0];
#pragma unroll
    for (int a = 1; a < NMAX; ++a) {
        if (a < n) {
            const float r = p[a] / q[a];
            if (r > bv) {
                bv = r;
                best = a;
            }
        }
    }
    return best;
}

// environment/episode.py:106-121 for one lane: the C chance outcomes of joint action (r, c) are 12*C contiguous bytes.
// `chosen` (optional): the index of the sampled outcome, for callers that replay the decision later (transition_apply).
template <int A>
__device__ __forceinline__ void transition_lane(const Trans *__restrict__ trans, int C, int s, int r, int c,
                                                const float *__restrict__ noise_c, uint64_t seed, uint64_t lane, uint32_t step,
                                                int &next, float &reward, int *chosen = nullptr) {
    const Trans *e = trans + (((int64_t)s * A + r) * A + c) * C;
    Trans best = e[0];
    int which = 0;
    if (C > 1) {  // a single outcome wins the race whatever the noise is: no draw needed (same result, the noise is counter-based)
        float q[RNAD_MAX_TRANSITIONS];
        if (noise_c)
            load_n<RNAD_MAX_TRANSITIONS>(noise_c, C, q);
        else
            exp_noise_n<RNAD_MAX_TRANSITIONS>(seed, lane, step, 1u, C, q);
        float bv = best.chance / q[0];
#pragma unroll
        for (int t = 1; t < RNAD_MAX_TRANSITIONS; ++t) {
            if (t < C) {
                const Trans et = e[t];
                const float rr = et.chance / q[t];
                if (rr > bv) {
                    bv = rr;
                    best = et;
                    which = t;
                }
            }
        }
    }
    next = best.next;
    reward = best.value * (next == 0 ? 1.0f : 0.0f);  // rewards *= (indices == 0): keeps -0.0
    if (chosen) *chosen = which;
}

// The transition of a decision taken earlier: outcome `chosen` of joint action (r, c) in state s.
template <int A>
__device__ __forceinline__ void transition_apply(const Trans *__restrict__ trans, int C, int s, int r, int c, int chosen, int &next,
                                                 float &reward) {
    const Trans best = trans[(((int64_t)s * A + r) * A + c) * C + chosen];
    next = best.next;
    reward = best.value * (next == 0 ? 1.0f : 0.0f);
}

}  // namespace dev
}  // namespace rnad
