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

// The SEEDED draw (include/rnad_rng.h): inverse CDF of this decision's uniform.  Same categorical distribution as the race, from one
// uniform instead of N exponentials.  (The explicit-noise entry points keep torch's race above.)
template <int N>
__device__ __forceinline__ int pick(const float *p, float u) {  // rnad_pick(p, N, u), unrolled so that p stays in registers
    float s = 0.0f;
#pragma unroll
    for (int a = 0; a < N; ++a) s += p[a];
    const float target = u * s;
    float c = 0.0f;
    int k = 0;
#pragma unroll
    for (int a = 0; a < N - 1; ++a) {
        c += p[a];
        k += c <= target ? 1 : 0;
    }
    return k;
}

// Runtime category count n <= NMAX without runtime-indexed arrays (those would live in scratch): fully unrolled, predicated on k < n.
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

// rnad_pick for a runtime category count n <= NMAX, same arithmetic (sums in index order).
template <int NMAX>
__device__ __forceinline__ int pick_n(int n, const float (&p)[NMAX], float u) {
    float s = 0.0f;
#pragma unroll
    for (int a = 0; a < NMAX; ++a)
        if (a < n) s += p[a];
    const float target = u * s;
    float c = 0.0f;
    int k = 0;
#pragma unroll
    for (int a = 0; a < NMAX - 1; ++a) {
        if (a + 1 < n) {
            c += p[a];
            k += c <= target ? 1 : 0;
        }
    }
    return k;
}

// environment/episode.py:106-121 for one lane: the C chance outcomes of joint action (r, c) are 12*C contiguous bytes.
// noise_c: explicit Exp(1) noise for the race, or nullptr: the seeded draw from u_chance (rnad_decision_uniforms(...)[2]).
// `chosen` (optional): the index of the sampled outcome, for callers that replay the decision later (transition_apply).
template <int A>
__device__ __forceinline__ void transition_lane(const Trans *__restrict__ trans, int C, int s, int r, int c,
                                                const float *__restrict__ noise_c, float u_chance, int &next, float &reward,
                                                int *chosen = nullptr) {
    const Trans *e = trans + (((int64_t)s * A + r) * A + c) * C;
    Trans best = e[0];
    int which = 0;
    if (C > 1) {  // a single outcome is drawn whatever the noise is: nothing to compute
        if (noise_c) {
            float q[RNAD_MAX_TRANSITIONS];
            load_n<RNAD_MAX_TRANSITIONS>(noise_c, C, q);
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
        } else {
            float ch[RNAD_MAX_TRANSITIONS];
#pragma unroll
            for (int t = 0; t < RNAD_MAX_TRANSITIONS; ++t) ch[t] = t < C ? e[t].chance : 0.0f;
            which = pick_n<RNAD_MAX_TRANSITIONS>(C, ch, u_chance);
            best = e[which];
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
