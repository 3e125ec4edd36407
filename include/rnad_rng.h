/*
 * rnad_rng.h -- the seeded-draw contract of the rollout (public, header-only).
 *
 * The reference draws from torch's GLOBAL generator (reference nn/net.py:49,
 * environment/episode.py:118): `torch.multinomial(p, 1)`, one categorical sample per lane.  A
 * global sequential generator cannot be reproduced by a data-parallel kernel, so a rollout takes
 * its randomness in one of two ways:
 *
 *   EXPLICIT NOISE  the caller hands in q ~ Exp(1) per category and the sampler is torch's own
 *                   n_sample == 1 algorithm, the first maximum of p / q (Distributions.cpp): with
 *                   the q the reference consumed, the reference's actions (the golden rollouts).
 *   SEEDED          every decision of lane `lane` is a function of (seed, lane, env step): one
 *                   philox4x32-10 call per GAME TRANSITION (the row player's step t, the column
 *                   player's step t + 1 and the chance draw that follows) gives three uniforms
 *
 *     x[0..3] = philox4x32-10(counter = {lane_lo, lane_hi, t_even | 2 << 24, 0},
 *                             key     = {seed_lo, seed_hi})           t_even = t & ~1
 *     u[i] = (2 * (x[i] >> 9) + 1) * 2^-24      in (0, 1), exact in fp32;  i = 0: row player's
 *                                               action, 1: column player's action, 2: chance
 *
 *                   and each is turned into a category by the inverse CDF (rnad_pick): the first
 *                   category whose running sum of p exceeds u * sum(p).  Same distribution as the
 *                   race -- Cat(p / sum p), categories with p == 0 never drawn, probabilities
 *                   realised on the 2^-23 grid of u -- at one generator call per transition and no
 *                   logarithms or divisions, which is what the seeded rollout kernels are bound by.
 *
 * `lane` is the GLOBAL episode number (rank offset + local lane), so an N-GPU sharded rollout
 * plays exactly the episodes a 1-GPU rollout would.
 *
 * Everything here is integer arithmetic and single fp32 operations (convert, multiply, add,
 * compare; fmaf in the logarithm), correctly rounded on the host and on gfx950, so the host
 * (oracle, gcc) and the device (HIP kernels, hipcc) produce the same bits.  Compile with
 * -ffp-contract=off on both sides.
 *
 * rnad_exp_noise (q = -ln(u) by an fmaf-only polynomial, absolute error vs libm < 4e-8) is the
 * reproducible source of Exp(1) noise for callers of the EXPLICIT entry points.
 */
#ifndef RNAD_RNG_H
#define RNAD_RNG_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define RNAD_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define RNAD_HD static inline
#endif

RNAD_HD void rnad_philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

/* q = -ln(u), u = (2*(x>>9)+1) * 2^-24. */
RNAD_HD float rnad_neg_log_u(uint32_t x) {
    const float u = (float)(2u * (x >> 9) + 1u) * 5.9604644775390625e-08f; /* 2^-24, exact */
    uint32_t bits;
    memcpy(&bits, &u, 4);
    int e = (int)(bits >> 23) - 127;
    bits = (bits & 0x007FFFFFu) | 0x3F800000u;
    float m;
    memcpy(&m, &bits, 4); /* m in [1, 2) */
    if (m >= 1.41421354f) {
        m *= 0.5f;
        e += 1;
    }
    const float f = m - 1.0f; /* [-0.2929, 0.4142) */
    float p = -0x1.31335ap-4f;
    p = fmaf(p, f, 0x1.064786p-3f);
    p = fmaf(p, f, -0x1.0fb036p-3f);
    p = fmaf(p, f, 0x1.22cf1p-3f);
    p = fmaf(p, f, -0x1.5423bp-3f);
    p = fmaf(p, f, 0x1.999e86p-3f);
    p = fmaf(p, f, -0x1.000424p-2f);
    p = fmaf(p, f, 0x1.55555ep-2f);
    p = fmaf(p, f, -0x1.fffff8p-2f);
    p = fmaf(p, f, 1.0f);
    const float lnm = p * f;                               /* ln(m) */
    const float r = fmaf(-(float)e, 0.693147182f, -lnm);   /* -(e ln2 + ln m) */
    return r > 1.17549435e-38f ? r : 1.17549435e-38f;
}

/* n Exp(1) variates for (seed, lane, step, stream), slots 0..n-1:
 * philox counter {lane_lo, lane_hi, step | stream << 24, slot / 4}, output slot % 4. */
RNAD_HD void rnad_exp_noise(uint64_t seed, uint64_t lane, uint32_t step, uint32_t stream, int n, float *out) {
    for (int j = 0; j < n; j += 4) {
        uint32_t c[4] = {(uint32_t)lane, (uint32_t)(lane >> 32), step | (stream << 24), (uint32_t)(j >> 2)};
        rnad_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        for (int i = 0; i < 4 && j + i < n; ++i) out[j + i] = rnad_neg_log_u(c[i]);
    }
}

/* u = (2 * (x >> 9) + 1) * 2^-24: one of the 2^23 odd multiples of 2^-24, never 0 or 1. */
RNAD_HD float rnad_uniform(uint32_t x) { return (float)(2u * (x >> 9) + 1u) * 5.9604644775390625e-08f; }

/* The three uniforms of the game transition that env step t belongs to (t or t ^ 1 give the same):
 * u[0] the row player's action draw, u[1] the column player's, u[2] the chance draw. */
RNAD_HD void rnad_decision_uniforms(uint64_t seed, uint64_t lane, uint32_t t, float u[3]) {
    uint32_t c[4] = {(uint32_t)lane, (uint32_t)(lane >> 32), (t & ~1u) | (2u << 24), 0u};
    rnad_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    u[0] = rnad_uniform(c[0]);
    u[1] = rnad_uniform(c[1]);
    u[2] = rnad_uniform(c[2]);
}

/* Inverse-CDF draw from the n weights p >= 0: the number of running sums c_a = p_0 + .. + p_a
 * (fp32, in index order) that are <= u * c_{n-1}, i.e. the first category whose running sum
 * exceeds the target.  u <= 1 - 2^-24 makes RN(u s) < s for every normal s (s 2^-24 is at least
 * half an ulp of s, and s - ulp/2 is representable when it is exactly half), so the count stops
 * at n - 1 at the latest, and the category it stops at has c_k > c_{k-1}: p_k > 0. */
RNAD_HD int rnad_pick(const float *p, int n, float u) {
    float s = 0.0f;
    for (int a = 0; a < n; ++a) s += p[a];
    const float target = u * s;
    float c = 0.0f;
    int k = 0;
    for (int a = 0; a + 1 < n; ++a) {
        c += p[a];
        k += c <= target ? 1 : 0;
    }
    return k;
}

#endif /* RNAD_RNG_H */
