/*
 * rnad_rng.h -- the seeded-noise contract of the rollout (public, header-only).
 *
 * The reference draws from torch's GLOBAL generator (reference nn/net.py:49,
 * environment/episode.py:118): `torch.multinomial(p, 1)` which on CPU is `argmax(p / q)`,
 * q ~ Exp(1).  A global sequential generator cannot be reproduced by a data-parallel kernel, so
 * the contract here is: the SAMPLER is exactly the reference's (first-max argmax of p / q), and
 * the NOISE q is either handed in explicitly (tests replay the q the reference consumed) or, in
 * seeded mode, is this counter-based function of (seed, global lane, step, stream, slot):
 *
 *     x = philox4x32-10(counter = {lane_lo, lane_hi, step | stream << 24, slot / 4},
 *                       key     = {seed_lo, seed_hi})[slot % 4]
 *     u = (2 * (x >> 9) + 1) * 2^-24            in (0, 1), exact in fp32
 *     q = -ln(u)                                by the fmaf-only polynomial below
 *
 * stream 0 = action draw (slot = action id), stream 1 = chance draw (slot = chance outcome).
 * `lane` is the GLOBAL episode number (rank offset + local lane), so an N-GPU sharded rollout
 * consumes exactly the noise a 1-GPU rollout would.
 *
 * -ln(u) uses only integer ops and fmaf, which are correctly rounded on the host and on gfx950,
 * so the host (oracle, gcc) and the device (HIP kernels, hipcc) produce the same bits.  Compile
 * with -ffp-contract=off on both sides.  Absolute error vs libm: < 4e-8 (fp32 rounding level).
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

/* n Exp(1) variates for (seed, lane, step, stream), slots 0..n-1. */
RNAD_HD void rnad_exp_noise(uint64_t seed, uint64_t lane, uint32_t step, uint32_t stream, int n, float *out) {
    for (int j = 0; j < n; j += 4) {
        uint32_t c[4] = {(uint32_t)lane, (uint32_t)(lane >> 32), step | (stream << 24), (uint32_t)(j >> 2)};
        rnad_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        for (int i = 0; i < 4 && j + i < n; ++i) out[j + i] = rnad_neg_log_u(c[i]);
    }
}

#endif /* RNAD_RNG_H */
