// learn_math.hpp -- per-row device arithmetic of the R-NaD update, shared by learn.hip and bucket.hip (gfx950).
//
// One function per reference function, in the reference's fp32 operation order (files including this are built with
// -ffp-contract=off): nn/net.py:45-46,76-77 (policy head), learn/vtrace.py:24-55 (process_policy), :249-333 (one step of
// _loop_v_trace), :410-429 (get_loss_nerd row with its closed-form gradient).  Citations are baskuit/R-NaD file:line.
#pragma once

#include "common.hpp"

namespace rnad {
namespace dev {

// nn/net.py:45-46,76-77 (same function as in rollout.hip, kept local so each kernel file stands alone)
template <int A>
__device__ __forceinline__ void policy_head(const float (&logit)[A], uint32_t legal_bits, float (&policy)[A], float (&log_policy)[A]) {
    float ex[A];
    float s = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        ex[a] = ((legal_bits >> a) & 1) ? expf(logit[a]) : 0.0f;
        s += fabsf(ex[a]);
        s2 += ex[a];
    }
    const float d = fmaxf(s, 1e-12f);
    const float ls = logf(s2);
#pragma unroll
    for (int a = 0; a < A; ++a) {
        policy[a] = ex[a] / d;
        log_policy[a] = ((legal_bits >> a) & 1) ? logit[a] - ls : 0.0f;
    }
}

template <int A>
__device__ __forceinline__ void log_policy_only(const float (&logit)[A], uint32_t legal_bits, float (&log_policy)[A]) {
    float s2 = 0.0f;
#pragma unroll
    for (int a = 0; a < A; ++a) s2 += ((legal_bits >> a) & 1) ? expf(logit[a]) : 0.0f;
    const float ls = logf(s2);
#pragma unroll
    for (int a = 0; a < A; ++a) log_policy[a] = ((legal_bits >> a) & 1) ? logit[a] - ls : 0.0f;
}

// learn/vtrace.py:24-55 for one row.  `mask` holds the caller's mask VALUES (0/1 in practice).
template <int A>
__device__ __forceinline__ void process_policy_row(const float (&pi)[A], const float (&mask)[A], int n_disc, float eps,
                                                   float (&out)[A]) {
    float mx = pi[0];
#pragma unroll
    for (int a = 1; a < A; ++a) mx = pi[a] > mx ? pi[a] : mx;
    const bool all_below = mx < eps;  // :37 "prevent degen case where all < eps"
    float m[A], p[A];
    float s = 0.0f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        m[a] = mask[a] * ((pi[a] >= eps) || all_below ? 1.0f : 0.0f);  // :34-39
        s += m[a] * pi[a];
    }
#pragma unroll
    for (int a = 0; a < A; ++a) {
        p[a] = m[a] * pi[a] / s;  // :40
        out[a] = 0.0f;
    }
    // argsort(descending) with ties in index order == rank by (#greater) + (#equal with lower index)  (:46)
    int rank[A];
#pragma unroll
    for (int a = 0; a < A; ++a) {
        int r = 0;
#pragma unroll
        for (int j = 0; j < A; ++j) r += (p[j] > p[a]) || (p[j] == p[a] && j < a);
        rank[a] = r;
    }
    float leftover = (float)n_disc;
    const float nf = (float)n_disc;
#pragma unroll
    for (int i = 0; i < A; ++i) {  // :47-51
#pragma unroll
        for (int a = 0; a < A; ++a) {
            if (rank[a] == i) {
                const float block = (float)(int32_t)ceilf(nf * p[a]);
                const float x = fminf(leftover, block);
                leftover -= x;
                out[a] += x;
            }
        }
    }
#pragma unroll
    for (int a = 0; a < A; ++a) out[a] /= nf;  // :52
}

__device__ __forceinline__ float clamp_max(float x, float hi) { return x != x ? x : (x < hi ? x : hi); }  // torch.clamp(max=)

// Carry of the reverse scan, learn/vtrace.py:58-67 and :241-247.
struct Carry {
    float r = 0.0f, ru = 0.0f, nv = 0.0f, nvt = 0.0f, is = 1.0f;
};

struct VtHp {
    float neg_eta, lambda_, c, rho, gamma;
};

// One timestep of _loop_v_trace (learn/vtrace.py:249-333) for one lane and one `player`.
//   ours: valid && player_id == player.  oh[a] is the action one-hot VALUE (literal multiply as in :291).
// Writes vt / q[] (zeros unless ours) and advances the carry.
template <int A>
__device__ __forceinline__ void vtrace_step(Carry &cy, const VtHp &hp, bool valid, bool ours, float valid_f, float vv, float rew,
                                            const float (&mu)[A], const float (&pi)[A], const float (&logpi)[A],
                                            const float (&oh)[A], float &vt_out, float (&q_out)[A]) {
    // _policy_ratio (:199-204): sum(a_oh * pi) * valid + (1 - valid)
    float s_pi = 0.0f, s_mu = 0.0f, s_one = 0.0f, ent = 0.0f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        s_pi += oh[a] * pi[a];
        s_mu += oh[a] * mu[a];
        s_one += oh[a] * 1.0f;
        ent += pi[a] * logpi[a];
    }
    const float inv = 1.0f - valid_f;
    const float sel_mu = s_mu * valid_f + inv;
    const float cs = (s_pi * valid_f + inv) / sel_mu;
    const float inv_mu = (s_one * valid_f + inv) / sel_mu;
    const float po = (ours ? 1.0f : -1.0f) * valid_f;  // _player_others (:83-87)
    const float ere = hp.neg_eta * ent * po;           // eta_reg_entropy (:234-238)

    const float ru = rew + hp.gamma * cy.ru + ere;  // reward_uncorrected (:262)
    const float dr = rew + hp.gamma * cy.r;         // discounted_reward (:263)
    const float w = cs * cy.is;
    if (valid && ours) {
        // our_v_target (:266-282)
        const float vt = vv + clamp_max(w, hp.rho) * (ru + hp.gamma * cy.nv - vv) +
                         hp.lambda_ * clamp_max(w, hp.c) * hp.gamma * (cy.nvt - cy.nv);
        const float tail = dr + hp.gamma * cy.is * cy.nvt - vv;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            const float elp = hp.neg_eta * logpi[a] * po;  // eta_log_policy (:239)
            q_out[a] = vv + elp + oh[a] * inv_mu * tail;   // our_learning_output (:288-300)
        }
        vt_out = vt;
        cy.r = 0.0f; cy.ru = 0.0f; cy.nv = vv; cy.nvt = vt; cy.is = 1.0f;  // our_carry (:306-312)
    } else {
        vt_out = 0.0f;
#pragma unroll
        for (int a = 0; a < A; ++a) q_out[a] = 0.0f;
        if (valid) {  // opp_carry (:313-319)
            cy.r = ere + cs * dr; cy.ru = ru; cy.nv = hp.gamma * cy.nv; cy.nvt = hp.gamma * cy.nvt; cy.is = w;
        } else {  // reset_carry (:320)
            cy = Carry{};
        }
    }
}

// x / A, correctly rounded, without the 11-instruction IEEE division sequence: exact for a power of two (multiply by the exact
// reciprocal); otherwise q = x * RN(1 / A) corrected once with the exact residual, q' = fma(fma(-A, q, x), RN(1 / A), q).  Checked
// exhaustively against x / A over every fp32 x (A = 3, 5, 6, 7): identical bits for every |x| >= 1e-30 and for zero (below that
// the two can differ in the last bit of a number that is < 2^-99: nothing downstream resolves it).
template <int A>
__device__ __forceinline__ float div_by(float x) {
#ifdef RNAD_NO_OPT_DIV
    return x / (float)A;
#else
    if constexpr ((A & (A - 1)) == 0) {
        return x * (1.0f / (float)A);
    } else {
        constexpr float r = 1.0f / (float)A;
        const float q = x * r;
        return __builtin_fmaf(__builtin_fmaf(-(float)A, q, x), r, q);
    }
#endif
}

// get_loss_nerd for one row and one player (learn/vtrace.py:410-429), with the closed-form gradient
//   d/dlogit sum_a legal*l*f = w - legal * sum(w) / A,  w = legal * f   (f detached, :367,:418)
template <int A>
__device__ __forceinline__ float nerd_row(const float (&logit)[A], const float (&pi)[A], const float (&q)[A], const float (&legal)[A],
                                          float clip, float thr, float (&grad)[A]) {
    float base = 0.0f, mean = 0.0f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        base += pi[a] * q[a];
        mean += logit[a] * legal[a];
    }
    mean = mean / (float)A;  // torch.mean over ALL A (:420)
    float w[A], wsum = 0.0f, nerd = 0.0f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        float adv = q[a] - base;                                   // :415 (is_c == 1, :416)
        adv = adv != adv ? adv : fminf(fmaxf(adv, -clip), clip);   // :417
        const float l = logit[a] - mean;
        const float f = (l > -thr ? 1.0f : 0.0f) * fminf(adv, 0.0f) + (l < thr ? 1.0f : 0.0f) * fmaxf(adv, 0.0f);  // :362-366
        nerd += legal[a] * (l * f);                                // :424-428
        w[a] = legal[a] * f;
        wsum += w[a];
    }
#pragma unroll
    for (int a = 0; a < A; ++a) grad[a] = w[a] - div_by<A>(legal[a] * wsum);
    return nerd;
}


// floats per row record of the tabular learner (lg[A] | v | v_target | lr[A] | lr2[A] | pad), a multiple of 4
template <int A>
constexpr int kRecStride = (3 * A + 2 + 3) & ~3;

__device__ __forceinline__ float norm_of(const double *norm) {
    const float n = (float)norm[0];
    return n + (n == 0.0f ? 1.0f : 0.0f);  // normalization + (normalization == 0.0)  (:374,:389)
}

}  // namespace dev
}  // namespace rnad
