// row_records.hpp -- the (player, state) row records of the bucketed tabular update: layouts and the one function that fills them.
// Shared by bucket.hip (k_row_records: the five net-output tables -> records) and mlp_rows.hip (the same records out of the epilogue of
// the fused table forward), so that both write the same bits for the same logits.  Citations are baskuit/R-NaD file:line.
#pragma once

#include "learn_math.hpp"

namespace rnad {
namespace dev {

// Row record of the bucketed update, kRowStride<A> floats:  logit[A] | v | v_target | pi_processed[A] | log_policy_reg[A] |
// legal bits | pi[A] | pad   (64 bytes at A = 3; the learner reads the first kRowLearn<A> floats = 48 bytes, the rollout pi).  From the five net-output tables: pi / log_pi = policy head of the learner (rnad.py:373,
// net.py:74-77), pi_processed = process_policy (rnad.py:374), log_policy_reg = log_pi - (alpha log_pi_reg + (1 - alpha) log_pi_reg_)
// (rnad.py:382) -- the per-slot arithmetic of k_learn_fused that does not depend on the slot.
template <int A>
constexpr int kRowStride = (4 * A + 3 + 3) & ~3;
template <int A>
constexpr int kRowLearn = (3 * A + 3 + 3) & ~3;  // what k_bucket_learn fetches of a record

// "Fast" record of the on-policy learner (k_bucket_learn<A, true, .>), kFastStride<A> floats = 64 bytes at A = 3:
//   v | v_target | e0 | bits | pi_processed[A] | elp[A] | cs[A] | inv_mu[A]
// Everything of a slot's V-trace / NeuRD arithmetic whose operands are the row's alone, computed with the operations (and in the
// order) learn_math.hpp's vtrace_step / nerd_row use per slot, once per row instead:
//   e0 = -eta * sum_a pi_processed[a] * log_policy_reg[a]   eta_reg_entropy up to the sign of _player_others (vtrace.py:234-238)
//   elp[a] = -eta * log_policy_reg[a]                        eta_log_policy of the mover (:239)
//   cs[a] = pi_processed[a] / pi[a], inv_mu[a] = 1 / pi[a]   _policy_ratio with the actor's own pi as mu, had action a been taken (:199-204)
//   bits = legal | (legal & logit - mean > -threshold) << 8 | (legal & logit - mean < threshold) << 16    the gates of
//          apply_force_with_threshold (:362-366), closed for illegal actions (whose force :424-428 multiplies by legal == 0)
template <int A>
constexpr int kFastStride = 4 + 4 * A;
// The actor's policy rows as a table of their own (16 bytes per row at A <= 4): the whole configs[1] table is 2 MB and stays in an
// XCD's L2, where the 64-byte row records -- 8.5 MB, of which the rollout wants 12 bytes per row -- do not.
template <int A>
constexpr int kPolStride = (A + 3) & ~3;

// Records of row r from its five net outputs: lg = the learner's logits, vr / vtr = the learner's / target's value, lr / lr2 = the
// logits of the two regularisation nets, bits = the mover's legal-action bits.  rec / fast / pol_rows: tables (any may be NULL).
template <int A>
__device__ __forceinline__ void write_row_records(int64_t r, const float (&lg)[A], float vr, float vtr, const float (&lr)[A],
                                                  const float (&lr2)[A], uint32_t bits, const rnad_learn_params_t &hp,
                                                  float *__restrict__ rec, float *__restrict__ fast, float *__restrict__ pol_rows) {
    float legal[A], pi[A], lp[A], lpr[A], lpr2[A], pip[A];
#pragma unroll
    for (int a = 0; a < A; ++a) legal[a] = (float)((bits >> a) & 1);
    policy_head<A>(lg, bits, pi, lp);
    log_policy_only<A>(lr, bits, lpr);
    log_policy_only<A>(lr2, bits, lpr2);
    process_policy_row<A>(pi, legal, hp.n_disc, hp.eps_threshold, pip);
    // both records are assembled in registers and leave as 16-byte stores (a store instruction per float would touch 64 different
    // 64-byte segments each)
    if (rec) {
        float o[kRowStride<A>];
#pragma unroll
        for (int a = 0; a < A; ++a) {
            o[a] = lg[a];
            o[A + 2 + a] = pip[a];
            o[2 * A + 2 + a] = lp[a] - (hp.alpha * lpr[a] + hp.one_minus_alpha * lpr2[a]);
            o[3 * A + 3 + a] = pi[a];
        }
        o[A] = vr;
        o[A + 1] = vtr;
        o[3 * A + 2] = __uint_as_float(bits);
#pragma unroll
        for (int u = 4 * A + 3; u < kRowStride<A>; ++u) o[u] = 0.0f;
        float4 *o4 = reinterpret_cast<float4 *>(rec + r * kRowStride<A>);
#pragma unroll
        for (int u = 0; u < kRowStride<A> / 4; ++u) o4[u] = float4{o[4 * u], o[4 * u + 1], o[4 * u + 2], o[4 * u + 3]};
    }
    if (pol_rows) {  // the actor's policy rows on their own, kPolStride<A> floats apart: the table the rollout kernels gather from
        float4 *p4 = reinterpret_cast<float4 *>(pol_rows + r * kPolStride<A>);
#pragma unroll
        for (int u = 0; u < kPolStride<A> / 4; ++u)
            p4[u] = float4{4 * u < A ? pi[4 * u] : 0.0f, 4 * u + 1 < A ? pi[4 * u + 1] : 0.0f, 4 * u + 2 < A ? pi[4 * u + 2] : 0.0f,
                           4 * u + 3 < A ? pi[4 * u + 3] : 0.0f};
    }
    if (!fast) return;
    float f[kFastStride<A>];
    const float neg_eta = -hp.eta;
    float ent = 0.0f, mean = 0.0f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        const float lpol = lp[a] - (hp.alpha * lpr[a] + hp.one_minus_alpha * lpr2[a]);
        ent += pip[a] * lpol;
        mean += lg[a] * legal[a];
        f[4 + a] = pip[a];
        f[4 + A + a] = neg_eta * lpol;
        f[4 + 2 * A + a] = pip[a] / pi[a];
        f[4 + 3 * A + a] = 1.0f / pi[a];
    }
    mean = mean / (float)A;
    uint32_t gates = bits & 0xffu;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        const float l = lg[a] - mean;
        // (an illegal action's force is multiplied by legal == 0 further down, vtrace.py:424-428: its gates are left closed instead)
        gates |= ((l > -hp.threshold ? 1u : 0u) & (bits >> a)) << (8 + a);
        gates |= ((l < hp.threshold ? 1u : 0u) & (bits >> a)) << (16 + a);
    }
    f[0] = vr;
    f[1] = vtr;
    f[2] = neg_eta * ent;
    f[3] = __uint_as_float(gates);
    float4 *f4 = reinterpret_cast<float4 *>(fast + r * kFastStride<A>);
#pragma unroll
    for (int u = 0; u < kFastStride<A> / 4; ++u) f4[u] = float4{f[4 * u], f[4 * u + 1], f[4 * u + 2], f[4 * u + 3]};
}

}  // namespace dev
}  // namespace rnad
