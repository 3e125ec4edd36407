/*
 * rnad_oracle.c -- CPU restatement of the baskuit/R-NaD self-play hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP kernels in
 * r-nad_amd/csrc/: only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
 * load it.  The product (r-nad_amd/) never links, imports or falls back to anything in oracle/.
 *
 * Parity status: PINNED.  Every function below is checked in tests/test_oracle_golden.py against
 * fixtures under tests/golden/ that were produced by importing the reference itself
 * (tests/golden/make_golden.py).  Tensors use the REFERENCE's layouts ([S,C,A,A] tables, int64
 * indices, one-hot actions), not the packed layouts of the HIP library, so that the packing code
 * is checked too.  One exception, stated where it stands: the reference draws with
 * torch.multinomial from torch's global generator, which has no counterpart in a seeded rollout;
 * oracle_sample / oracle_transition restate torch's algorithm on recorded noise (pinned by the
 * golden rollouts), oracle_pick / oracle_transition_pick / oracle_uniforms restate the seeded
 * draw of include/rnad_rng.h (pinned by its documented counter layout, the philox known answers,
 * its definition in numpy and chi-square tests: tests/test_rng.py).
 *
 * All arithmetic is fp32 in the reference's operation order; build with -ffp-contract=off so the
 * compiler cannot fuse a*b+c (the reference runs one torch op per arithmetic step).
 *
 * Citations are file:line in /root/reference (baskuit/R-NaD).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "rnad_rng.h"

#define EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------------
 * environment/episode.py:62-68  States.observations()
 *   row view : cat([ev, legal], dim=1)                 obs[b,0,i,j] =  ev[s,i,j]   obs[b,1,i,j] = legal[s,i,j]
 *   col view : cat([-ev, legal], dim=1).swapaxes(2,3)  obs[b,0,i,j] = -ev[s,j,i]   obs[b,1,i,j] = legal[s,j,i]
 *   the view is picked per lane by player_to_move; `-ev` turns +0.0 into -0.0 and that is kept.
 * environment/episode.py:208      masks = observations[:, 1, :, 0]
 * ---------------------------------------------------------------------------------------------- */
EXPORT void oracle_observe(int64_t B, int A, const float *ev, const float *legal, const int64_t *idx,
                           const int64_t *player, float *obs, float *mask) {
    const int AA = A * A;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        const float *e = ev + idx[b] * AA;
        const float *l = legal + idx[b] * AA;
        float *o = obs + b * 2 * AA;
        for (int i = 0; i < A; ++i)
            for (int j = 0; j < A; ++j) {
                if (player[b] == 0) {
                    o[i * A + j] = e[i * A + j];
                    o[AA + i * A + j] = l[i * A + j];
                } else {
                    o[i * A + j] = -e[j * A + i];
                    o[AA + i * A + j] = l[j * A + i];
                }
            }
        if (mask)
            for (int i = 0; i < A; ++i) mask[b * A + i] = o[AA + i * A];
    }
}

/* ------------------------------------------------------------------------------------------------
 * nn/net.py:49 and environment/episode.py:118   torch.multinomial(p, num_samples=1) on CPU.
 * torch (aten/src/ATen/native/Distributions.cpp, multinomial_out, n_sample == 1 path) computes
 *     q = empty_like(p).exponential_(1);  q = p / q;  result = argmax(q, dim=-1)
 * argmax keeps the FIRST maximal element.  `noise` is that q (recorded in the fixtures).
 * ---------------------------------------------------------------------------------------------- */
static inline int race_argmax(int n, const float *p, const float *q) {
    int best = 0;
    float bv = p[0] / q[0];
    for (int a = 1; a < n; ++a) {
        float r = p[a] / q[a];
        if (r > bv) {
            bv = r;
            best = a;
        }
    }
    return best;
}

EXPORT void oracle_sample(int64_t B, int A, const float *policy, const float *noise, int64_t *actions) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) actions[b] = race_argmax(A, policy + b * A, noise + b * A);
}

/* The SEEDED draw of include/rnad_rng.h (shared, bit for bit, with the HIP kernels): torch.multinomial's
 * contract -- one sample of Cat(p / sum p) -- from the decision's own counter-based uniform by the
 * inverse CDF.  u[b] is lane b's uniform for this decision (oracle_uniforms). */
EXPORT void oracle_pick(int64_t B, int A, const float *policy, const float *u, int64_t *actions) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) actions[b] = rnad_pick(policy + b * A, A, u[b]);
}

/* u[b, 0..2] = the uniforms of lane lane0 + b for the game transition env step t belongs to:
 * [0] row player's action, [1] column player's action, [2] chance outcome. */
EXPORT void oracle_uniforms(int64_t B, uint64_t seed, int64_t lane0, int t, float *u) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) rnad_decision_uniforms(seed, (uint64_t)(lane0 + b), (uint32_t)t, u + 3 * b);
}

/* Reproducible Exp(1) noise for the explicit-noise entry points: lane `lane`, step `t`, stream
 * `stream` of the counter-based generator in rnad_rng.h. */
EXPORT void oracle_noise(int64_t B, int n, uint64_t seed, int64_t lane0, int t, int stream, float *noise) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) rnad_exp_noise(seed, (uint64_t)(lane0 + b), (uint32_t)t, (uint32_t)stream, n, noise + b * n);
}

/* ------------------------------------------------------------------------------------------------
 * environment/episode.py:96-125  States.step(), column-player branch (:102-123)
 *   p = chance[s,:,r,c]; t = multinomial(p); s' = index[s,t,r,c]; reward = value[s,t,r,c] * (s'==0)
 * The row-player branch (:99-101) only stashes the action and returns zeros; callers do that.
 * ---------------------------------------------------------------------------------------------- */
EXPORT void oracle_transition(int64_t B, int A, int C, const int64_t *index_t, const float *chance,
                              const float *value, const int64_t *idx, const int64_t *row_a,
                              const int64_t *col_a, const float *noise, int64_t *idx_out, float *reward) {
    const int AA = A * A;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        float p[64];
        const int64_t base = idx[b] * C * AA + row_a[b] * A + col_a[b];
        for (int t = 0; t < C; ++t) p[t] = chance[base + (int64_t)t * AA];
        const int t = race_argmax(C, p, noise + b * C);
        const int64_t nxt = index_t[base + (int64_t)t * AA];
        idx_out[b] = nxt;
        reward[b] = value[base + (int64_t)t * AA] * (nxt == 0 ? 1.0f : 0.0f); /* rewards *= (indices == 0) */
    }
}

/* The same step with the seeded chance draw: outcome = rnad_pick(chance[s,:,r,c], u[b]). */
EXPORT void oracle_transition_pick(int64_t B, int A, int C, const int64_t *index_t, const float *chance,
                                   const float *value, const int64_t *idx, const int64_t *row_a,
                                   const int64_t *col_a, const float *u, int64_t *idx_out, float *reward) {
    const int AA = A * A;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        float p[64];
        const int64_t base = idx[b] * C * AA + row_a[b] * A + col_a[b];
        for (int t = 0; t < C; ++t) p[t] = chance[base + (int64_t)t * AA];
        const int t = rnad_pick(p, C, u[b]);
        const int64_t nxt = index_t[base + (int64_t)t * AA];
        idx_out[b] = nxt;
        reward[b] = value[base + (int64_t)t * AA] * (nxt == 0 ? 1.0f : 0.0f);
    }
}

/* ------------------------------------------------------------------------------------------------
 * nn/net.py:37-51 (forward) / :64-85 (forward_batch): the policy head on top of the logits.
 *   exp_logits = where(legal, exp(logits), 0); policy = exp_logits / max(sum|exp_logits|, 1e-12)
 *   log_policy = where(legal, logits - log(sum(exp_logits)), 0)
 * ---------------------------------------------------------------------------------------------- */
EXPORT void oracle_policy_head(int64_t N, int A, const float *logits, const float *mask, float *policy,
                               float *log_policy) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        float ex[64];
        float s = 0.0f;
        for (int a = 0; a < A; ++a) {
            ex[a] = mask[n * A + a] != 0.0f ? expf(logits[n * A + a]) : 0.0f;
            s += fabsf(ex[a]);
        }
        const float d = s > 1e-12f ? s : 1e-12f;
        for (int a = 0; a < A; ++a) policy[n * A + a] = ex[a] / d;
        if (log_policy) {
            float s2 = 0.0f;
            for (int a = 0; a < A; ++a) s2 += ex[a];
            const float ls = logf(s2);
            for (int a = 0; a < A; ++a)
                log_policy[n * A + a] = mask[n * A + a] != 0.0f ? logits[n * A + a] - ls : 0.0f;
        }
    }
}

/* nn/net.py:42-43: value = fc1(relu(fc0 x)), logits = pfc1(relu(pfc0 x)); weights in torch Linear
 * layout ([out, in] row-major).  Used by the CPU-baseline rollout and the NashConv inference. */
EXPORT void oracle_mlp_forward(int64_t N, int A, int W, const float *vw0, const float *vb0, const float *vw1,
                               const float *vb1, const float *pw0, const float *pb0, const float *pw1,
                               const float *pb1, const float *x, float *logits, float *value) {
    const int K = 2 * A * A;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        const float *xi = x + n * K;
        float accv = vb1[0];
        float accp[64];
        for (int a = 0; a < A; ++a) accp[a] = pb1[a];
        for (int h = 0; h < W; ++h) {
            float sv = vb0[h], sp = pb0[h];
            for (int k = 0; k < K; ++k) {
                sv += vw0[h * K + k] * xi[k];
                sp += pw0[h * K + k] * xi[k];
            }
            sv = sv > 0.0f ? sv : 0.0f;
            sp = sp > 0.0f ? sp : 0.0f;
            accv += vw1[h] * sv;
            for (int a = 0; a < A; ++a) accp[a] += pw1[a * W + h] * sp;
        }
        value[n] = accv;
        for (int a = 0; a < A; ++a) logits[n * A + a] = accp[a];
    }
}

/* ------------------------------------------------------------------------------------------------
 * learn/vtrace.py:24-55  process_policy(policy, mask, n_disc, epsilon_threshold)
 *   mask   = mask * ((p >= eps) + (max(p) < eps))                       (:34-39)
 *   p      = mask*p / sum(mask*p)                                       (:40)
 *   blocks = ceil(n*p) as int32; visit actions in argsort(p, descending) order (ties: lower index
 *   first, as torch's CPU sort does for these sizes); x = min(leftover, block); leftover -= x (:43-51)
 *   result = x / n                                                       (:52)
 * ---------------------------------------------------------------------------------------------- */
EXPORT void oracle_process_policy(int64_t N, int A, const float *policy, const float *mask, int n_disc,
                                  float eps, float *out) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        float p[64], m[64];
        int order[64];
        const float *pi = policy + n * A;
        float mx = pi[0];
        for (int a = 1; a < A; ++a) mx = pi[a] > mx ? pi[a] : mx;
        const float all_below = mx < eps ? 1.0f : 0.0f;
        float s = 0.0f;
        for (int a = 0; a < A; ++a) {
            /* bool + bool -> bool (logical or), then mask(float) * bool */
            const float keep = ((pi[a] >= eps ? 1.0f : 0.0f) + all_below) != 0.0f ? 1.0f : 0.0f;
            m[a] = mask[n * A + a] * keep;
            s += m[a] * pi[a];
        }
        for (int a = 0; a < A; ++a) p[a] = m[a] * pi[a] / s;
        for (int a = 0; a < A; ++a) order[a] = a;
        for (int i = 1; i < A; ++i) { /* stable insertion sort, descending */
            int k = order[i], j = i - 1;
            while (j >= 0 && p[order[j]] < p[k]) {
                order[j + 1] = order[j];
                --j;
            }
            order[j + 1] = k;
        }
        float leftover = (float)n_disc;
        for (int a = 0; a < A; ++a) out[n * A + a] = 0.0f;
        for (int i = 0; i < A; ++i) {
            const int a = order[i];
            const float block = (float)(int32_t)ceilf((float)n_disc * p[a]);
            const float x = leftover < block ? leftover : block;
            leftover -= x;
            out[n * A + a] += x;
        }
        for (int a = 0; a < A; ++a) out[n * A + a] /= (float)n_disc;
    }
}

/* ------------------------------------------------------------------------------------------------
 * learn/vtrace.py:207-352  v_trace(...) for one `player`, incl. _has_played (:141-177),
 * _policy_ratio (:180-204), _player_others (:70-87) and the reverse scan (:249-350).
 *
 * Layouts: v [T,B,1], valid [T,B] f32, player_id [T,B] i64, mu/pi/logpi/a_oh [T,B,A], reward [T,B].
 * Outputs: v_target [T,B,1], has_played [T,B] i64, q (= learning_output) [T,B,A].
 * `_has_played` never sets its carry (:155-157), so has_played == valid && player_id == player.
 * ---------------------------------------------------------------------------------------------- */
EXPORT void oracle_vtrace(int T, int64_t B, int A, const float *v, const float *valid, const int64_t *player_id,
                          const float *mu, const float *pi, const float *logpi, const float *a_oh,
                          const float *reward, int player, float eta, float lambda_, float c, float rho,
                          float gamma, float *v_target, int64_t *has_played, float *q) {
    const float neg_eta = -eta;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        /* LoopVTraceCarry init (:241-247) */
        float c_r = 0.0f, c_ru = 0.0f, c_nv = 0.0f, c_nvt = 0.0f, c_is = 1.0f;
        for (int t = T - 1; t >= 0; --t) {
            const int64_t i = (int64_t)t * B + b;
            const float val = valid[i];
            const int ours = player_id[i] == player;
            /* _policy_ratio (:199-204): sum(a_oh * pi) * valid + (1 - valid) */
            float s_pi = 0.0f, s_mu = 0.0f, s_one = 0.0f, ent = 0.0f;
            for (int a = 0; a < A; ++a) {
                s_pi += a_oh[i * A + a] * pi[i * A + a];
                s_mu += a_oh[i * A + a] * mu[i * A + a];
                s_one += a_oh[i * A + a] * 1.0f;
                ent += pi[i * A + a] * logpi[i * A + a];
            }
            const float sel_pi = s_pi * val + (1.0f - val);
            const float sel_mu = s_mu * val + (1.0f - val);
            const float sel_one = s_one * val + (1.0f - val);
            const float cs = sel_pi / sel_mu;
            const float inv_mu = sel_one / sel_mu;
            /* _player_others (:83-87): (2*(pid==player) - 1) * valid */
            const float po = (float)(2 * ours - 1) * val;
            const float ere = neg_eta * ent * po; /* eta_reg_entropy (:234-238) */
            const float rew = reward[i];
            const float vv = v[i];

            const float ru = rew + gamma * c_ru + ere; /* reward_uncorrected (:262) */
            const float dr = rew + gamma * c_r;        /* discounted_reward  (:263) */
            const float w = cs * c_is;
            const float wr = w < rho ? w : (w != w ? w : rho); /* clamp(max=rho), NaN propagates */
            const float wc = w < c ? w : (w != w ? w : c);
            /* our_v_target (:266-282) */
            const float vt = vv + wr * (ru + gamma * c_nv - vv) + lambda_ * wc * gamma * (c_nvt - c_nv);
            has_played[i] = (val != 0.0f && ours) ? 1 : 0;
            if (val != 0.0f && ours) {
                v_target[i] = vt;
                for (int a = 0; a < A; ++a) {
                    const float elp = neg_eta * logpi[i * A + a] * po; /* eta_log_policy (:239) */
                    /* our_learning_output (:288-300) */
                    q[i * A + a] = vv + elp + a_oh[i * A + a] * inv_mu * (dr + gamma * c_is * c_nvt - vv);
                }
                c_r = 0.0f; c_ru = 0.0f; c_nv = vv; c_nvt = vt; c_is = 1.0f; /* our_carry (:306-312) */
            } else {
                v_target[i] = 0.0f;
                for (int a = 0; a < A; ++a) q[i * A + a] = 0.0f;
                if (val != 0.0f) { /* opp_carry (:313-319) */
                    c_r = ere + cs * dr; c_ru = ru; c_nv = gamma * c_nv; c_nvt = gamma * c_nvt; c_is = w;
                } else { /* reset_carry (:320) */
                    c_r = 0.0f; c_ru = 0.0f; c_nv = 0.0f; c_nvt = 0.0f; c_is = 1.0f;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * learn/vtrace.py:377-393  get_loss_v, with its closed-form gradient.
 *   L = sum_P sum(m_P * (v - vt_P)^2) / max(N_P, 1),  N_P = sum(m_P);  dL/dv = sum_P 2 m_P (v - vt_P) / N_P
 * m_P is the int64 has_played mask.  `dv` (may be NULL) is ACCUMULATED into with weight `scale`.
 * ---------------------------------------------------------------------------------------------- */
EXPORT double oracle_loss_v(int64_t N, const float *v, const float *vt0, const float *vt1, const int64_t *m0,
                            const int64_t *m1, float scale, float *dv) {
    const float *vts[2] = {vt0, vt1};
    const int64_t *ms[2] = {m0, m1};
    double total = 0.0;
    for (int p = 0; p < 2; ++p) {
        double s = 0.0;
        int64_t cnt = 0;
        for (int64_t n = 0; n < N; ++n) {
            const float d = v[n] - vts[p][n];
            s += (double)((float)ms[p][n] * (d * d));
            cnt += ms[p][n];
        }
        const float norm = (float)(cnt + (cnt == 0));
        total += s / norm;
        if (dv)
            for (int64_t n = 0; n < N; ++n) dv[n] += scale * (2.0f * (float)ms[p][n] * (v[n] - vts[p][n]) / norm);
    }
    return total;
}

/* ------------------------------------------------------------------------------------------------
 * learn/vtrace.py:396-431  get_loss_nerd (+ apply_force_with_threshold :355-367, renormalize :370-374)
 *   adv  = clip(q_k - sum_a(pi * q_k), -clip, clip)                     (:415-417, is_c == 1)
 *   l    = logit - mean_a(logit * legal)        (mean over ALL A)        (:420)
 *   f    = (l > -thr) * min(adv, 0) + (l < thr) * max(adv, 0)            (:362-366)
 *   L_k  = - sum(m_k * sum_a legal * l * f) / max(sum m_k, 1),  m_k = valid * (player_id == k)   (:424-429)
 *   dL/dlogit = - m_k (w - legal * sum_a(w) / A) / N_k with w = legal * f   (f is detached, :367,:418)
 * ---------------------------------------------------------------------------------------------- */
EXPORT double oracle_loss_nerd(int64_t N, int A, const float *logit, const float *pi, const float *q0,
                               const float *q1, const float *valid, const int64_t *player_id,
                               const float *legal, float clip, float thr, float scale, float *dlogit) {
    const float *qs[2] = {q0, q1};
    double total = 0.0;
    for (int k = 0; k < 2; ++k) {
        double cnt = 0.0;
        for (int64_t n = 0; n < N; ++n) cnt += valid[n] * (player_id[n] == k ? 1.0f : 0.0f);
        const float norm = (float)(cnt + (cnt == 0.0));
        double s = 0.0;
        for (int64_t n = 0; n < N; ++n) {
            const float m = valid[n] * (player_id[n] == k ? 1.0f : 0.0f);
            const float *qq = qs[k] + n * A;
            float base = 0.0f, mean = 0.0f;
            for (int a = 0; a < A; ++a) {
                base += pi[n * A + a] * qq[a];
                mean += logit[n * A + a] * legal[n * A + a];
            }
            mean = mean / (float)A;
            float w[64], wsum = 0.0f, nerd = 0.0f;
            for (int a = 0; a < A; ++a) {
                float adv = qq[a] - base;
                adv = adv < -clip ? -clip : (adv > clip ? clip : adv);
                const float l = logit[n * A + a] - mean;
                const float f = (l > -thr ? 1.0f : 0.0f) * (adv < 0.0f ? adv : 0.0f) +
                                (l < thr ? 1.0f : 0.0f) * (adv > 0.0f ? adv : 0.0f);
                nerd += legal[n * A + a] * (l * f);
                w[a] = legal[n * A + a] * f;
                wsum += w[a];
            }
            s += (double)(nerd * m);
            if (dlogit)
                for (int a = 0; a < A; ++a)
                    dlogit[n * A + a] += scale * (-(m * (w[a] - legal[n * A + a] * wsum / (float)A)) / norm);
        }
        total += -(s / norm);
    }
    return total;
}

/* ------------------------------------------------------------------------------------------------
 * util/metric.py:93-175  NashConvData.get_nashconv (recursive best-response values).
 * Faithful to two quirks of the reference:
 *   - the recursion passes self.joint_policy, not its `joint_policy` argument (:148-151): the state
 *     `state_index` the call starts from uses `root_policy`, every descendant uses `table_policy`;
 *   - reach probabilities index the flattened outer product pi_col x pi_row (:130-132) with the
 *     [t, r, c] flat index, i.e. they use pi_col[r] * pi_row[c].
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int A, C;
    const int64_t *index_t;
    const float *value, *chance, *legal, *table_policy;
    float *row_best, *col_best, *reach;
    int32_t *depth;
} nc_ctx;

static void nc_rec(const nc_ctx *x, const float *pi, int64_t s, float reach, int depth) {
    const int A = x->A, C = x->C, AA = A * A;
    float *rowm = (float *)calloc((size_t)C * AA, sizeof(float));
    float *colm = (float *)calloc((size_t)C * AA, sizeof(float));
    const float *pi_row = pi, *pi_col = pi + A;
    int maxd = 0;
    for (int f = 0; f < C * AA; ++f) {
        const float tp = x->chance[s * C * AA + f];
        if (!(tp > 0.0f)) continue;
        const int64_t child = x->index_t[s * C * AA + f];
        float rb, cb;
        if (child == 0) {
            rb = x->value[s * C * AA + f];
            cb = -rb;
        } else {
            const int rc = f % AA;
            const float jp = pi_col[rc / A] * pi_row[rc % A]; /* flatten(matmul(pi_col, pi_row)) (:130-132) */
            nc_rec(x, x->table_policy + child * 2 * A, child, reach * jp * tp, depth + 1);
            rb = x->row_best[child];
            cb = x->col_best[child];
            if (x->depth[child] > maxd) maxd = x->depth[child];
        }
        rowm[f] = rb * tp;
        colm[f] = cb * tp;
    }
    float best_r = -INFINITY, best_c = -INFINITY;
    for (int i = 0; i < A; ++i) { /* row responses: (sum_t rowm)[i,:] . pi_col, legal rows only (:167-172) */
        if (x->legal[s * AA + i * A] == 0.0f) continue;
        float acc = 0.0f;
        for (int j = 0; j < A; ++j) {
            float m = 0.0f;
            for (int t = 0; t < C; ++t) m += rowm[t * AA + i * A + j];
            acc += m * pi_col[j];
        }
        if (acc > best_r) best_r = acc;
    }
    for (int j = 0; j < A; ++j) {
        if (x->legal[s * AA + j] == 0.0f) continue;
        float acc = 0.0f;
        for (int i = 0; i < A; ++i) {
            float m = 0.0f;
            for (int t = 0; t < C; ++t) m += colm[t * AA + i * A + j];
            acc += pi_row[i] * m;
        }
        if (acc > best_c) best_c = acc;
    }
    x->row_best[s] = best_r;
    x->col_best[s] = best_c;
    x->reach[s] = reach;
    x->depth[s] = 1 + maxd;
    free(rowm);
    free(colm);
}

EXPORT void oracle_nashconv(int A, int C, const int64_t *index_t, const float *value, const float *chance,
                            const float *legal, const float *root_policy, const float *table_policy,
                            int64_t state_index, float reach, float *row_best, float *col_best,
                            float *reach_out, int32_t *depth_out) {
    nc_ctx x = {A, C, index_t, value, chance, legal, table_policy, row_best, col_best, reach_out, depth_out};
    nc_rec(&x, root_policy, state_index, reach, 0);
}

/* ------------------------------------------------------------------------------------------------
 * environment/episode.py:175-230  Episodes.generate(): the whole rollout loop on the CPU, with the
 * MLP above and the seeded sampler.  Used as bench.py's cpu_baseline ("port") and to check the HIP
 * rollout driver end to end.  Trajectory tensors are [T,B,...] as the reference stacks them (:218-225).
 * Returns the number of steps T actually played (loop ends when every lane is absorbed, :194).
 * ---------------------------------------------------------------------------------------------- */
EXPORT int oracle_rollout(int64_t B, int A, int C, int W, int T_cap, const int64_t *index_t, const float *value,
                          const float *chance, const float *ev, const float *legal, const float *const *w,
                          uint64_t seed, int64_t lane0, int64_t *indices, float *observations, float *masks,
                          float *policy, int64_t *actions, float *rewards, float *values, float *logits_out) {
    const int AA = A * A;
    int64_t *idx = (int64_t *)malloc(sizeof(int64_t) * B);
    int64_t *player = (int64_t *)malloc(sizeof(int64_t) * B);
    int64_t *row_a = (int64_t *)malloc(sizeof(int64_t) * B);
    float *u3 = (float *)malloc(sizeof(float) * B * 3), *u = (float *)malloc(sizeof(float) * B);
    float *logits = (float *)malloc(sizeof(float) * B * A);
    for (int64_t b = 0; b < B; ++b) idx[b] = 1;
    int t = 0;
    for (; t < T_cap; ++t) {
        int alive = 0;
        for (int64_t b = 0; b < B; ++b) alive |= idx[b] != 0;
        if (!alive) break;
        float *obs_t = observations + (int64_t)t * B * 2 * AA;
        float *mask_t = masks + (int64_t)t * B * A;
        float *pol_t = policy + (int64_t)t * B * A;
        int64_t *act_t = actions + (int64_t)t * B;
        for (int64_t b = 0; b < B; ++b) player[b] = t & 1;
        memcpy(indices + (int64_t)t * B, idx, sizeof(int64_t) * B);
        oracle_observe(B, A, ev, legal, idx, player, obs_t, mask_t);
        oracle_mlp_forward(B, A, W, w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], obs_t, logits,
                           values + (int64_t)t * B);
        if (logits_out) memcpy(logits_out + (int64_t)t * B * A, logits, sizeof(float) * B * A);
        oracle_policy_head(B, A, logits, mask_t, pol_t, NULL);
        if ((t & 1) == 0) oracle_uniforms(B, seed, lane0, t, u3);  /* one generator call per game transition */
        for (int64_t b = 0; b < B; ++b) u[b] = u3[3 * b + (t & 1)];
        oracle_pick(B, A, pol_t, u, act_t);
        if ((t & 1) == 0) {
            memcpy(row_a, act_t, sizeof(int64_t) * B);
            memset(rewards + (int64_t)t * B, 0, sizeof(float) * B);
        } else {
            for (int64_t b = 0; b < B; ++b) u[b] = u3[3 * b + 2];
            oracle_transition_pick(B, A, C, index_t, chance, value, idx, row_a, act_t, u, idx, rewards + (int64_t)t * B);
        }
    }
    free(idx); free(player); free(row_a); free(u3); free(u); free(logits);
    return t;
}

/* Known-answer hook for the counter-based generator (tests/test_rng.py). */
EXPORT void oracle_philox(uint32_t *ctr, uint32_t k0, uint32_t k1) { rnad_philox4x32_10(ctr, k0, k1); }
EXPORT float oracle_neg_log_u(uint32_t x) { return rnad_neg_log_u(x); }
EXPORT float oracle_uniform(uint32_t x) { return rnad_uniform(x); }
