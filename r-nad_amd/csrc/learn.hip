// learn.hip -- the R-NaD / NeuRD update with two-player V-trace (gfx950).
//
// Replaces: learn/vtrace.py:24-55 (process_policy, K4), :141-352 (v_trace and helpers, K5), :355-431 (losses, K6) and the
// tensor program of learn/rnad.py:365-425 between the four forward_batch calls and loss.backward() (fused kernel).
// Citations are baskuit/R-NaD file:line.
//
// Mapping: one lane (thread) per episode, sequential over the T <= 32 timesteps.  With the reference's [T, B, ...]
// layout every per-step load is a full-width coalesced access across the 64 lanes of a wave; the scan carry lives in
// registers.  The V-trace recurrence is not an associative scan -- the carry goes through min(cs * is, rho) and a
// three-way select per step -- and B >> 256 CUs x 2048 lanes, so parallelism comes from the batch, not from T.
// All fp32 expressions keep the reference's association order; the file is built with -ffp-contract=off.
#include "common.hpp"
#include "learn_math.hpp"

#include <algorithm>

using namespace rnad;
using namespace rnad::dev;

namespace {

constexpr int kThreads = 256;
inline unsigned blocks_for(int64_t n) { return (unsigned)((n + kThreads - 1) / kThreads); }

// Sum `x` over the block in double and add it to *dst with one atomic.
__device__ __forceinline__ void block_atomic_add(double x, double *dst) {
    __shared__ double part[kThreads / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < kThreads / 64; ++i) s += part[i];
        if (s != 0.0) atomicAdd(dst, s);
    }
}

// ---------------------------------------------------------------------------------------- kernels
template <int A>
__global__ __launch_bounds__(kThreads) void k_process_policy(int64_t N, const float *__restrict__ policy, const float *__restrict__ mask,
                                                             int n_disc, float eps, float *__restrict__ out) {
    const int64_t n = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (n >= N) return;
    float pi[A], m[A], o[A];
#pragma unroll
    for (int a = 0; a < A; ++a) {
        pi[a] = policy[n * A + a];
        m[a] = mask[n * A + a];
    }
    process_policy_row<A>(pi, m, n_disc, eps, o);
#pragma unroll
    for (int a = 0; a < A; ++a) out[n * A + a] = o[a];
}

template <int A, bool ONEHOT>
__global__ __launch_bounds__(kThreads) void k_vtrace(int T, int64_t B, const float *__restrict__ v, const float *__restrict__ valid,
                                                     const int32_t *__restrict__ player_id, const float *__restrict__ mu,
                                                     const float *__restrict__ pi, const float *__restrict__ logpi,
                                                     const void *__restrict__ actions, const float *__restrict__ reward, int player,
                                                     VtHp hp, float *__restrict__ v_target, int32_t *__restrict__ has_played,
                                                     float *__restrict__ q) {
    const int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (b >= B) return;
    Carry cy;
    for (int t = T - 1; t >= 0; --t) {
        const int64_t i = (int64_t)t * B + b;
        const float val = valid[i];
        const bool ours = (player_id ? player_id[i] : (t & 1)) == player;
        float m[A], p[A], lp[A], oh[A], qo[A];
        int act = 0;
        if (!ONEHOT) act = ((const int32_t *)actions)[i];
#pragma unroll
        for (int a = 0; a < A; ++a) {
            m[a] = mu[i * A + a];
            p[a] = pi[i * A + a];
            lp[a] = logpi[i * A + a];
            oh[a] = ONEHOT ? ((const float *)actions)[i * A + a] : (act == a ? 1.0f : 0.0f);
        }
        float vt;
        vtrace_step<A>(cy, hp, val != 0.0f, ours, val, v[i], reward[i], m, p, lp, oh, vt, qo);
        v_target[i] = vt;
        if (has_played) has_played[i] = (val != 0.0f && ours) ? 1 : 0;  // _has_played (:141-177): its carry is never set
#pragma unroll
        for (int a = 0; a < A; ++a) q[i * A + a] = qo[a];
    }
}

__global__ __launch_bounds__(kThreads) void k_mask_sum(int64_t N, const float *__restrict__ mask, double *__restrict__ out) {
    double s = 0.0;
    for (int64_t n = (int64_t)blockIdx.x * kThreads + threadIdx.x; n < N; n += (int64_t)gridDim.x * kThreads) s += (double)mask[n];
    block_atomic_add(s, out);
}

template <bool ACC>
__global__ __launch_bounds__(kThreads) void k_loss_v(int64_t N, const float *__restrict__ v, const float *__restrict__ vt,
                                                     const float *__restrict__ mask, const double *__restrict__ norm, float weight,
                                                     double *__restrict__ loss, float *__restrict__ dv) {
    const int64_t n = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const float nf = norm_of(norm);
    double part = 0.0;
    if (n < N) {
        const float m = mask[n];
        const float d = v[n] - vt[n];
        part = (double)(m * (d * d));  // :387
        if (dv) {
            const float g = weight * (2.0f * m * d / nf);
            dv[n] = ACC ? dv[n] + g : g;
        }
    }
    if (loss) block_atomic_add(part / (double)nf, loss);
}

template <int A, bool ACC>
__global__ __launch_bounds__(kThreads) void k_loss_nerd(int64_t N, const float *__restrict__ logit, const float *__restrict__ pi,
                                                        const float *__restrict__ q, const float *__restrict__ mask,
                                                        const float *__restrict__ legal, const double *__restrict__ norm, float clip,
                                                        float thr, float weight, double *__restrict__ loss,
                                                        float *__restrict__ dlogit) {
    const int64_t n = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const float nf = norm_of(norm);
    double part = 0.0;
    if (n < N) {
        float l[A], p[A], qq[A], lg[A], g[A];
#pragma unroll
        for (int a = 0; a < A; ++a) {
            l[a] = logit[n * A + a];
            p[a] = pi[n * A + a];
            qq[a] = q[n * A + a];
            lg[a] = legal[n * A + a];
        }
        const float m = mask[n];
        const float nerd = nerd_row<A>(l, p, qq, lg, clip, thr, g);
        part = -(double)(nerd * m);  // :429
        if (dlogit) {
#pragma unroll
            for (int a = 0; a < A; ++a) {
                const float gg = weight * (-(m * g[a]) / nf);
                dlogit[n * A + a] = ACC ? dlogit[n * A + a] + gg : gg;
            }
        }
    }
    if (loss) block_atomic_add(part / (double)nf, loss);
}

// ---------------------------------------------------------------------------------------- per-row gradient sums
// dlogit_tab[row] = sum of dlogit over the slots of that row, dv_tab likewise (row = player * S + state), REPRODUCIBLY: the
// addends are converted to 64-bit fixed point -- one scale per output column, 2^40 / 2^ceil(log2(max |g|)) with the maximum
// taken over all slots by the gather kernel -- and integer addition is associative, so the atomics may land in any order.
// An addend is at most 2^40 units and a row has at most 2^21 slots here (B <= 2^31 / T), so the sums stay below 2^62; the
// rounding of an addend is at most 2^-41 of the column maximum, far below fp32 resolution of the sums.
//
// Rows near the root are hit by the whole batch (every lane passes through the root, a ninth of them through each of its
// children, ...): atomics on them serialise in the memory system (measured with fp64 atomics: 7.5 ns per same-line atomic,
// 62 ms per update if every slot issued its own).  The states of the top levels -- whole levels, top down, as many as fit
// 96 KiB of LDS (tree->n_hot) -- are therefore summed per block in an LDS table by persistent blocks and flushed once per
// block; the remaining rows are deep in the tree, hit by few slots each, and take global atomics directly (25 G/s spread).
constexpr int kFixedBits = 40;

__device__ __forceinline__ double fixed_scale(uint32_t max_bits) {  // 2^kFixedBits / 2^ceil(log2 max), 1 if the column is all zero
    if (max_bits == 0) return 1.0;
    const int e = (int)((max_bits >> 23) & 0xff) - 127 + 1;  // |g| < 2^e for every addend (denormal maxima: e = -126, still an upper bound)
    return ldexp(1.0, kFixedBits - e);
}

template <int A>
__global__ __launch_bounds__(kThreads) void k_row_sums(int T, int64_t B, int64_t S, const int32_t *__restrict__ indices,
                                                       const float *__restrict__ dlogit, const float *__restrict__ dv,
                                                       const uint32_t *__restrict__ gmax, const int32_t *__restrict__ hot_slot,
                                                       const int32_t *__restrict__ hot_state, int n_hot,
                                                       unsigned long long *__restrict__ acc) {
    extern __shared__ unsigned long long hot[];  // [2][n_hot][A + 1]
    for (int i = threadIdx.x; i < 2 * n_hot * (A + 1); i += kThreads) hot[i] = 0ull;
    __syncthreads();
    double scale[A + 1];
#pragma unroll
    for (int a = 0; a <= A; ++a) scale[a] = fixed_scale(gmax[a]);
    const int64_t N = (int64_t)T * B;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < N; i += (int64_t)gridDim.x * kThreads) {
        const int state = indices[i];
        if (state == 0) continue;  // invalid slots carry zero gradients (rnad.py:369)
        const int P = (int)((i / B) & 1);
        const int hs = hot_slot[state];
        unsigned long long *dst = hs >= 0 ? hot + ((int64_t)P * n_hot + hs) * (A + 1) : acc + ((int64_t)P * S + state) * (A + 1);
        long long q[A + 1];
#pragma unroll
        for (int a = 0; a < A; ++a) q[a] = __double2ll_rn((double)dlogit[i * A + a] * scale[a]);
        q[A] = __double2ll_rn((double)dv[i] * scale[A]);
        // a full wave on one row (the first two steps: every lane is at the root): add up in registers, one lane updates the table
        // -- integer sums, so still order-independent
        const bool full_wave = __ballot(1) == ~0ull;
        if (full_wave && __all((int)(dst == (unsigned long long *)__shfl((long long)dst, 0, 64)))) {
#pragma unroll
            for (int a = 0; a <= A; ++a) {
                long long v = q[a];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                q[a] = v;
            }
            if ((threadIdx.x & 63) == 0) {
#pragma unroll
                for (int a = 0; a <= A; ++a) atomicAdd(dst + a, (unsigned long long)q[a]);
            }
            continue;
        }
#pragma unroll
        for (int a = 0; a <= A; ++a) atomicAdd(dst + a, (unsigned long long)q[a]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * n_hot; i += kThreads) {
        const int64_t row = (int64_t)(i / n_hot) * S + hot_state[i % n_hot];
#pragma unroll
        for (int a = 0; a <= A; ++a) {
            const unsigned long long x = hot[(int64_t)i * (A + 1) + a];
            if (x != 0ull) atomicAdd(acc + row * (A + 1) + a, x);
        }
    }
}

// fixed point -> fp32 tables
template <int A>
__global__ __launch_bounds__(kThreads) void k_tab_finish(int64_t rows, const unsigned long long *__restrict__ acc,
                                                         const uint32_t *__restrict__ gmax, float *__restrict__ dlogit,
                                                         float *__restrict__ dv) {
    const int64_t r = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (r >= rows) return;
#pragma unroll
    for (int a = 0; a < A; ++a) dlogit[r * A + a] = (float)((double)(long long)acc[r * (A + 1) + a] / fixed_scale(gmax[a]));
    dv[r] = (float)((double)(long long)acc[r * (A + 1) + A] / fixed_scale(gmax[A]));
}

template <int A>
__global__ __launch_bounds__(kThreads) void k_pack_records(int64_t rows, const float *__restrict__ logit, const float *__restrict__ v,
                                                           const float *__restrict__ vt, const float *__restrict__ lr,
                                                           const float *__restrict__ lr2, float *__restrict__ rec) {
    const int64_t r = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (r >= rows) return;
    float *o = rec + r * kRecStride<A>;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        o[a] = logit[r * A + a];
        o[A + 2 + a] = lr[r * A + a];
        o[2 * A + 2 + a] = lr2[r * A + a];
    }
    o[A] = v[r];
    o[A + 1] = vt[r];
#pragma unroll
    for (int j = 3 * A + 2; j < kRecStride<A>; ++j) o[j] = 0.0f;
}

// The fused learner pass (header comment of rnad_learn_fused).  Per (t, b): 69 B read, 16 B written (A = 3).
// TAB = false: logit_ / v_ / vtn_ / lreg_ / lreg2_ are per-slot arrays [T,B,(A)] and dlogit / dv are written per slot.
// TAB = true ("tabular" evaluation): the observation is a function of (state, player to move) alone, so the nets were evaluated
// once per (player, state), row = player * S + state, and every slot gathers its row: logit_ then points to ONE table of
// records [2S][kRecStride] = lg[A] | v | v_target | lr[A] | lr2[A] | pad (k_pack_records; one 48-byte gather per slot at
// A = 3 instead of five scattered ones), the other four pointers are unused.  dlogit / dv are written per slot either way.
template <int A, bool TAB>
__global__ __launch_bounds__(kThreads) void k_learn_fused(int T, int64_t B, const int32_t *__restrict__ indices,
                                                          const uint8_t *__restrict__ mbits, const int32_t *__restrict__ actions,
                                                          const float *__restrict__ rewards, const float *__restrict__ mu_,
                                                          const float *__restrict__ logit_, const float *__restrict__ v_,
                                                          const float *__restrict__ vtn_, const float *__restrict__ lreg_,
                                                          const float *__restrict__ lreg2_, const double *__restrict__ norm,
                                                          rnad_learn_params_t hp, double *__restrict__ losses,
                                                          float *__restrict__ dlogit, float *__restrict__ dv,
                                                          float *__restrict__ pi_out, float *__restrict__ vt_out,
                                                          float *__restrict__ q_out, int64_t S, uint32_t *__restrict__ gmax) {
    // gmax (TAB, optional): [A + 1] running maxima of |dL/dlogit[:, a]|, |dL/dv| over all slots, as float bit patterns (the
    // fixed-point scales of k_row_sums); the caller zeroes it.
    uint32_t mx[A + 1];
#pragma unroll
    for (int a = 0; a <= A; ++a) mx[a] = 0u;
    double part_v = 0.0, part_n = 0.0;
    const float nf0 = norm_of(norm), nf1 = norm_of(norm + 1);
    const VtHp vh{-hp.eta, hp.lambda_, hp.c, hp.rho, hp.gamma};
    for (int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x; b < B; b += (int64_t)gridDim.x * kThreads) {
        Carry cy[2];
        for (int t = T - 1; t >= 0; --t) {
            const int64_t i = (int64_t)t * B + b;
            const int state = indices[i];
            const bool valid = state != 0;  // rnad.py:369
            const float valid_f = valid ? 1.0f : 0.0f;
            const int P = t & 1;  // turns[t, :] (episode.py:96-98)
            const int64_t row = TAB ? (int64_t)P * S + state : i;  // where this slot's net outputs live
            const uint32_t bits = mbits[i];
            const int act = actions[i];
            float mu[A], lg[A], lr[A], lr2[A], legal[A], oh[A];
            float rec[kRecStride<A>];  // TAB: this row's record lg[A] | v | v_target | lr[A] | lr2[A], fetched as 16-byte pieces
            if (TAB) {
                const float4 *rp = reinterpret_cast<const float4 *>(logit_ + row * kRecStride<A>);
#pragma unroll
                for (int j = 0; j < kRecStride<A> / 4; ++j) {
                    const float4 r4 = rp[j];
                    rec[4 * j] = r4.x; rec[4 * j + 1] = r4.y; rec[4 * j + 2] = r4.z; rec[4 * j + 3] = r4.w;
                }
            }
#pragma unroll
            for (int a = 0; a < A; ++a) {
                mu[a] = mu_[i * A + a];
                lg[a] = TAB ? rec[a] : logit_[row * A + a];
                lr[a] = TAB ? rec[A + 2 + a] : lreg_[row * A + a];
                lr2[a] = TAB ? rec[2 * A + 2 + a] : lreg2_[row * A + a];
                legal[a] = (float)((bits >> a) & 1);
                oh[a] = act == a ? 1.0f : 0.0f;
            }
            float pi[A], lp[A], lpr[A], lpr2[A], pip[A], lpol[A];
            policy_head<A>(lg, bits, pi, lp);             // net.forward_batch of the learner (rnad.py:373)
            log_policy_only<A>(lr, bits, lpr);            // net_reg (rnad.py:379)
            log_policy_only<A>(lr2, bits, lpr2);          // net_reg_ (rnad.py:380)
            process_policy_row<A>(pi, legal, hp.n_disc, hp.eps_threshold, pip);  // rnad.py:374
#pragma unroll
            for (int a = 0; a < A; ++a) lpol[a] = lp[a] - (hp.alpha * lpr[a] + hp.one_minus_alpha * lpr2[a]);  // rnad.py:382
            const float rew = rewards[i];
            const float vtn = TAB ? rec[A + 1] : vtn_[row];
            float vt[2], q[2][A];
            vtrace_step<A>(cy[0], vh, valid, P == 0, valid_f, vtn, rew, mu, pip, lpol, oh, vt[0], q[0]);   // player 0 (rnad.py:384-406)
            vtrace_step<A>(cy[1], vh, valid, P == 1, valid_f, vtn, -rew, mu, pip, lpol, oh, vt[1], q[1]);  // player 1: rewards = -r (:368)
            // losses: only player P has a non-zero mask at this step (has_played_P = valid && turn == P)
            float g_v = 0.0f, g_l[A];
#pragma unroll
            for (int a = 0; a < A; ++a) g_l[a] = 0.0f;
            if (valid) {
                const float nfp = P ? nf1 : nf0;
                const float vv = TAB ? rec[A] : v_[row];
                const float vtp = P ? vt[1] : vt[0];
                const float d = vv - vtp;
                part_v += (double)(d * d) / (double)nfp;
                g_v = hp.w_v * (2.0f * d / nfp);
                float qp[A], g[A];
#pragma unroll
                for (int a = 0; a < A; ++a) qp[a] = P ? q[1][a] : q[0][a];
                const float nerd = nerd_row<A>(lg, pip, qp, legal, hp.clip, hp.threshold, g);
                part_n += -(double)nerd / (double)nfp;
#pragma unroll
                for (int a = 0; a < A; ++a) g_l[a] = hp.w_n * (-g[a] / nfp);
            }
            if (TAB && gmax) {
#pragma unroll
                for (int a = 0; a < A; ++a) mx[a] = max(mx[a], __float_as_uint(fabsf(g_l[a])));
                mx[A] = max(mx[A], __float_as_uint(fabsf(g_v)));
            }
            dv[i] = g_v;
#pragma unroll
            for (int a = 0; a < A; ++a) dlogit[i * A + a] = g_l[a];
            if (pi_out) {
#pragma unroll
                for (int a = 0; a < A; ++a) pi_out[i * A + a] = pi[a];
            }
            if (vt_out) {
                vt_out[i] = vt[0];
                vt_out[(int64_t)T * B + i] = vt[1];
            }
            if (q_out) {
#pragma unroll
                for (int a = 0; a < A; ++a) {
                    q_out[i * A + a] = q[0][a];
                    q_out[((int64_t)T * B + i) * A + a] = q[1][a];
                }
            }
        }
    }
    if (TAB && gmax) {
#pragma unroll
        for (int a = 0; a <= A; ++a) {
            uint32_t m = mx[a];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
            // one lane per wave, and only if it would raise the maximum (16 K same-line atomics cost 0.5 ms otherwise)
            if ((threadIdx.x & 63) == 0 && m > __atomic_load_n(gmax + a, __ATOMIC_RELAXED)) atomicMax(gmax + a, m);
        }
    }
    if (losses) {
        block_atomic_add(part_v, losses);
        __syncthreads();
        block_atomic_add(part_n, losses + 1);
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------- entry points
extern "C" int rnad_process_policy(int64_t N, int A, const float *policy, const float *mask, int n_disc, float eps, float *out,
                                   void *stream) {
    RNAD_REQUIRE(policy && mask && out, "rnad_process_policy: null argument");
    RNAD_REQUIRE(n_disc >= 1, "rnad_process_policy: n_disc must be positive");
    if (N == 0) return 0;
    RNAD_DISPATCH_A(A, hipLaunchKernelGGL((k_process_policy<kA>), dim3(blocks_for(N)), dim3(kThreads), 0, (hipStream_t)stream, N, policy,
                                          mask, n_disc, eps, out));
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int rnad_vtrace(int T, int64_t B, int A, const float *v, const float *valid, const int32_t *player_id, const float *mu,
                           const float *pi, const float *logpi, const void *actions, int actions_onehot, const float *reward,
                           int player, float eta, float lambda_, float c, float rho, float gamma, float *v_target,
                           int32_t *has_played, float *q, void *stream) {
    RNAD_REQUIRE(v && valid && mu && pi && logpi && actions && reward && v_target && q, "rnad_vtrace: null argument");
    RNAD_REQUIRE(T >= 0 && B >= 0, "rnad_vtrace: negative shape");
    if (T == 0 || B == 0) return 0;
    const VtHp hp{-eta, lambda_, c, rho, gamma};
    RNAD_DISPATCH_A(A, {
        if (actions_onehot)
            hipLaunchKernelGGL((k_vtrace<kA, true>), dim3(blocks_for(B)), dim3(kThreads), 0, (hipStream_t)stream, T, B, v, valid, player_id,
                               mu, pi, logpi, actions, reward, player, hp, v_target, has_played, q);
        else
            hipLaunchKernelGGL((k_vtrace<kA, false>), dim3(blocks_for(B)), dim3(kThreads), 0, (hipStream_t)stream, T, B, v, valid, player_id,
                               mu, pi, logpi, actions, reward, player, hp, v_target, has_played, q);
    });
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int rnad_mask_sum(int64_t N, const float *mask, double *out, void *stream) {
    RNAD_REQUIRE(mask && out, "rnad_mask_sum: null argument");
    if (int rc = zero_async(out, sizeof(double), (hipStream_t)stream)) return rc;
    if (N == 0) return 0;
    const unsigned grid = (unsigned)std::min<int64_t>(blocks_for(N), 2048);
    hipLaunchKernelGGL(k_mask_sum, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, N, mask, out);
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int rnad_loss_v(int64_t N, const float *v, const float *v_target, const float *mask, const double *norm, float weight,
                           double *loss, float *dv, int accumulate, void *stream) {
    RNAD_REQUIRE(v && v_target && mask && norm, "rnad_loss_v: null argument");
    if (N == 0) return 0;
    if (accumulate)
        hipLaunchKernelGGL((k_loss_v<true>), dim3(blocks_for(N)), dim3(kThreads), 0, (hipStream_t)stream, N, v, v_target, mask, norm, weight,
                           loss, dv);
    else
        hipLaunchKernelGGL((k_loss_v<false>), dim3(blocks_for(N)), dim3(kThreads), 0, (hipStream_t)stream, N, v, v_target, mask, norm, weight,
                           loss, dv);
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int rnad_loss_nerd(int64_t N, int A, const float *logit, const float *pi, const float *q, const float *mask,
                              const float *legal, const double *norm, float clip, float threshold, float weight, double *loss,
                              float *dlogit, int accumulate, void *stream) {
    RNAD_REQUIRE(logit && pi && q && mask && legal && norm, "rnad_loss_nerd: null argument");
    if (N == 0) return 0;
    RNAD_DISPATCH_A(A, {
        if (accumulate)
            hipLaunchKernelGGL((k_loss_nerd<kA, true>), dim3(blocks_for(N)), dim3(kThreads), 0, (hipStream_t)stream, N, logit, pi, q, mask,
                               legal, norm, clip, threshold, weight, loss, dlogit);
        else
            hipLaunchKernelGGL((k_loss_nerd<kA, false>), dim3(blocks_for(N)), dim3(kThreads), 0, (hipStream_t)stream, N, logit, pi, q, mask,
                               legal, norm, clip, threshold, weight, loss, dlogit);
    });
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int rnad_learn_fused(int T, int64_t B, int A, const int32_t *indices, const uint8_t *mask_bits, const int32_t *actions,
                                const float *rewards, const float *mu, const float *logit, const float *v, const float *v_target_net,
                                const float *logit_reg, const float *logit_reg_, const double *norm, const rnad_learn_params_t *hp,
                                double *losses, float *dlogit, float *dv, float *pi_out, float *v_target_out, float *q_out,
                                void *stream_) {
    RNAD_REQUIRE(indices && mask_bits && actions && rewards && mu && logit && v && v_target_net && logit_reg && logit_reg_ && norm &&
                     hp && dlogit && dv,
                 "rnad_learn_fused: null argument");
    RNAD_REQUIRE(T >= 0 && B >= 0, "rnad_learn_fused: negative shape");
    RNAD_REQUIRE(hp->n_disc >= 1, "rnad_learn_fused: n_disc must be positive");
    hipStream_t stream = (hipStream_t)stream_;
    if (losses) {
        if (int rc = zero_async(losses, 2 * sizeof(double), stream)) return rc;
    }
    if (T == 0 || B == 0) return 0;
    ProfScope prof(PROF_LEARN, stream);
    RNAD_DISPATCH_A(A, hipLaunchKernelGGL((k_learn_fused<kA, false>), dim3(blocks_for(B)), dim3(kThreads), 0, stream, T, B, indices,
                                          mask_bits, actions, rewards, mu, logit, v, v_target_net, logit_reg, logit_reg_, norm, *hp, losses,
                                          dlogit, dv, pi_out, v_target_out, q_out, (int64_t)0, (uint32_t *)nullptr));
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

static int learn_fused_gather_impl(const rnad_tree_t *tree, int T, int64_t B, const int32_t *indices, const uint8_t *mask_bits,
                                   const int32_t *actions, const float *rewards, const float *mu, const float *logit_tab,
                                   const float *v_tab, const float *v_target_tab, const float *logit_reg_tab, const float *logit_reg_tab_,
                                   const double *norm, const rnad_learn_params_t *hp, double *losses, float *dlogit, float *dv,
                                   uint32_t *gmax, float *rec, hipStream_t stream) {
    RNAD_REQUIRE(tree && indices && mask_bits && actions && rewards && mu && logit_tab && v_tab && v_target_tab && logit_reg_tab &&
                     logit_reg_tab_ && norm && hp && dlogit && dv && rec,
                 "rnad_learn_fused_gather: null argument");
    RNAD_REQUIRE(T >= 0 && B >= 0, "rnad_learn_fused_gather: negative shape");
    RNAD_REQUIRE(hp->n_disc >= 1, "rnad_learn_fused_gather: n_disc must be positive");
    if (losses) {
        if (int rc = zero_async(losses, 2 * sizeof(double), stream)) return rc;
    }
    if (gmax) {
        if (int rc = zero_async(gmax, sizeof(uint32_t) * (tree->A + 1), stream)) return rc;
    }
    if (T == 0 || B == 0) return 0;
    ProfScope prof(PROF_LEARN, stream);
    RNAD_DISPATCH_A(tree->A, hipLaunchKernelGGL((k_pack_records<kA>), dim3(blocks_for(2 * tree->S)), dim3(kThreads), 0, stream, 2 * tree->S,
                                                logit_tab, v_tab, v_target_tab, logit_reg_tab, logit_reg_tab_, rec));
    RNAD_DISPATCH_A(tree->A, hipLaunchKernelGGL((k_learn_fused<kA, true>), dim3(blocks_for(B)), dim3(kThreads), 0, stream, T, B, indices,
                                                mask_bits, actions, rewards, mu, (const float *)rec, (const float *)nullptr,
                                                (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, norm, *hp, losses,
                                                dlogit, dv, (float *)nullptr, (float *)nullptr, (float *)nullptr, tree->S, gmax));
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------- rnad_row_sums (stand-alone)
namespace {
// column maxima of |dlogit[:, a]| and |dv| over the valid slots, as float bit patterns (the fixed-point scales of k_row_sums)
template <int A>
__global__ __launch_bounds__(kThreads) void k_grad_maxima(int64_t N, const int32_t *__restrict__ indices, const float *__restrict__ dlogit,
                                                          const float *__restrict__ dv, uint32_t *__restrict__ gmax) {
    uint32_t mx[A + 1];
#pragma unroll
    for (int a = 0; a <= A; ++a) mx[a] = 0u;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < N; i += (int64_t)gridDim.x * kThreads) {
        if (indices[i] == 0) continue;
#pragma unroll
        for (int a = 0; a < A; ++a) mx[a] = max(mx[a], __float_as_uint(fabsf(dlogit[i * A + a])));
        mx[A] = max(mx[A], __float_as_uint(fabsf(dv[i])));
    }
#pragma unroll
    for (int a = 0; a <= A; ++a) {
        uint32_t m = mx[a];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
        if ((threadIdx.x & 63) == 0 && m > __atomic_load_n(gmax + a, __ATOMIC_RELAXED)) atomicMax(gmax + a, m);
    }
}
}  // namespace

static int row_sums_launch(const rnad_tree_t *tree, int T, int64_t B, const int32_t *indices, const float *dlogit, const float *dv,
                           const uint32_t *gmax, unsigned long long *acc, float *dlogit_tab, float *dv_tab, hipStream_t stream) {
    const int A = tree->A;
    const int64_t S = tree->S, N = (int64_t)T * B;
    if (int rc = zero_async(acc, sizeof(unsigned long long) * 2 * S * (A + 1), stream)) return rc;
    if (N > 0) {
        // persistent blocks: the LDS table of hot rows is flushed once per block, so few blocks walking many slots each
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, tree->device);
        const size_t lds = (size_t)2 * tree->n_hot * (A + 1) * sizeof(unsigned long long);
        const int per_cu = std::max(1, std::min(8, (int)((160 * 1024) / std::max<size_t>(lds, 1))));
        const unsigned grid = (unsigned)std::min<int64_t>(blocks_for(N), (int64_t)cus * per_cu);
#define RNAD_ROWSUM_LAUNCH()                                                                                                       \
    do {                                                                                                                           \
        auto kern = k_row_sums<kA>;                                                                                                \
        if (lds > 64 * 1024) RNAD_HIP_OK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, stream, T, B, S, indices, dlogit, dv, gmax,                      \
                           (const int32_t *)tree->hot_slot, (const int32_t *)tree->level_order, tree->n_hot, acc);                 \
    } while (0)
        RNAD_DISPATCH_A(A, RNAD_ROWSUM_LAUNCH());
#undef RNAD_ROWSUM_LAUNCH
    }
    RNAD_DISPATCH_A(A, hipLaunchKernelGGL((k_tab_finish<kA>), dim3(blocks_for(2 * S)), dim3(kThreads), 0, stream, 2 * S,
                                          (const unsigned long long *)acc, gmax, dlogit_tab, dv_tab));
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

// Per-row sums of given per-slot gradients (what rnad_learn_fused_tabular does after its gather pass), for callers that computed
// dL/dlogit [T,B,A], dL/dv [T,B] themselves -- e.g. autograd through a table gather.  Slots with indices == 0 are skipped.
// workspace: rnad_row_sums_workspace(tree) bytes, 8-byte aligned.  B <= 2^21.
extern "C" int64_t rnad_row_sums_workspace(const rnad_tree_t *tree) {
    if (!tree) return -1;
    return 2 * tree->S * (tree->A + 1) * 8 + 64;
}

extern "C" int rnad_row_sums(const rnad_tree_t *tree, int T, int64_t B, const int32_t *indices, const float *dlogit, const float *dv,
                             void *workspace, float *dlogit_tab, float *dv_tab, void *stream_) {
    RNAD_REQUIRE(tree && indices && dlogit && dv && workspace && dlogit_tab && dv_tab, "rnad_row_sums: null argument");
    RNAD_REQUIRE(T >= 0 && B >= 0 && B <= ((int64_t)1 << 21), "rnad_row_sums: bad shape (at most 2^21 lanes per call)");
    hipStream_t stream = (hipStream_t)stream_;
    const int A = tree->A;
    const int64_t N = (int64_t)T * B;
    unsigned long long *acc = (unsigned long long *)workspace;
    uint32_t *gmax = (uint32_t *)((char *)workspace + 2 * tree->S * (A + 1) * 8);
    if (int rc = zero_async(gmax, sizeof(uint32_t) * (A + 1), stream)) return rc;
    if (N > 0) {
        const unsigned grid = (unsigned)std::min<int64_t>(blocks_for(N), 2048);
        RNAD_DISPATCH_A(A, hipLaunchKernelGGL((k_grad_maxima<kA>), dim3(grid), dim3(kThreads), 0, stream, N, indices, dlogit, dv, gmax));
    }
    return row_sums_launch(tree, T, B, indices, dlogit, dv, gmax, acc, dlogit_tab, dv_tab, stream);
}

// Tables in, per-slot gradients out: the forward evaluations are deduplicated (2S rows instead of T*B slots), the backward is
// not -- dlogit [T,B,A] and dv [T,B] are the bits rnad_learn_fused produces from per-slot net outputs, so a per-slot
// rnad_mlp_backward gives bit-identical weight gradients.
extern "C" int rnad_learn_fused_gather(const rnad_tree_t *tree, int T, int64_t B, const int32_t *indices, const uint8_t *mask_bits,
                                       const int32_t *actions, const float *rewards, const float *mu, const float *logit_tab,
                                       const float *v_tab, const float *v_target_tab, const float *logit_reg_tab,
                                       const float *logit_reg_tab_, const double *norm, const rnad_learn_params_t *hp, double *losses,
                                       void *workspace, float *dlogit, float *dv, void *stream) {
    return learn_fused_gather_impl(tree, T, B, indices, mask_bits, actions, rewards, mu, logit_tab, v_tab, v_target_tab, logit_reg_tab,
                                   logit_reg_tab_, norm, hp, losses, dlogit, dv, nullptr, (float *)workspace, (hipStream_t)stream);
}

extern "C" int64_t rnad_learn_gather_workspace(const rnad_tree_t *tree) {  // bytes: the [2S] record table
    if (!tree) return -1;
    return 2 * tree->S * (int64_t)((3 * tree->A + 2 + 3) & ~3) * 4;
}

// ... and the per-row sums of those per-slot gradients (k_row_sums), for ONE backward over the 2S distinct observations.
// workspace: rnad_learn_tabular_workspace(tree, T, B) bytes = per-slot dlogit / dv + the fixed-point table + the column maxima.
extern "C" int64_t rnad_learn_tabular_workspace(const rnad_tree_t *tree, int T, int64_t B) {
    if (!tree || T < 0 || B < 0) return -1;
    const int64_t A = tree->A, N = (int64_t)T * B;
    return N * (A + 1) * 4 + 2 * tree->S * (A + 1) * 8 + 64 + 16 + rnad_learn_gather_workspace(tree);
}

extern "C" int rnad_learn_fused_tabular(const rnad_tree_t *tree, int T, int64_t B, const int32_t *indices, const uint8_t *mask_bits,
                                        const int32_t *actions, const float *rewards, const float *mu, const float *logit_tab,
                                        const float *v_tab, const float *v_target_tab, const float *logit_reg_tab,
                                        const float *logit_reg_tab_, const double *norm, const rnad_learn_params_t *hp, double *losses,
                                        void *workspace, float *dlogit_tab, float *dv_tab, void *stream_) {
    RNAD_REQUIRE(tree && workspace && dlogit_tab && dv_tab, "rnad_learn_fused_tabular: null argument");
    RNAD_REQUIRE(T >= 0 && B >= 0 && (int64_t)T * B < ((int64_t)1 << 40), "rnad_learn_fused_tabular: bad shape");
    RNAD_REQUIRE(B <= ((int64_t)1 << 21), "rnad_learn_fused_tabular: more than 2^21 lanes per call overflow the fixed-point row sums");
    hipStream_t stream = (hipStream_t)stream_;
    const int A = tree->A;
    const int64_t S = tree->S, N = (int64_t)T * B;
    char *ws = (char *)workspace;
    unsigned long long *acc = (unsigned long long *)ws;                       // [2S][A + 1], 8-byte aligned at the front
    uint32_t *gmax = (uint32_t *)(ws + 2 * S * (A + 1) * 8);                  // [A + 1]
    float *dlogit = (float *)(ws + 2 * S * (A + 1) * 8 + 64);                 // [T,B,A]
    float *dv = dlogit + N * A;                                               // [T,B]
    float *rec = (float *)(((uintptr_t)(dv + N) + 15) & ~(uintptr_t)15);          // [2S][record], 16-byte aligned
    if (int rc = learn_fused_gather_impl(tree, T, B, indices, mask_bits, actions, rewards, mu, logit_tab, v_tab, v_target_tab,
                                         logit_reg_tab, logit_reg_tab_, norm, hp, losses, dlogit, dv, gmax, rec, stream))
        return rc;
    (void)S;
    return row_sums_launch(tree, T, B, indices, dlogit, dv, gmax, acc, dlogit_tab, dv_tab, stream);
}

// ---------------------------------------------------------------------------------------- gradient clipping
// torch.nn.utils.clip_grad_norm_ (learn/rnad.py:456) over ONE flat gradient bucket: total 2-norm, then
// g *= min(max_norm / (norm + 1e-6), 1).  One block (the bucket is 10 756 floats on configs[1]) instead of ~8 small torch
// launches per update; the norm is accumulated in fp64 in a fixed order.  total_norm (optional) receives the norm.
namespace {
__global__ __launch_bounds__(1024) void k_clip_grad_norm(int64_t n, float *__restrict__ g, float max_norm, float *__restrict__ total_norm) {
    __shared__ double part[16];
    __shared__ float coef_s;
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s += (double)g[i] * (double)g[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += part[i];
        const float norm = (float)sqrt(t);
        if (total_norm) *total_norm = norm;
        const float c = max_norm / (norm + 1e-6f);
        coef_s = c < 1.0f ? c : 1.0f;
    }
    __syncthreads();
    const float c = coef_s;
    for (int64_t i = threadIdx.x; i < n; i += 1024) g[i] *= c;  // torch multiplies by the clamped coefficient unconditionally
}
}  // namespace

extern "C" int rnad_clip_grad_norm(int64_t n, float *grads, float max_norm, float *total_norm, void *stream) {
    RNAD_REQUIRE(n >= 0 && (n == 0 || grads), "rnad_clip_grad_norm: null argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_clip_grad_norm, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, grads, max_norm, total_norm);
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}
