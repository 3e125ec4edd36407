// optim.hip -- the tail of a training step in one launch: gradient clipping, Adam, EMA target (gfx950).
//
// Replaces learn/rnad.py:456 (clip_grad_norm_), :514 (optimizer.step() of torch.optim.Adam) and :516-523 (the EMA target update)
// for the policy/value MLP, whose 10 756 parameters make each of those ~8 torch launches pure launch latency (~45 us of a
// 0.5 ms step).  Per element of the flat gradient bucket: 2-norm of the bucket -> clip coefficient -> Adam -> EMA.
// The arithmetic is torch's: clip g *= min(max_norm / (norm + 1e-6), 1); Adam as in torch's fused kernel (no weight decay, no
// amsgrad: what the reference constructs at rnad.py:232-237) with the bias corrections taken in double from the step counter;
// EMA target = target * (1 - gamma) + gamma * param, two roundings like _foreach_mul_ + _foreach_add_(alpha).
// The optimiser state lives in torch.optim.Adam's own tensors (exp_avg, exp_avg_sq, step), so checkpoints keep the reference format.
// Citations are baskuit/R-NaD file:line.
#include "mlp_common.hpp"

#include <cmath>

using namespace rnad;

namespace {

constexpr int kOptThreads = 1024;
constexpr int kMaxTensors = 8;

struct OptTensors {
    int n;
    int64_t offset[kMaxTensors + 1];  // of each tensor inside the flat gradient bucket
    float *param[kMaxTensors], *exp_avg[kMaxTensors], *exp_avg_sq[kMaxTensors], *step[kMaxTensors], *target[kMaxTensors];
};

// Everything here is latency (the bucket is 43 KB).  One element per thread, ceil(n / 1024) workgroups; EVERY workgroup first takes
// the 2-norm of the whole bucket for itself -- the same loads in the same order, hence the same clip coefficient everywhere, and no
// inter-workgroup dependency -- with all of a thread's loads in flight together.  The gradients are read only (the clipped values
// go straight into Adam; nothing reads .grad between clip_grad_norm_ and zero_grad in learn/rnad.py:456-514), so a workgroup may
// still be summing while another one updates.  The per-tensor pointers and Adam scalars sit in LDS (indexed per lane without
// waterfall loops).  The step counters are advanced by whichever workgroup takes the last ticket: by then all have read them.
// The ticket counter is the CALLER's (one zero-initialised word per optimiser: two trainers stepping on two streams of one device
// each count their own workgroups).
constexpr int kNormBatch = 8;

// Where element e of Linear tensor k (MLP_KEYS order: value_fc0.weight, .bias, value_fc1.weight, .bias, policy_fc0.weight, ...) sits in
// the packed LDS image the MLP kernels read (mlp_common.hpp; k_mlp_pack is the forward map).
// fold: the image of the FOLD kernels (rnad_mlp_pack_fold_multi): the expected-value columns in the first-layer region, the raw legal
// columns in their own region behind the output biases.
__device__ __forceinline__ int image_index(int k, int e, int A, int W, bool fold) {
    using namespace rnad_mlp;
    const int OBS = 2 * A * A, K = fold ? ((A * A + 2) & ~1) : OBS, KS = K / 2;
    switch (k) {
        case 0:
        case 4: {
            const int h = e / OBS + (k == 4 ? W : 0), kk = e % OBS;
            if (fold && kk >= A * A) return img_legal(K, W, A) + (kk - A * A) * 2 * W + h;
            return (h / kTile) * (KS * 64) + (kk / 2) * 64 + (kk % 2) * 32 + (h % kTile);
        }
        case 1: return img_b0(K, W) + e;
        case 5: return img_b0(K, W) + W + e;
        case 2: return img_w1v(K, W) + e;
        case 3: return img_b1(K, W, A);
        case 6: return img_w1p(K, W) + e;
        default: return img_b1(K, W, A) + 1 + e;
    }
}

// mlp_A > 0: the n == 8 tensors are the Linear tensors of the fused MLP in MLP_KEYS order, and every updated weight (and EMA target
// weight) is also written into its slot of the packed images packed_param / packed_target -- the images the next step's forward and
// backward kernels read, kept current here instead of by a k_mlp_pack launch per step.
__global__ __launch_bounds__(kOptThreads) void k_optimizer_step(OptTensors ts, const float *__restrict__ grads, rnad_adam_params_t hp,
                                                                float *__restrict__ total_norm, int mlp_A, int mlp_W, int mlp_fold,
                                                                float *__restrict__ packed_param, float *__restrict__ packed_target,
                                                                rnad_step_queue_t *__restrict__ advance, unsigned int *__restrict__ ticket) {
    __shared__ double part[kOptThreads / 64];
    __shared__ float coef_s, step_size_s[kMaxTensors], bc2s_s[kMaxTensors];
    __shared__ float *ptr_s[4][kMaxTensors];
    __shared__ int64_t off_s[kMaxTensors + 1];
    const int64_t n = ts.offset[ts.n];
    // every tensor has its own step counter in torch's state (they move together): requested first, used after the norm
    float step_now = 0.0f;
    if ((int)threadIdx.x < kMaxTensors) {
        const int k = threadIdx.x;
        const bool on = k < ts.n;
        ptr_s[0][k] = on ? ts.param[k] : nullptr;
        ptr_s[1][k] = on ? ts.exp_avg[k] : nullptr;
        ptr_s[2][k] = on ? ts.exp_avg_sq[k] : nullptr;
        ptr_s[3][k] = on ? ts.target[k] : nullptr;
        off_s[k] = on ? ts.offset[k] : n;
        if (k == 0) off_s[kMaxTensors] = n;
        if (on) step_now = *ts.step[k];
    }
    // this thread's own element: requested before the norm so that its latency hides behind it
    const int64_t i = (int64_t)blockIdx.x * kOptThreads + threadIdx.x;
    const float g_own = i < n ? grads[i] : 0.0f;
    double s = 0.0;
    // r06: 16-byte loads where the bucket allows (one round trip for buckets of up to 32 K floats -- configs[3]'s 27 653 took four batches of
    // 4-byte loads); the squares are added in the order of the elements a thread holds either way, so a run is reproducible, and the
    // per-thread partition is part of the kernel's own (documented) summation order
    if ((reinterpret_cast<uintptr_t>(grads) & 15) == 0) {
        const int64_t n4 = n / 4;
        const float4 *g4 = reinterpret_cast<const float4 *>(grads);
        for (int64_t base = threadIdx.x; base < n4; base += (int64_t)kOptThreads * kNormBatch) {
            float4 g[kNormBatch];
#pragma unroll
            for (int u = 0; u < kNormBatch; ++u) {
                const int64_t e = base + (int64_t)u * kOptThreads;
                g[u] = e < n4 ? g4[e] : float4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < kNormBatch; ++u)
                s += ((double)g[u].x * (double)g[u].x + (double)g[u].y * (double)g[u].y) + ((double)g[u].z * (double)g[u].z + (double)g[u].w * (double)g[u].w);
        }
        const int64_t e = 4 * n4 + threadIdx.x;  // (the last n % 4 elements)
        if (e < n) s += (double)grads[e] * (double)grads[e];
    } else
    for (int64_t base = threadIdx.x; base < n; base += (int64_t)kOptThreads * kNormBatch) {
        float g[kNormBatch];
#pragma unroll
        for (int u = 0; u < kNormBatch; ++u) {
            const int64_t e = base + (int64_t)u * kOptThreads;
            g[u] = e < n ? grads[e] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < kNormBatch; ++u) s += (double)g[u] * (double)g[u];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if ((int)threadIdx.x < ts.n) {
        // bias corrections in double from the step counter.  r06: beta^step by squaring (the counter is an integer below 2^24: at most 24
        // squarings and 24 products, a few ulps of a DOUBLE from libm's pow -- invisible once rounded to the floats below) instead of two
        // calls of pow, ~600 fp64 instructions on the launch's critical path (wave 0 also finishes the norm)
        const float step = step_now + 1.0f;
        unsigned e = (unsigned)step;
        double p1 = 1.0, p2 = 1.0, b1 = (double)hp.beta1, b2 = (double)hp.beta2;
        for (; e; e >>= 1) {
            if (e & 1u) { p1 *= b1; p2 *= b2; }
            b1 *= b1;
            b2 *= b2;
        }
        step_size_s[threadIdx.x] = (float)((double)hp.lr / (1.0 - p1));
        bc2s_s[threadIdx.x] = (float)sqrt(1.0 - p2);
    }
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kOptThreads / 64; ++w) t += part[w];
        const float norm = (float)sqrt(t);
        if (total_norm && blockIdx.x == 0) *total_norm = norm;
        const float c = hp.max_norm / (norm + 1e-6f);
        coef_s = c < 1.0f ? c : 1.0f;  // torch multiplies by the clamped coefficient unconditionally
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
            *ticket = 0;
            for (int t = 0; t < ts.n; ++t) *ts.step[t] += 1.0f;
            if (advance) {  // the step is over: the scalars of the next one (every reader of `live` ran in an earlier launch)
                const int64_t c = advance->cursor + 1;
                advance->cursor = c;
                if (c < advance->n) advance->live = advance->ahead[c];
            }
        }
    }
    if (i >= n) return;
    int k = 0;
#pragma unroll
    for (int q = 1; q < kMaxTensors; ++q) k += i >= off_s[q] ? 1 : 0;
    const int32_t e = (int32_t)(i - off_s[k]);
    float *pp = ptr_s[0][k] + e, *pm = ptr_s[1][k] + e, *pv = ptr_s[2][k] + e, *pt = ptr_s[3][k] ? ptr_s[3][k] + e : nullptr;
    const float m = *pm, v = *pv, p = *pp, tg = pt ? *pt : 0.0f;
    const float grad = g_own * coef_s, w = 1.0f - hp.beta1;
    const float m_new = w < 0.5f ? m + w * (grad - m) : grad - (grad - m) * (1.0f - w);  // at::lerp
    const float v_new = hp.beta2 * v + (1.0f - hp.beta2) * grad * grad;
    *pm = m_new;
    *pv = v_new;
    const float denom = sqrtf(v_new) / bc2s_s[k] + hp.eps;
    const float p_new = p - step_size_s[k] * m_new / denom;
    *pp = p_new;
    const float t_new = tg * (1.0f - hp.ema) + hp.ema * p_new;
    if (pt) *pt = t_new;
    if (mlp_A > 0) {
        const int at = image_index(k, e, mlp_A, mlp_W, mlp_fold != 0);
        if (packed_param) packed_param[at] = p_new;
        if (packed_target && pt) packed_target[at] = t_new;
    }
}

}  // namespace

extern "C" int rnad_optimizer_step(int n_tensors, const int64_t *sizes, float *const *param, float *grads, float *const *exp_avg,
                                   float *const *exp_avg_sq, float *const *step, float *const *target, const rnad_adam_params_t *hp,
                                   float *total_norm, int mlp_A, int mlp_W, int mlp_fold, float *packed_param, float *packed_target,
                                   rnad_step_queue_t *advance, uint32_t *ticket, void *stream) {
    RNAD_REQUIRE(sizes && param && grads && exp_avg && exp_avg_sq && step && hp && ticket, "rnad_optimizer_step: null argument");
    if (mlp_A > 0) {
        RNAD_REQUIRE(n_tensors == 8 && mlp_A <= RNAD_MAX_ACTIONS && mlp_W >= rnad_mlp::kTile && mlp_W % rnad_mlp::kTile == 0,
                     "rnad_optimizer_step: packed images go with the 8 Linear tensors of the fused MLP (A=%d, width=%d)", mlp_A, mlp_W);
        const int64_t K = 2 * (int64_t)mlp_A * mlp_A;
        const int64_t want[8] = {mlp_W * K, mlp_W, mlp_W, 1, mlp_W * K, mlp_W, (int64_t)mlp_A * mlp_W, mlp_A};
        for (int k = 0; k < 8; ++k) RNAD_REQUIRE(sizes[k] == want[k], "rnad_optimizer_step: tensor %d does not have the MLP's shape", k);
        RNAD_REQUIRE(!packed_target || target, "rnad_optimizer_step: a packed target image needs the target tensors");
    }
    RNAD_REQUIRE(n_tensors >= 1 && n_tensors <= kMaxTensors, "rnad_optimizer_step: 1..%d tensors", kMaxTensors);
    OptTensors ts{};
    ts.n = n_tensors;
    ts.offset[0] = 0;
    for (int k = 0; k < n_tensors; ++k) {
        RNAD_REQUIRE(sizes[k] >= 0 && param[k] && exp_avg[k] && exp_avg_sq[k] && step[k], "rnad_optimizer_step: null tensor %d", k);
        ts.offset[k + 1] = ts.offset[k] + sizes[k];
        ts.param[k] = param[k]; ts.exp_avg[k] = exp_avg[k]; ts.exp_avg_sq[k] = exp_avg_sq[k]; ts.step[k] = step[k];
        ts.target[k] = target ? target[k] : nullptr;
    }
    const unsigned grid = (unsigned)std::max<int64_t>(1, (ts.offset[n_tensors] + kOptThreads - 1) / kOptThreads);
    RNAD_REQUIRE(!mlp_fold || mlp_A >= 2, "rnad_optimizer_step: the legal fold needs at least two actions");
    hipLaunchKernelGGL(k_optimizer_step, dim3(grid), dim3(kOptThreads), 0, (hipStream_t)stream, ts, (const float *)grads, *hp, total_norm, mlp_A,
                       mlp_W, mlp_fold, mlp_A > 0 ? packed_param : nullptr, mlp_A > 0 ? packed_target : nullptr, advance, (unsigned int *)ticket);
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}
