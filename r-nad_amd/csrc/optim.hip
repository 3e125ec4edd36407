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
#include "common.hpp"

#include <cmath>

using namespace rnad;

namespace {

constexpr int kOptThreads = 512;
constexpr int kMaxTensors = 8;

struct OptTensors {
    int n;
    int64_t offset[kMaxTensors + 1];  // of each tensor inside the flat gradient bucket
    float *param[kMaxTensors], *exp_avg[kMaxTensors], *exp_avg_sq[kMaxTensors], *step[kMaxTensors], *target[kMaxTensors];
};

// ONE workgroup (the bucket is 43 KB), everything latency: the loads of kBatch strided elements per thread are issued together
// (one memory round trip per batch instead of one per element), the per-tensor pointers sit in LDS (indexed per lane without
// waterfall loops), the Adam scalars are computed once per tensor.  A single workgroup needs no inter-workgroup ordering for the
// in-place clip and the step counters.
constexpr int kBatch = 8;  // 512 threads x 8: registers only (a kernel that needs scratch memory is best kept out of captured graphs)

__global__ __launch_bounds__(kOptThreads) void k_optimizer_step(OptTensors ts, float *__restrict__ grads, rnad_adam_params_t hp,
                                                                float *__restrict__ total_norm) {
    __shared__ double part[kOptThreads / 64];
    __shared__ float coef_s, step_s[kMaxTensors], step_size_s[kMaxTensors], bc2s_s[kMaxTensors];
    __shared__ float *ptr_s[4][kMaxTensors];
    __shared__ int64_t off_s[kMaxTensors + 1];
    const int64_t n = ts.offset[ts.n];
    if ((int)threadIdx.x < kMaxTensors) {
        const int k = threadIdx.x;
        const bool on = k < ts.n;
        ptr_s[0][k] = on ? ts.param[k] : nullptr;
        ptr_s[1][k] = on ? ts.exp_avg[k] : nullptr;
        ptr_s[2][k] = on ? ts.exp_avg_sq[k] : nullptr;
        ptr_s[3][k] = on ? ts.target[k] : nullptr;
        off_s[k] = on ? ts.offset[k] : n;
        if (k == 0) off_s[kMaxTensors] = n;
        if (on) {  // every tensor has its own step counter in torch's state; they move together
            const float step = *ts.step[k] + 1.0f;
            const double bc1 = 1.0 - pow((double)hp.beta1, (double)step);
            step_s[k] = step;
            step_size_s[k] = (float)((double)hp.lr / bc1);
            bc2s_s[k] = (float)sqrt(1.0 - pow((double)hp.beta2, (double)step));
        }
    }
    double s = 0.0;
    for (int64_t base = threadIdx.x; base < n; base += (int64_t)kOptThreads * kBatch) {
        float g[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int64_t i = base + (int64_t)u * kOptThreads;
            g[u] = i < n ? grads[i] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) s += (double)g[u] * (double)g[u];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < kOptThreads / 64; ++i) t += part[i];
        const float norm = (float)sqrt(t);
        if (total_norm) *total_norm = norm;
        const float c = hp.max_norm / (norm + 1e-6f);
        coef_s = c < 1.0f ? c : 1.0f;  // torch multiplies by the clamped coefficient unconditionally
    }
    __syncthreads();
    const float coef = coef_s, w = 1.0f - hp.beta1;
    for (int64_t base = threadIdx.x; base < n; base += (int64_t)kOptThreads * kBatch) {
        int k[kBatch];
        int32_t e[kBatch];
        float g[kBatch], m[kBatch], v[kBatch], p[kBatch], tg[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {  // all loads of the batch first
            const int64_t i = base + (int64_t)u * kOptThreads;
            const bool on = i < n;
            int kk = 0;
#pragma unroll
            for (int q = 1; q < kMaxTensors; ++q) kk += (on && i >= off_s[q]) ? 1 : 0;
            k[u] = on ? kk : -1;
            e[u] = on ? (int32_t)(i - off_s[kk]) : 0;
            g[u] = on ? grads[i] : 0.0f;
            m[u] = on ? ptr_s[1][kk][e[u]] : 0.0f;
            v[u] = on ? ptr_s[2][kk][e[u]] : 0.0f;
            p[u] = on ? ptr_s[0][kk][e[u]] : 0.0f;
            tg[u] = (on && ptr_s[3][kk]) ? ptr_s[3][kk][e[u]] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            if (k[u] < 0) continue;
            const int kk = k[u];
            const float grad = g[u] * coef;
            grads[base + (int64_t)u * kOptThreads] = grad;
            const float m_new = w < 0.5f ? m[u] + w * (grad - m[u]) : grad - (grad - m[u]) * (1.0f - w);  // at::lerp
            const float v_new = hp.beta2 * v[u] + (1.0f - hp.beta2) * grad * grad;
            ptr_s[1][kk][e[u]] = m_new;
            ptr_s[2][kk][e[u]] = v_new;
            const float denom = sqrtf(v_new) / bc2s_s[kk] + hp.eps;
            const float p_new = p[u] - step_size_s[kk] * m_new / denom;
            ptr_s[0][kk][e[u]] = p_new;
            if (ptr_s[3][kk]) ptr_s[3][kk][e[u]] = tg[u] * (1.0f - hp.ema) + hp.ema * p_new;
        }
    }
    if ((int)threadIdx.x < ts.n) *ts.step[threadIdx.x] = step_s[threadIdx.x];  // read above by this same thread
}

}  // namespace

extern "C" int rnad_optimizer_step(int n_tensors, const int64_t *sizes, float *const *param, float *grads, float *const *exp_avg,
                                   float *const *exp_avg_sq, float *const *step, float *const *target, const rnad_adam_params_t *hp,
                                   float *total_norm, void *stream) {
    RNAD_REQUIRE(sizes && param && grads && exp_avg && exp_avg_sq && step && hp, "rnad_optimizer_step: null argument");
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
    hipLaunchKernelGGL(k_optimizer_step, dim3(1), dim3(kOptThreads), 0, (hipStream_t)stream, ts, grads, *hp, total_norm);
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}
