// mlp.hip -- fused policy/value MLP forward on the fp32 matrix cores (gfx950).
//
// Replaces the tensor program of nn/net.py:40-43 (and :70-73 in forward_batch):
//     value  = value_fc1 (relu(value_fc0 (x)))        x = observation flattened to 2*A*A floats
//     logits = policy_fc1(relu(policy_fc0(x)))
// Citations are baskuit/R-NaD file:line.
//
// Why a kernel: rocprof of the PyTorch-ROCm version (profiles/r01a_*) shows the two hidden activations [N, 256] fp32
// going to HBM and back four times per head (GEMM out, relu in/out, GEMM in): 99.5 % of a training step.  Here the hidden
// layer never leaves the register file.
//
// Mapping (wave64, v_mfma_f32_32x32x2_f32, exact fp32 == an fmaf chain):
//   C[hidden, sample] = W0aug[hidden, k] * Xaug[k, sample]     M = 32 hidden units, N = 32 samples, K = 2 per MFMA
//   A operand  lane l: W0aug[tile*32 + (l & 31)][2*ks + (l >> 5)]   from LDS ([k][2W] layout: conflict-free)
//   B operand  lane l: x[sample0 + (l & 31)][2*ks + (l >> 5)]       one VGPR per k-step, loaded once per 32 samples
//   the first-layer bias is the accumulator's initial value (row K of the LDS image) -> K / 2 MFMAs per tile
//   C layout   lane l holds sample (l & 31) and hidden rows (r & 3) + 8 (r >> 2) + 4 (l >> 5), r in [0, 16):
//              four consecutive hidden units per register quad -> relu, then the second layer as VALU FMAs against
//              float4 reads of W1 from LDS; the two half-waves hold complementary rows and are summed with one DPP add.
// Both heads share the B operand; a wave walks 2 * W / 32 hidden tiles per 32 samples.  Matrix-pipe time per sample
// tile = 2 * (W / 32) * (K / 2) * 64 cycles.
#include "common.hpp"

#include <algorithm>

using namespace rnad;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
#ifndef RNAD_MLP_FWD_THREADS
#define RNAD_MLP_FWD_THREADS 256
#endif
constexpr int kFwdThreads = RNAD_MLP_FWD_THREADS;  // forward block size: waves of one block share one LDS weight image
constexpr int kTile = 32;  // samples per wave-tile and hidden units per MFMA tile
constexpr int kB1Pad = 12;  // floats reserved for the 1 + A output biases at the end of the packed image (multiple of 4)

template <typename T>
__device__ __forceinline__ float load_obs(const T *p);
template <>
__device__ __forceinline__ float load_obs<float>(const float *p) { return *p; }
template <>
__device__ __forceinline__ float load_obs<__half>(const __half *p) { return __half2float(*p); }

// LDS image (floats) = the packed weight image:  w0[(K + 2)][2W]  |  w1v[W]  |  w1p[A][W]  |  b1[1 + A] (padded to 12)
//
// HEADS: 1 = value only, 2 = policy only, 3 = both.  A wave keeps TWO hidden tiles in flight (two independent
// accumulator chains) and is software-pipelined by hand: the MFMA chains of the next tile pair are issued before the
// relu / second-layer VALU epilogue of the current pair.  (Measured on gfx950: fp32 MFMA and fp32 VALU work do NOT
// overlap -- kernel time is the sum of the two -- so what counts is the VALU instruction count of the epilogue: built
// with -mllvm -amdgpu-mfma-vgpr-form (no v_accvgpr_read) and -fno-honor-nans (no canonicalising v_max before relu).)
// z tile = b0 + W0 x.  The accumulator starts as the first-layer bias of this lane's 16 hidden rows -- four broadcast float4
// reads of the bias row (row K) of the LDS image, no VALU work -- then K / 2 MFMAs walk the input features: one MFMA fewer
// per tile than carrying the bias as an extra k-step.  `wt` points at column 0 of the hidden tile in w0.
template <int A>
__device__ __forceinline__ f32x16 mfma_chain(const float *__restrict__ wt, int W2, int col, int half, const float (&xk)[A * A]) {
    constexpr int K = 2 * A * A;
    const float *brow = wt + K * W2 + 4 * half;
    f32x16 c;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4 *>(brow + 8 * g);
        c[4 * g + 0] = b.x; c[4 * g + 1] = b.y; c[4 * g + 2] = b.z; c[4 * g + 3] = b.w;
    }
    const float *wa = wt + col;
#pragma unroll
    for (int ks = 0; ks < A * A; ++ks) c = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[(2 * ks + half) * W2], xk[ks], c, 0, 0, 0);
    return c;
}

// Second layer for one 32x32 hidden tile.  Four independent partial sums per output keep the fma chains short;
// explicit fmaf: the summation order here is this kernel's own (nothing in the reference fixes it).
__device__ __forceinline__ void epilogue_value(const f32x16 &c, const float *__restrict__ w1, float &acc) {
    float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 w = *reinterpret_cast<const float4 *>(w1 + 8 * g);
        p[0] = fmaf(w.x, fmaxf(c[4 * g + 0], 0.0f), p[0]);
        p[1] = fmaf(w.y, fmaxf(c[4 * g + 1], 0.0f), p[1]);
        p[2] = fmaf(w.z, fmaxf(c[4 * g + 2], 0.0f), p[2]);
        p[3] = fmaf(w.w, fmaxf(c[4 * g + 3], 0.0f), p[3]);
    }
    acc += (p[0] + p[1]) + (p[2] + p[3]);
}

template <int A>
__device__ __forceinline__ void epilogue_policy(const f32x16 &c, const float *__restrict__ w1, int W, float (&acc)[A]) {
    float p[A][4];
#pragma unroll
    for (int a = 0; a < A; ++a) p[a][0] = p[a][1] = p[a][2] = p[a][3] = 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float h0 = fmaxf(c[4 * g + 0], 0.0f), h1 = fmaxf(c[4 * g + 1], 0.0f);
        const float h2 = fmaxf(c[4 * g + 2], 0.0f), h3 = fmaxf(c[4 * g + 3], 0.0f);
#pragma unroll
        for (int a = 0; a < A; ++a) {
            const float4 w = *reinterpret_cast<const float4 *>(w1 + a * W + 8 * g);
            p[a][0] = fmaf(w.x, h0, p[a][0]);
            p[a][1] = fmaf(w.y, h1, p[a][1]);
            p[a][2] = fmaf(w.z, h2, p[a][2]);
            p[a][3] = fmaf(w.w, h3, p[a][3]);
        }
    }
#pragma unroll
    for (int a = 0; a < A; ++a) acc[a] += (p[a][0] + p[a][1]) + (p[a][2] + p[a][3]);
}

template <int A, typename ObsT, int HEADS>
__global__ __launch_bounds__(kFwdThreads) void k_mlp_forward(int64_t N, int W, const float *__restrict__ packed,
                                                          const ObsT *__restrict__ obs, float *__restrict__ logits,
                                                          float *__restrict__ value) {
    constexpr int K = 2 * A * A, KS = K / 2;  // MFMA k-steps per hidden tile
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int W2 = 2 * W;
    float *w0 = lds;                    // [(K + 2)][2W]
    float *w1v = lds + (K + 2) * W2;    // [W]
    float *w1p = w1v + W;               // [A][W]
    {   // weights: one coalesced 16-byte copy of the image rnad_mlp_pack laid out (w0 | w1v | w1p | b1)
        const int n4 = ((K + 2) * W2 + (1 + A) * W + kB1Pad) / 4;
        const float4 *src = reinterpret_cast<const float4 *>(packed);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < n4; i += kFwdThreads) dst[i] = src[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int T = W / kTile;  // hidden tiles per head
    const float *b1 = w1p + A * W;  // [1 + A]: value_fc1.bias, policy_fc1.bias
    const float bv = b1[0];
    float bp[A];
#pragma unroll
    for (int a = 0; a < A; ++a) bp[a] = b1[1 + a];
    // tile pair p of this launch: both heads -> (value tile p, policy tile p); one head -> its tiles (2p, 2p + 1)
    const int n_pairs = HEADS == 3 ? T : T / 2;  // single-head launches need an even tile count (the launcher sees to it)
    const int first = HEADS == 2 ? T : 0;
    const int stride0 = HEADS == 3 ? 1 : 2, off1 = HEADS == 3 ? T : 1;

    const int64_t n_tiles = (N + kTile - 1) / kTile;
    for (int64_t tile = (int64_t)blockIdx.x * (kFwdThreads / 64) + wave; tile < n_tiles; tile += (int64_t)gridDim.x * (kFwdThreads / 64)) {
        const int64_t sample = tile * kTile + col;
        const bool live = sample < N;
        float xk[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xk[ks] = live ? load_obs<ObsT>(obs + sample * K + 2 * ks + half) : 0.0f;

        float acc_v = 0.0f, acc_p[A];
#pragma unroll
        for (int a = 0; a < A; ++a) acc_p[a] = 0.0f;

        // epilogue of the tile pair whose first tile is `e0` (second: e0 + off1)
        auto finish = [&](const f32x16 &d0, const f32x16 &d1, int e0) {
            const int e1 = e0 + off1;
            if (HEADS == 1) {
                epilogue_value(d0, w1v + e0 * kTile + 4 * half, acc_v);
                epilogue_value(d1, w1v + e1 * kTile + 4 * half, acc_v);
            } else if (HEADS == 2) {
                epilogue_policy<A>(d0, w1p + (e0 - T) * kTile + 4 * half, W, acc_p);
                epilogue_policy<A>(d1, w1p + (e1 - T) * kTile + 4 * half, W, acc_p);
            } else {
                epilogue_value(d0, w1v + e0 * kTile + 4 * half, acc_v);
                epilogue_policy<A>(d1, w1p + (e1 - T) * kTile + 4 * half, W, acc_p);
            }
        };
        auto chain = [&](int t) { return mfma_chain<A>(w0 + t * kTile, W2, col, half, xk); };
        int t0 = first;
        for (int p = 0; p < n_pairs; ++p) {
            const f32x16 c0 = chain(t0), c1 = chain(t0 + off1);
            finish(c0, c1, t0);
            t0 += stride0;
        }
        // the two half-waves hold complementary hidden rows of the same 32 samples
        if (HEADS & 1) acc_v += __shfl_xor(acc_v, 32, 64);
        if (HEADS & 2) {
#pragma unroll
            for (int a = 0; a < A; ++a) acc_p[a] += __shfl_xor(acc_p[a], 32, 64);
        }
        if (live && half == 0) {
            if ((HEADS & 1) && value) value[sample] = acc_v + bv;
            if ((HEADS & 2) && logits) {
#pragma unroll
                for (int a = 0; a < A; ++a) logits[sample * A + a] = acc_p[a] + bp[a];
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------ backward
// Gradients of the 8 Linear tensors given dL/dlogits [N, A] and dL/dvalue [N] -- what autograd computes for
// nn/net.py:40-43 -- in ONE pass over the samples with the hidden layer recomputed on chip:
//     z = W0aug x (MFMA, as in the forward)          h = relu(z)
//     dW1[o, j] += dout[o] * h[j]                     (VALU, per-lane partial sums over this lane's samples)
//     dz[j] = (z[j] > 0) * sum_o W1[o, j] * dout[o]
//     dW0aug[j, k] += dz[j] * xaug[k]                 (MFMA with the 32 SAMPLES of the tile as the contraction dimension:
//                                                      A = dz^T via a per-wave 32x33 LDS transpose, B = x rows; the
//                                                      bias gradient is the k = K column because xaug[K] = 1)
// Block = W/32 waves; wave w owns hidden tile w of BOTH heads for every sample tile the block visits, so its two
// 32x32 dW0aug accumulators (32 registers) and its dW1 partials (16 + 16 A registers) stay resident for the whole
// launch.  Blocks write their partial gradients to `partial`; k_mlp_reduce sums them in a fixed order (deterministic).
template <int A, typename ObsT, int MAXT>
__global__ __launch_bounds__(MAXT) void k_mlp_backward(int64_t N, int W, const float *__restrict__ packed,
                                                       const ObsT *__restrict__ obs, const float *__restrict__ dlogit,
                                                       const float *__restrict__ dv, float *__restrict__ partial, int P) {
    constexpr int K = 2 * A * A, KS = K / 2;
    constexpr int FT = (K + 1 + kTile - 1) / kTile;  // 32-wide feature tiles of the augmented input (x | 1)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int W2 = 2 * W, nthreads = blockDim.x;
    float *w0 = lds;
    float *w1v = lds + (K + 2) * W2;
    float *w1p = w1v + W;
    float *scratch = w1p + A * W + kB1Pad;  // after b1: [waves][32][33]
    {
        const int n4 = ((K + 2) * W2 + (1 + A) * W + kB1Pad) / 4;
        const float4 *src = reinterpret_cast<const float4 *>(packed);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < n4; i += nthreads) dst[i] = src[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    float *tr = scratch + wave * (kTile * 33);
    // blockIdx.y selects a group of (blockDim.x / 64) hidden tiles; this wave owns one of them, in both heads
    const int own = blockIdx.y * (nthreads >> 6) + wave;
    const int tile_v = own, tile_p = W / kTile + own;

    f32x16 gW0v[FT], gW0p[FT];
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
        gW0v[ft] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        gW0p[ft] = gW0v[ft];
    }
    float gW1v[16], gW1p[A][16], gb1v = 0.0f, gb1p[A];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        gW1v[r] = 0.0f;
#pragma unroll
        for (int a = 0; a < A; ++a) gW1p[a][r] = 0.0f;
    }
#pragma unroll
    for (int a = 0; a < A; ++a) gb1p[a] = 0.0f;

    const int64_t n_tiles = (N + kTile - 1) / kTile;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t s0 = tile * kTile;
        const int64_t sample = s0 + col;
        const bool live = sample < N;
        float xk[KS];   // B operand of the forward product: x[sample = col][2 ks + half]
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xk[ks] = live ? load_obs<ObsT>(obs + sample * K + 2 * ks + half) : 0.0f;
        float xt[FT][16];   // B operand of the weight-gradient product: xaug[sample = 2 ks + half][feature = 32 ft + col]
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) {
            const int f = ft * kTile + col;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const int64_t sk = s0 + 2 * ks + half;
                float x = 0.0f;
                if (sk < N) x = f < K ? load_obs<ObsT>(obs + sk * K + f) : (f == K ? 1.0f : 0.0f);
                xt[ft][ks] = x;
            }
        }
        const float dvs = live ? dv[sample] : 0.0f;
        float dl[A];
#pragma unroll
        for (int a = 0; a < A; ++a) dl[a] = live ? dlogit[sample * A + a] : 0.0f;
        gb1v += dvs;
#pragma unroll
        for (int a = 0; a < A; ++a) gb1p[a] += dl[a];

        // ---------------- value head, hidden tile `tile_v`
        {
            const f32x16 c = mfma_chain<A>(w0 + tile_v * kTile, W2, col, half, xk);
            const float *w1 = w1v + tile_v * kTile + 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 w = *reinterpret_cast<const float4 *>(w1 + 8 * g);
                const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * g + j;
                    const float z = c[r];
                    gW1v[r] = fmaf(dvs, fmaxf(z, 0.0f), gW1v[r]);
                    tr[(j + 8 * g + 4 * half) * 33 + col] = z > 0.0f ? wv[j] * dvs : 0.0f;  // dz, stored [hidden][sample]
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const float at = tr[col * 33 + 2 * ks + half];
#pragma unroll
                for (int ft = 0; ft < FT; ++ft) gW0v[ft] = __builtin_amdgcn_mfma_f32_32x32x2f32(at, xt[ft][ks], gW0v[ft], 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
        // ---------------- policy head, hidden tile `tile_p`
        {
            const f32x16 c = mfma_chain<A>(w0 + tile_p * kTile, W2, col, half, xk);
            const float *w1 = w1p + own * kTile + 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float dh[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int a = 0; a < A; ++a) {
                    const float4 w = *reinterpret_cast<const float4 *>(w1 + a * W + 8 * g);
                    dh[0] = fmaf(w.x, dl[a], dh[0]); dh[1] = fmaf(w.y, dl[a], dh[1]);
                    dh[2] = fmaf(w.z, dl[a], dh[2]); dh[3] = fmaf(w.w, dl[a], dh[3]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * g + j;
                    const float z = c[r];
                    const float h = fmaxf(z, 0.0f);
#pragma unroll
                    for (int a = 0; a < A; ++a) gW1p[a][r] = fmaf(dl[a], h, gW1p[a][r]);
                    tr[(j + 8 * g + 4 * half) * 33 + col] = z > 0.0f ? dh[j] : 0.0f;
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const float at = tr[col * 33 + 2 * ks + half];
#pragma unroll
                for (int ft = 0; ft < FT; ++ft) gW0p[ft] = __builtin_amdgcn_mfma_f32_32x32x2f32(at, xt[ft][ks], gW0p[ft], 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }

    // ---------------- write this block's partial gradients
    // layout: dW0aug [2W][32 FT] | dW1v [W] | dW1p [A][W] | db1v | db1p [A]
    constexpr int FW = FT * kTile;
    float *out = partial + (int64_t)blockIdx.x * P;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) {
            out[(tile_v * kTile + row) * FW + ft * kTile + col] = gW0v[ft][r];
            out[(tile_p * kTile + row) * FW + ft * kTile + col] = gW0p[ft][r];
        }
    }
    float *o1 = out + W2 * FW;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = gW1v[r];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);  // over the 32 sample lanes of this half-wave
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        if (col == 0) o1[tile_v * kTile + row] = v;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            float p = gW1p[a][r];
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) p += __shfl_xor(p, off, 64);
            if (col == 0) o1[W + a * W + own * kTile + row] = p;
        }
    }
    if (own == 0) {  // every wave saw the same samples: one of them reports the output-bias gradients
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) gb1v += __shfl_xor(gb1v, off, 64);
        if (lane == 0) o1[W + A * W] = gb1v;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            float p = gb1p[a];
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) p += __shfl_xor(p, off, 64);
            if (lane == 0) o1[W + A * W + 1 + a] = p;
        }
    }
}

// Sum the per-block partials (fixed order, fp64 accumulate) into the eight gradient tensors (torch Linear layouts).
template <int A>
__global__ __launch_bounds__(kThreads) void k_mlp_reduce(int nblocks, int W, int P, const float *__restrict__ partial,
                                                         float *__restrict__ g_vw0, float *__restrict__ g_vb0, float *__restrict__ g_vw1,
                                                         float *__restrict__ g_vb1, float *__restrict__ g_pw0, float *__restrict__ g_pb0,
                                                         float *__restrict__ g_pw1, float *__restrict__ g_pb1) {
    constexpr int K = 2 * A * A, FW = ((K + 1 + kTile - 1) / kTile) * kTile;
    const int e = blockIdx.x * kThreads + threadIdx.x;
    const int total = 2 * W * FW + W + A * W + 1 + A;
    if (e >= total) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += (double)partial[(int64_t)b * P + e];
    const float v = (float)s;
    const int n0 = 2 * W * FW;
    if (e < n0) {
        const int h = e / FW, k = e % FW;
        float *gw = h < W ? g_vw0 : g_pw0, *gb = h < W ? g_vb0 : g_pb0;
        const int hh = h < W ? h : h - W;
        if (k < K) gw[hh * K + k] = v;
        else if (k == K) gb[hh] = v;
    } else if (e < n0 + W) {
        g_vw1[e - n0] = v;
    } else if (e < n0 + W + A * W) {
        g_pw1[e - n0 - W] = v;
    } else if (e == n0 + W + A * W) {
        g_vb1[0] = v;
    } else {
        g_pb1[e - n0 - W - A * W - 1] = v;
    }
}

// Lay the eight torch Linear tensors out as the LDS image the kernels copy in one go:
//   w0[(K + 2)][2W] (k-major; row K = first-layer biases, row K + 1 = 0)  |  w1v[W]  |  w1p[A][W]  |  b1[1 + A], padded to 12
__global__ __launch_bounds__(kThreads) void k_mlp_pack(int A, int W, const float *__restrict__ vw0, const float *__restrict__ vb0,
                                                       const float *__restrict__ vw1, const float *__restrict__ vb1,
                                                       const float *__restrict__ pw0, const float *__restrict__ pb0,
                                                       const float *__restrict__ pw1, const float *__restrict__ pb1,
                                                       float *__restrict__ packed, int total) {
    const int K = 2 * A * A, W2 = 2 * W;
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= total) return;
    const int n0 = (K + 2) * W2;
    float x = 0.0f;
    if (i < n0) {
        const int k = i / W2, h = i % W2;
        if (k < K) x = h < W ? vw0[h * K + k] : pw0[(h - W) * K + k];
        else if (k == K) x = h < W ? vb0[h] : pb0[h - W];
    } else if (i < n0 + W) {
        x = vw1[i - n0];
    } else if (i < n0 + W + A * W) {
        x = pw1[i - n0 - W];
    } else if (i == n0 + W + A * W) {
        x = vb1[0];
    } else if (i < n0 + W + A * W + 1 + A) {
        x = pb1[i - n0 - W - A * W - 1];
    }
    packed[i] = x;
}

}  // namespace

static inline int mlp_packed_floats(int A, int W) { return (2 * A * A + 2) * 2 * W + (1 + A) * W + kB1Pad; }

extern "C" int64_t rnad_mlp_packed_size(int A, int W) { return mlp_packed_floats(A, W); }

extern "C" int rnad_mlp_pack(int A, int W, const float *vw0, const float *vb0, const float *vw1, const float *vb1, const float *pw0,
                             const float *pb0, const float *pw1, const float *pb1, float *packed, void *stream) {
    RNAD_REQUIRE(vw0 && vb0 && vw1 && vb1 && pw0 && pb0 && pw1 && pb1 && packed, "rnad_mlp_pack: null argument");
    RNAD_REQUIRE(A >= 1 && A <= RNAD_MAX_ACTIONS && W >= kTile && W % kTile == 0, "rnad_mlp_pack: bad shape (A=%d, width=%d)", A, W);
    const int total = mlp_packed_floats(A, W);
    hipLaunchKernelGGL(k_mlp_pack, dim3((total + kThreads - 1) / kThreads), dim3(kThreads), 0, (hipStream_t)stream, A, W, vw0, vb0, vw1,
                       vb1, pw0, pb0, pw1, pb1, packed, total);
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int rnad_mlp_forward(int64_t N, int A, int W, const float *packed, const void *obs, int obs_half, float *logits, float *value,
                                void *stream_) {
    RNAD_REQUIRE(packed && obs && (logits || value), "rnad_mlp_forward: null argument");
    RNAD_REQUIRE(W >= kTile && W % kTile == 0, "rnad_mlp_forward: width %d must be a positive multiple of %d", W, kTile);
    RNAD_REQUIRE(N >= 0, "rnad_mlp_forward: negative batch");
    if (N == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    const int K = 2 * A * A;
    (void)K;
    const size_t lds_bytes = (size_t)mlp_packed_floats(A, W) * sizeof(float);
    RNAD_REQUIRE(lds_bytes <= 160 * 1024, "rnad_mlp_forward: weights (%zu B) do not fit the 160 KiB LDS (A=%d, width=%d)", lds_bytes, A, W);
    int dev = 0, cus = 256;
    RNAD_HIP_OK(hipGetDevice(&dev));
    RNAD_HIP_OK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    constexpr int kWaves = kFwdThreads / 64;
    const int blocks_per_cu = std::max(1, std::min(12 / kWaves, (int)(160 * 1024 / lds_bytes)));
    const int64_t n_tiles = (N + kTile - 1) / kTile;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_tiles + kWaves - 1) / kWaves, (int64_t)cus * blocks_per_cu));
    int heads = (value ? 1 : 0) | (logits ? 2 : 0);
    if ((W / kTile) % 2) heads = 3;  // odd tile count: the paired single-head kernels do not apply; compute both, store the wanted one
    ProfScope prof(PROF_MLP, stream);
#define RNAD_MLP_LAUNCH2(T_, H_)                                                                                                  \
    do {                                                                                                                           \
        auto kern = k_mlp_forward<kA, T_, H_>;                                                                                     \
        if (lds_bytes > 64 * 1024)                                                                                                 \
            RNAD_HIP_OK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));      \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kFwdThreads), lds_bytes, stream, N, W, packed, (const T_ *)obs, logits, value);            \
                                                                        \
    } while (0)
#define RNAD_MLP_LAUNCH(T_)                                   \
    do {                                                      \
        if (heads == 1) RNAD_MLP_LAUNCH2(T_, 1);              \
        else if (heads == 2) RNAD_MLP_LAUNCH2(T_, 2);         \
        else RNAD_MLP_LAUNCH2(T_, 3);                         \
    } while (0)
    RNAD_DISPATCH_A(A, {
        if (obs_half)
            RNAD_MLP_LAUNCH(__half);
        else
            RNAD_MLP_LAUNCH(float);
    });
#undef RNAD_MLP_LAUNCH2
#undef RNAD_MLP_LAUNCH
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

struct BwdPlan {
    int waves, groups, grid_x, P, total;
    size_t lds_bytes;
};

// One wave per hidden tile (of both heads).  With one feature tile (A <= 3) a wave needs ~230 VGPRs: 8 waves per block, two
// per SIMD.  With more feature tiles it needs up to ~400: 4 waves per block, one per SIMD, and blockIdx.y walks the tile groups.
static bool mlp_backward_plan(int64_t N, int W, int A, BwdPlan *p) {
    const int K = 2 * A * A, T = W / kTile, FT = (K + 1 + kTile - 1) / kTile;
    int waves = FT == 1 ? 8 : 4;
    while (waves > 1 && T % waves) waves >>= 1;
    p->waves = waves;
    p->groups = T / waves;
    p->lds_bytes = ((size_t)mlp_packed_floats(A, W) + (size_t)waves * kTile * 33) * sizeof(float);
    p->total = 2 * W * FT * kTile + W + A * W + 1 + A;
    p->P = (p->total + 3) & ~3;
    if (p->lds_bytes > 160 * 1024) return false;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int64_t n_tiles = (N + kTile - 1) / kTile;
    p->grid_x = (int)std::max<int64_t>(1, std::min<int64_t>(n_tiles, std::max(1, cus / p->groups)));
    return true;
}

extern "C" int64_t rnad_mlp_backward_workspace(int64_t N, int A, int W) {
    BwdPlan p;
    if (A < 1 || A > RNAD_MAX_ACTIONS || W < kTile || W % kTile || !mlp_backward_plan(N, W, A, &p)) return -1;
    return (int64_t)p.grid_x * p.P * (int64_t)sizeof(float);
}

extern "C" int rnad_mlp_backward(int64_t N, int A, int W, const float *packed, const void *obs, int obs_half, const float *dlogits,
                                 const float *dvalue, float *g_vw0, float *g_vb0, float *g_vw1, float *g_vb1, float *g_pw0,
                                 float *g_pb0, float *g_pw1, float *g_pb1, float *workspace, void *stream_) {
    RNAD_REQUIRE(packed && obs && dlogits && dvalue && g_vw0 && g_vb0 && g_vw1 && g_vb1 && g_pw0 && g_pb0 && g_pw1 && g_pb1 && workspace,
                 "rnad_mlp_backward: null argument");
    RNAD_REQUIRE(W >= kTile && W % kTile == 0, "rnad_mlp_backward: width %d must be a positive multiple of %d", W, kTile);
    RNAD_REQUIRE(N >= 1, "rnad_mlp_backward: empty batch");
    hipStream_t stream = (hipStream_t)stream_;
    BwdPlan plan;
    RNAD_REQUIRE(A >= 1 && A <= RNAD_MAX_ACTIONS && mlp_backward_plan(N, W, A, &plan),
                 "rnad_mlp_backward: weights do not fit the LDS (A=%d, width=%d)", A, W);
    const int grid = plan.grid_x, P = plan.P;
    const size_t lds_bytes = plan.lds_bytes;
    const int threads = 64 * plan.waves;
    {
        ProfScope prof(PROF_MLP_BWD, stream);
#define RNAD_MLPB_LAUNCH(T_, MAXT_)                                                                                                \
    do {                                                                                                                           \
        auto kern = k_mlp_backward<kA, T_, MAXT_>;                                                                                 \
        if (lds_bytes > 64 * 1024)                                                                                                 \
            RNAD_HIP_OK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));      \
        hipLaunchKernelGGL(kern, dim3(grid, plan.groups), dim3(threads), lds_bytes, stream, N, W, packed, (const T_ *)obs, dlogits, \
                           dvalue, workspace, P);                                                                                  \
    } while (0)
        RNAD_DISPATCH_A(A, {
            constexpr int kMaxT = (2 * kA * kA + 1 <= kTile) ? 512 : 256;
            if (obs_half) RNAD_MLPB_LAUNCH(__half, kMaxT);
            else RNAD_MLPB_LAUNCH(float, kMaxT);
        });
#undef RNAD_MLPB_LAUNCH
        RNAD_HIP_OK(hipGetLastError());
    }
    const int total = plan.total;
    const unsigned rgrid = (unsigned)((total + kThreads - 1) / kThreads);
    RNAD_DISPATCH_A(A, hipLaunchKernelGGL((k_mlp_reduce<kA>), dim3(rgrid), dim3(kThreads), 0, stream, grid, W, P, workspace, g_vw0, g_vb0,
                                          g_vw1, g_vb1, g_pw0, g_pb0, g_pw1, g_pb1));
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}
