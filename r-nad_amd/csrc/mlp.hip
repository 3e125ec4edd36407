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
#include <cstdlib>

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

// Packed weight image (floats), copied verbatim into LDS by every block:
//   w0t [2T][K/2][2][32]   first-layer weights, hidden tile major: element ((tile * K/2 + ks) * 2 + half) * 32 + col is
//                          W0[hidden = 32 tile + col][k = 2 ks + half] -- exactly the A operand of MFMA k-step ks for lane
//                          (col, half), so one base address + immediate offsets ks * 256 B serve a whole chain
//   b0  [2W]               first-layer biases (value head | policy head)
//   w1v [W], w1p [A][W]    second-layer weights
//   b1  [1 + A] (pad 12)   second-layer biases
// T = W / 32 hidden tiles per head; tiles 0..T-1 = value head, T..2T-1 = policy head.
__host__ __device__ constexpr int img_b0(int K, int W) { return 2 * W * K; }
__host__ __device__ constexpr int img_w1v(int K, int W) { return img_b0(K, W) + 2 * W; }
__host__ __device__ constexpr int img_w1p(int K, int W) { return img_w1v(K, W) + W; }
__host__ __device__ constexpr int img_b1(int K, int W, int A) { return img_w1p(K, W) + A * W; }
__host__ __device__ constexpr int img_floats(int K, int W, int A) { return img_b1(K, W, A) + kB1Pad; }

typedef float f32x2 __attribute__((ext_vector_type(2)));

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
// z tile = b0 + W0 x.  The accumulator starts as the first-layer bias of this lane's 16 hidden rows (four broadcast float4
// reads, no VALU work), then K / 2 MFMAs walk the input features; their A operands are loaded up front from one base
// address with immediate offsets.
template <int A>
__device__ __forceinline__ f32x16 mfma_chain(const float *__restrict__ lds, int W, int tile, int col, int half, const float (&xk)[A * A]) {
    constexpr int K = 2 * A * A, KS = A * A;
    const float *wa = lds + tile * (KS * 64) + half * 32 + col;
    const float *brow = lds + img_b0(K, W) + tile * kTile + 4 * half;
    float a[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[ks] = wa[ks * 64];
    f32x16 c;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4 *>(brow + 8 * g);
        c[4 * g + 0] = b.x; c[4 * g + 1] = b.y; c[4 * g + 2] = b.z; c[4 * g + 3] = b.w;
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], xk[ks], c, 0, 0, 0);
    return c;
}

__device__ __forceinline__ f32x2 relu2(float a, float b) { return f32x2{fmaxf(a, 0.0f), fmaxf(b, 0.0f)}; }

// Second layer for one 32x32 hidden tile, written on float pairs so that it compiles to v_pk_fma_f32 without register
// shuffles (fp32 MFMA and VALU work do not overlap on gfx950: every VALU instruction here is kernel time).  The summation
// order is this kernel's own; nothing in the reference fixes it.
__device__ __forceinline__ void epilogue_value(const f32x16 &c, const float *__restrict__ w1, f32x2 (&acc)[2]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 w = *reinterpret_cast<const float4 *>(w1 + 8 * g);
        acc[0] = __builtin_elementwise_fma(f32x2{w.x, w.y}, relu2(c[4 * g + 0], c[4 * g + 1]), acc[0]);
        acc[1] = __builtin_elementwise_fma(f32x2{w.z, w.w}, relu2(c[4 * g + 2], c[4 * g + 3]), acc[1]);
    }
}

template <int A>
__device__ __forceinline__ void epilogue_policy(const f32x16 &c, const float *__restrict__ w1, int W, f32x2 (&acc)[A][2]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x2 h01 = relu2(c[4 * g + 0], c[4 * g + 1]), h23 = relu2(c[4 * g + 2], c[4 * g + 3]);
#pragma unroll
        for (int a = 0; a < A; ++a) {
            const float4 w = *reinterpret_cast<const float4 *>(w1 + a * W + 8 * g);
            acc[a][0] = __builtin_elementwise_fma(f32x2{w.x, w.y}, h01, acc[a][0]);
            acc[a][1] = __builtin_elementwise_fma(f32x2{w.z, w.w}, h23, acc[a][1]);
        }
    }
}

template <int A, typename ObsT, int HEADS>
__global__ __launch_bounds__(kFwdThreads) void k_mlp_forward(int64_t N, int W, const float *__restrict__ packed,
                                                             const ObsT *__restrict__ obs, float *__restrict__ logits,
                                                             float *__restrict__ value) {
    constexpr int K = 2 * A * A, KS = K / 2;  // MFMA k-steps per hidden tile
    extern __shared__ __attribute__((aligned(16))) float lds[];
    {   // weights: one coalesced 16-byte copy of the image rnad_mlp_pack laid out
        const int n4 = img_floats(K, W, A) / 4;
        const float4 *src = reinterpret_cast<const float4 *>(packed);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < n4; i += kFwdThreads) dst[i] = src[i];
    }
    __syncthreads();
    const float *w1v = lds + img_w1v(K, W);
    const float *w1p = lds + img_w1p(K, W);
    const float *b1 = lds + img_b1(K, W, A);  // [1 + A]: value_fc1.bias, policy_fc1.bias

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int T = W / kTile;  // hidden tiles per head
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

        f32x2 acc_v[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}}, acc_p[A][2];
#pragma unroll
        for (int a = 0; a < A; ++a) acc_p[a][0] = acc_p[a][1] = f32x2{0.f, 0.f};

        int t0 = first;
        for (int p = 0; p < n_pairs; ++p) {
            const int t1 = t0 + off1;
            const f32x16 c0 = mfma_chain<A>(lds, W, t0, col, half, xk);
            const f32x16 c1 = mfma_chain<A>(lds, W, t1, col, half, xk);
            if (HEADS == 1) {
                epilogue_value(c0, w1v + t0 * kTile + 4 * half, acc_v);
                epilogue_value(c1, w1v + t1 * kTile + 4 * half, acc_v);
            } else if (HEADS == 2) {
                epilogue_policy<A>(c0, w1p + (t0 - T) * kTile + 4 * half, W, acc_p);
                epilogue_policy<A>(c1, w1p + (t1 - T) * kTile + 4 * half, W, acc_p);
            } else {
                epilogue_value(c0, w1v + t0 * kTile + 4 * half, acc_v);
                epilogue_policy<A>(c1, w1p + (t1 - T) * kTile + 4 * half, W, acc_p);
            }
            t0 += stride0;
        }
        // lane-local sums, then the two half-waves (complementary hidden rows of the same 32 samples)
        float out_v = (acc_v[0].x + acc_v[0].y) + (acc_v[1].x + acc_v[1].y), out_p[A];
#pragma unroll
        for (int a = 0; a < A; ++a) out_p[a] = (acc_p[a][0].x + acc_p[a][0].y) + (acc_p[a][1].x + acc_p[a][1].y);
        if (HEADS & 1) out_v += __shfl_xor(out_v, 32, 64);
        if (HEADS & 2) {
#pragma unroll
            for (int a = 0; a < A; ++a) out_p[a] += __shfl_xor(out_p[a], 32, 64);
        }
        if (live && half == 0) {
            if ((HEADS & 1) && value) value[sample] = out_v + bv;
            if ((HEADS & 2) && logits) {
#pragma unroll
                for (int a = 0; a < A; ++a) logits[sample * A + a] = out_p[a] + bp[a];
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
    const float *w1v = lds + img_w1v(K, W);
    const float *w1p = lds + img_w1p(K, W);
    float *scratch = lds + img_floats(K, W, A);  // after the image: [waves][32][33]
    {
        const int n4 = img_floats(K, W, A) / 4;
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
    f32x2 gW1v[8], gW1p[A][8];  // second-layer weight-gradient partials of this lane's 16 hidden rows, as register pairs
    float gb1v = 0.0f, gb1p[A];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        gW1v[r] = f32x2{0.f, 0.f};
#pragma unroll
        for (int a = 0; a < A; ++a) gW1p[a][r] = f32x2{0.f, 0.f};
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
            const f32x16 c = mfma_chain<A>(lds, W, tile_v, col, half, xk);
            const float *w1 = w1v + tile_v * kTile + 4 * half;
            const f32x2 dv2 = {dvs, dvs};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 w = *reinterpret_cast<const float4 *>(w1 + 8 * g);
                const f32x2 wq[2] = {f32x2{w.x, w.y}, f32x2{w.z, w.w}};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float z0 = c[4 * g + 2 * j], z1 = c[4 * g + 2 * j + 1];
                    gW1v[2 * g + j] = __builtin_elementwise_fma(dv2, relu2(z0, z1), gW1v[2 * g + j]);
                    const f32x2 dz = wq[j] * dv2;  // dL/dz where the unit is active
                    float *t = tr + (2 * j + 8 * g + 4 * half) * 33 + col;  // stored [hidden][sample]
                    t[0] = z0 > 0.0f ? dz.x : 0.0f;
                    t[33] = z1 > 0.0f ? dz.y : 0.0f;
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
            const f32x16 c = mfma_chain<A>(lds, W, tile_p, col, half, xk);
            const float *w1 = w1p + own * kTile + 4 * half;
            f32x2 dl2[A];
#pragma unroll
            for (int a = 0; a < A; ++a) dl2[a] = f32x2{dl[a], dl[a]};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x2 dh[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
                for (int a = 0; a < A; ++a) {
                    const float4 w = *reinterpret_cast<const float4 *>(w1 + a * W + 8 * g);
                    dh[0] = __builtin_elementwise_fma(f32x2{w.x, w.y}, dl2[a], dh[0]);
                    dh[1] = __builtin_elementwise_fma(f32x2{w.z, w.w}, dl2[a], dh[1]);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float z0 = c[4 * g + 2 * j], z1 = c[4 * g + 2 * j + 1];
                    const f32x2 h = relu2(z0, z1);
#pragma unroll
                    for (int a = 0; a < A; ++a) gW1p[a][2 * g + j] = __builtin_elementwise_fma(dl2[a], h, gW1p[a][2 * g + j]);
                    float *t = tr + (2 * j + 8 * g + 4 * half) * 33 + col;
                    t[0] = z0 > 0.0f ? dh[j].x : 0.0f;
                    t[33] = z1 > 0.0f ? dh[j].y : 0.0f;
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
        float v = gW1v[r >> 1][r & 1];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);  // over the 32 sample lanes of this half-wave
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        if (col == 0) o1[tile_v * kTile + row] = v;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            float p = gW1p[a][r >> 1][r & 1];
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

// Lay the eight torch Linear tensors out as the LDS image described at the top of this file.
__global__ __launch_bounds__(kThreads) void k_mlp_pack(int A, int W, const float *__restrict__ vw0, const float *__restrict__ vb0,
                                                       const float *__restrict__ vw1, const float *__restrict__ vb1,
                                                       const float *__restrict__ pw0, const float *__restrict__ pb0,
                                                       const float *__restrict__ pw1, const float *__restrict__ pb1,
                                                       float *__restrict__ packed, int total) {
    const int K = 2 * A * A, KS = A * A;
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= total) return;
    float x = 0.0f;
    if (i < img_b0(K, W)) {
        const int tile = i / (KS * 64), rem = i % (KS * 64);
        const int ks = rem / 64, half = (rem % 64) / 32, col = rem % 32;
        const int h = tile * kTile + col, k = 2 * ks + half;
        x = h < W ? vw0[h * K + k] : pw0[(h - W) * K + k];
    } else if (i < img_w1v(K, W)) {
        const int h = i - img_b0(K, W);
        x = h < W ? vb0[h] : pb0[h - W];
    } else if (i < img_w1p(K, W)) {
        x = vw1[i - img_w1v(K, W)];
    } else if (i < img_b1(K, W, A)) {
        x = pw1[i - img_w1p(K, W)];
    } else if (i == img_b1(K, W, A)) {
        x = vb1[0];
    } else if (i < img_b1(K, W, A) + 1 + A) {
        x = pb1[i - img_b1(K, W, A) - 1];
    }
    packed[i] = x;
}

}  // namespace

static inline int mlp_packed_floats(int A, int W) { return img_floats(2 * A * A, W, A); }

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
    size_t lds_pad = 0;
    if (const char *e = getenv("RNAD_MLP_LDS_PAD")) lds_pad = (size_t)atoi(e);  // experiment knob: fewer blocks per CU
    constexpr int kWaves = kFwdThreads / 64;
    const int blocks_per_cu = std::max(1, std::min(12 / kWaves, (int)(160 * 1024 / (lds_bytes + lds_pad))));
    const int64_t n_tiles = (N + kTile - 1) / kTile;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_tiles + kWaves - 1) / kWaves, (int64_t)cus * blocks_per_cu));
    int heads = (value ? 1 : 0) | (logits ? 2 : 0);
    if ((W / kTile) % 2) heads = 3;  // odd tile count: the paired single-head kernels do not apply; compute both, store the wanted one
    ProfScope prof(PROF_MLP, stream);
#define RNAD_MLP_LAUNCH2(T_, H_)                                                                                                  \
    do {                                                                                                                           \
        auto kern = k_mlp_forward<kA, T_, H_>;                                                                                     \
        if (lds_bytes + lds_pad > 64 * 1024)                                                                                       \
            RNAD_HIP_OK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_bytes + lds_pad))); \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kFwdThreads), lds_bytes + lds_pad, stream, N, W, packed, (const T_ *)obs, logits, value);            \
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
