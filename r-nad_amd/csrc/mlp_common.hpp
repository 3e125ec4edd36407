// mlp_common.hpp -- shared pieces of the fused MLP kernels (mlp_fwd.hip, mlp_bwd.hip): weight-image layout, observation
// loads, the first-layer MFMA chain.
#pragma once

#include "common.hpp"

#include <algorithm>
#include <cstdlib>

namespace rnad_mlp {


typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
#ifndef RNAD_MLP_FWD_THREADS
#define RNAD_MLP_FWD_THREADS 256
#endif
constexpr int kFwdThreads = RNAD_MLP_FWD_THREADS;  // forward block size: waves of one block share one LDS weight image
constexpr int kTile = 32;  // samples per wave-tile and hidden units per MFMA tile
// Floats per hidden unit of dW0aug in a backward partial: the K inputs and the bias column, rounded up to a 16-byte multiple (the
// partial rows are what k_mlp_reduce streams: a 32-float stride left 2/3 of every cache line of a K = 10 row unused).
// (Wider inputs keep whole 32-float tiles: hipcc 7.2 crashes in its AGPR-copy rewrite on the A = 8 instantiation otherwise.)
constexpr int bwd_feature_stride(int K) { return K + 1 <= kTile ? ((K + 1 + 3) & ~3) : ((K + 1 + kTile - 1) / kTile) * kTile; }
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
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------- the legal fold
// An observation is [expected value A x A | legal mask A x A] from the mover's point of view (environment/episode.py:62-68).  On a tree
// whose states all have the full A x A action set (every configuration of BASELINE.json) the legal plane is the SAME in every row --
// all ones -- except in the two rows of the absorbing state, where it is e0 = [1, 0, ..., 0] (tree.py:133 with one action each).  The
// first layer then is   z = W_ev ev + (b0 + W_legal 1) + [row is absorbing] (W_legal e0 - W_legal 1):
// A^2 + 1 input features (the expected values and the indicator 1 - legal[0][1]) instead of 2 A^2, i.e. 5 MFMA k-steps instead of 9 at
// A = 3 -- the same function of the same weights, summed in another order.  The FOLD instantiations of the MLP kernels take the
// observations in the usual layout (row stride 2 A^2), read only what they need, and fold the legal columns of the weight image into
// the bias and the indicator column when they load it (the image keeps the raw weights: the optimiser writes it element by element).
//   MlpShape<A, FOLD>::K   input features of the first layer as the kernels see them (even: an MFMA k-step takes two)
template <int A, bool FOLD>
struct MlpShape {
    static constexpr int OBS = 2 * A * A;                             // floats of an observation row
    static constexpr int K = FOLD ? ((A * A + 2) & ~1) : 2 * A * A;  // FOLD: ev [A^2] | indicator | zero padding to an even count
    static constexpr int KS = K / 2;
};

// Weight image -> LDS, 16 bytes per lane and request.  Eight requests are in flight per lane before the first LDS write: a
// plain copy loop waits for every global load in turn (an L2 round trip per 4 KiB), which is most of a rollout-step launch
// at 2^17 samples.
template <int NT>
__device__ __forceinline__ void load_image(const float *__restrict__ packed, float *__restrict__ lds, int n4) {
    const float4 *src = reinterpret_cast<const float4 *>(packed);
    float4 *dst = reinterpret_cast<float4 *>(lds);
    constexpr int U = 8;
    int i = threadIdx.x;
    for (; i + (U - 1) * NT < n4; i += U * NT) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[i + u * NT];
#pragma unroll
        for (int u = 0; u < U; ++u) dst[i + u * NT] = v[u];
    }
    for (; i < n4; i += NT) dst[i] = src[i];
}

// raw legal columns of a FOLD image, after the output biases: [A^2][2W] (column major: the fold reads them coalesced over hidden units)
__host__ __device__ constexpr int img_legal(int K, int W, int A) { return img_b1(K, W, A) + kB1Pad; }
__host__ __device__ constexpr int img_floats_fold(int K, int W, int A) { return img_legal(K, W, A) + A * A * 2 * W; }

template <typename T>
__device__ __forceinline__ float load_obs(const T *p);
template <>
__device__ __forceinline__ float load_obs<float>(const float *p) { return *p; }
template <>
__device__ __forceinline__ float load_obs<__half>(const __half *p) { return __half2float(*p); }

// Input feature k of an observation row as the first layer sees it.
template <int A, bool FOLD, typename T>
__device__ __forceinline__ float obs_feature(const T *row, int k) {
    if constexpr (!FOLD) {
        return load_obs<T>(row + k);
    } else {
        if (k < A * A) return load_obs<T>(row + k);
        if (k == A * A) return 1.0f - load_obs<T>(row + A * A + (A > 1 ? 1 : 0));  // 1 in the rows of the absorbing state, else 0
        return 0.0f;
    }
}

// FOLD: what a kernel does to the first layer of a freshly loaded weight image (`img`: LDS or registers behind a lambda):
//   s = sum_k W_legal[h][k] (k ascending), b0'[h] = b0[h] + s, w_ind[h] = W_legal[h][0] - s.
// The same order everywhere, so the recomputed hidden layer of the backward is the forward's.
template <int A, typename Load>
__device__ __forceinline__ void fold_hidden_unit(Load legal_col, float b0, float &b0_folded, float &w_ind) {
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < A * A; ++k) s += legal_col(k);
    b0_folded = b0 + s;
    w_ind = legal_col(0) - s;
}

// One hidden tile of the first layer for one 32-sample tile (the backward's recompute; the forward has its own
// two-sample-tile variant in mlp_fwd.hip).  z tile = b0 + W0 x.  The accumulator starts as the first-layer bias of this lane's
// 16 hidden rows (four broadcast float4 reads, no VALU work), then K / 2 MFMAs walk the input features; their A operands are
// loaded up front from one base address with immediate offsets.
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

// floats per sample row of the backward's LDS stage: the augmented input (x | 1) padded with zeros to whole MFMA feature tiles
// (16-wide tiles, plus one 4-wide tile when at most 4 features are left over), made odd
__host__ __device__ constexpr int bwd_stage_stride(int K) {
    const int rem = (K + 1) % 16, n16 = (K + 1) / 16 + (rem > 4 ? 1 : 0), lo = rem > 4 ? 0 : rem;
    return (n16 * 16 + (lo > 0 ? 4 : 0)) | 1;
}

static inline int mlp_packed_floats(int A, int W) { return img_floats(2 * A * A, W, A); }
static inline int mlp_fold_k(int A) { return (A * A + 2) & ~1; }
static inline int mlp_packed_floats_fold(int A, int W) { return img_floats_fold(mlp_fold_k(A), W, A); }

}  // namespace rnad_mlp
