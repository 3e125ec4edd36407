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

template <typename T>
__device__ __forceinline__ float load_obs(const T *p);
template <>
__device__ __forceinline__ float load_obs<float>(const float *p) { return *p; }
template <>
__device__ __forceinline__ float load_obs<__half>(const __half *p) { return __half2float(*p); }

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

// floats per sample row of the backward's LDS stage: the augmented input (x | 1) padded with zeros to whole MFMA feature tiles
// (16-wide tiles, plus one 4-wide tile when at most 4 features are left over), made odd
__host__ __device__ constexpr int bwd_stage_stride(int K) {
    const int rem = (K + 1) % 16, n16 = (K + 1) / 16 + (rem > 4 ? 1 : 0), lo = rem > 4 ? 0 : rem;
    return (n16 * 16 + (lo > 0 ? 4 : 0)) | 1;
}

static inline int mlp_packed_floats(int A, int W) { return img_floats(2 * A * A, W, A); }

}  // namespace rnad_mlp
