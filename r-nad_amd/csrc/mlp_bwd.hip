// mlp_bwd.hip -- backward of the fused policy/value MLP (see mlp_fwd.hip for the forward and the mapping).
// Built WITHOUT -mllvm -amdgpu-mfma-vgpr-form: that pass crashes hipcc (ROCm 7.2) on this kernel; the backward gained < 3 % from it.
#include "mlp_common.hpp"

using namespace rnad;
using namespace rnad_mlp;

namespace rnad_mlp {
size_t mlp_backward_t_lds(int A, bool fold);
int mlp_backward_t_blocks_per_cu(int A, int waves, bool fold);
int mlp_backward_t_launch(int A, int waves, dim3 grid, hipStream_t stream, int64_t N, int W, const float *packed, const void *obs,
                           int obs_half, const float *dlogits, const float *dvalue, float *workspace, int P, const int32_t *rows,
                           const int64_t *n_rows, bool fold);
}  // namespace rnad_mlp

namespace {

constexpr int kTS = 36;  // row stride (floats) of the per-wave dz tiles

// ------------------------------------------------------------------------------------------------ backward
// Gradients of the 8 Linear tensors given dL/dlogits [N, A] and dL/dvalue [N] -- what autograd computes for
// nn/net.py:40-43 -- in ONE pass over the samples with the hidden layer recomputed on chip:
//     z = W0aug x (MFMA, as in the forward)          h = relu(z)
//     dW1[o, j] += dout[o] * h[j]                     (VALU, per-lane partial sums over this lane's samples)
//     dz[j] = (z[j] > 0) * sum_o W1[o, j] * dout[o]
//     dW0aug[j, k] += dz[j] * xaug[k]                 (MFMA with the 32 SAMPLES of the tile as the contraction dimension:
//                                                      A = dz^T via per-wave 32x36 LDS transposes, B = xaug rows from the
//                                                      block's LDS stage; 16-wide feature tiles on 16x16x4, <= 4 left-over
//                                                      features on 4x4x1_16b; the bias gradient is the k = K column
//                                                      because xaug[K] = 1)
// Block = up to 8 waves; wave w owns hidden tile w of BOTH heads for every sample tile the block visits, so its dW0aug
// accumulators (2 heads x 2 row tiles x N16 feature tiles x 4 registers, + 16 for the left-overs) and its dW1 partials
// (16 + 16 A registers) stay resident for the whole launch.  Blocks write their partial gradients to `partial`; k_mlp_reduce
// sums them in a fixed order (deterministic).
template <int A, typename ObsT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_mlp_backward(int64_t N, int W, const float *__restrict__ packed,
                                                       const ObsT *__restrict__ obs, const float *__restrict__ dlogit,
                                                       const float *__restrict__ dv, float *__restrict__ partial, int P,
                                                       const int32_t *__restrict__ rows, const int64_t *__restrict__ n_rows) {
    if (n_rows) N = *n_rows;  // row-list launch (see k_mlp_forward): sample s is row rows[s]; the count lives in device memory
    constexpr int K = 2 * A * A, KS = K / 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int nthreads = 64 * WAVES;
    const int W2 = 2 * W;
    const float *w1v = lds + img_w1v(K, W);
    const float *w1p = lds + img_w1p(K, W);
    float *scratch = lds + img_floats(K, W, A);  // after the image: [waves][2][32][kTS]
    load_image<nthreads>(packed, lds, img_floats(K, W, A) / 4);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    // dz of the two heads, [hidden][sample]; row stride 36 floats: rows are 16-byte aligned (the left-over pass reads them as
    // float4) and the strided accesses of the 16-wide pass stay at the 2-way minimum of 64 lanes over 32 banks
    float *tr_v = scratch + wave * (2 * kTile * kTS), *tr_p = tr_v + kTile * kTS;
    // blockIdx.y selects a group of (blockDim.x / 64) hidden tiles; this wave owns one of them, in both heads
    const int own = blockIdx.y * WAVES + wave;
    const int tile_v = own, tile_p = W / kTile + own;

    // dW0aug[hidden 32][K + 1 features] of this wave's tile, per head: 16-wide feature tiles on v_mfma_f32_16x16x4_f32 (M = 16
    // hidden rows, N = 16 features, K = 4 samples per instruction) and, when at most 4 features are left over, those on
    // v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4 hidden rows x 4 features, one sample per instruction, both heads at once).  A
    // 32-wide tile (v_mfma_f32_32x32x2_f32) would spend 32/19 of the matrix time on the 19 features of A = 3.
    constexpr int REM = (K + 1) % 16, N16 = (K + 1) / 16 + (REM > 4 ? 1 : 0), LO = REM > 4 ? 0 : REM;
    constexpr int N16R = N16 > 0 ? N16 : 1;  // array extent
    f32x4 gW0v[2][N16R], gW0p[2][N16R], gW0lo[4];  // four left-over accumulators, one per sample % 4
#pragma unroll
    for (int c = 0; c < 4; ++c) gW0lo[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < N16R; ++nt) gW0v[mt][nt] = gW0p[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int m16 = lane & 15, q4 = lane >> 4;
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

    // The 32 samples of a tile (x, dL/dvalue, dL/dlogits: 32 (K + 1 + A) floats) are shared by every wave of the block: they
    // are fetched once with coalesced loads, one tile ahead, and handed over through a double-buffered LDS stage (one barrier
    // per tile).  Per-wave global loads of the same tile cost 13 % of the kernel (measured).
    // stage row = xaug padded to the MFMA feature tiles: x[0..K) | 1 | 0 ..., so the dW0 operands are plain loads; the stride
    // is made odd (no LDS bank conflicts between the 32 rows)
    constexpr int XS = bwd_stage_stride(K);
    // ... and the <= 4 left-over feature columns once more, transposed ([4][32]: a lane of the left-over pass reads one column)
    constexpr int XN = kTile * XS, XLO = XN + kTile + kTile * A, STG = XLO + 4 * kTile;  // floats per stage: xaug | dv | dlogits | xlo
    float *stage = scratch + WAVES * (2 * kTile * kTS);          // [2][STG]
    const int64_t n_tiles = (N + kTile - 1) / kTile;
    // TPS threads share one sample of the tile: thread (smp, part) loads elements part, part + TPS, ... of that sample's row.
    // One row id per thread and tile, itself prefetched a further tile ahead (row-list launches: a lookup that the x loads
    // would otherwise have to wait for, exposed, at the top of every tile).
    constexpr int TPS = nthreads / kTile, XU = (K + TPS - 1) / TPS, DU = (A + TPS - 1) / TPS;
    const int smp = threadIdx.x / TPS, part = threadIdx.x % TPS;
    float pre_x[XU], pre_dv = 0.0f, pre_dl[DU];
    auto row_of = [&](int64_t tile) -> int64_t {  // -1: past the end
        const int64_t sample = tile * kTile + smp;
        if (tile >= n_tiles || sample >= N) return -1;
        return rows ? (int64_t)rows[sample] : sample;
    };
    auto fetch = [&](int64_t row) {  // global -> registers
        const bool in = row >= 0;
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int k = part + u * TPS;
            pre_x[u] = (in && k < K) ? load_obs<ObsT>(obs + row * K + k) : 0.0f;
        }
        pre_dv = (in && part == 0) ? dv[row] : 0.0f;
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            const int a = part + u * TPS;
            pre_dl[u] = (in && a < A) ? dlogit[row * A + a] : 0.0f;
        }
    };
    auto park = [&](float *dst) {  // registers -> LDS stage
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int k = part + u * TPS;
            if (k < K) {
                dst[smp * XS + k] = pre_x[u];
                if (LO > 0 && k >= N16 * 16) dst[XLO + (k - N16 * 16) * kTile + smp] = pre_x[u];
            }
        }
        if (part == 0) dst[XN + smp] = pre_dv;
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            const int a = part + u * TPS;
            if (a < A) dst[XN + kTile + smp * A + a] = pre_dl[u];
        }
    };
    for (int i = threadIdx.x; i < 2 * kTile * (XS - K); i += nthreads) {  // the constant columns of both stages, written once
        const int buf = i / (kTile * (XS - K)), r = i % (kTile * (XS - K));
        const int smp_ = r / (XS - K), f = K + r % (XS - K);
        stage[buf * STG + smp_ * XS + f] = f == K ? 1.0f : 0.0f;
    }
    if (LO > 0)
        for (int i = threadIdx.x; i < 2 * 4 * kTile; i += nthreads) {
            const int buf = i / (4 * kTile), f = N16 * 16 + (i % (4 * kTile)) / kTile;
            if (f >= K) stage[buf * STG + XLO + i % (4 * kTile)] = f == K ? 1.0f : 0.0f;
        }
    int cur = 0;
    if ((int64_t)blockIdx.x < n_tiles) {
        fetch(row_of(blockIdx.x));
        park(stage);
    }
    int64_t row_next = row_of((int64_t)blockIdx.x + gridDim.x);
    __syncthreads();
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const bool more = tile + gridDim.x < n_tiles;
        if (more) fetch(row_next);  // next tile's loads are in flight during this tile's matrix work
        row_next = row_of(tile + 2 * (int64_t)gridDim.x);
        const float *xs = stage + cur * STG;
        float xk[KS];   // B operand of the forward product: x[sample = col][2 ks + half]
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xk[ks] = xs[col * XS + 2 * ks + half];
        const float dvs = xs[XN + col];  // zero for samples past N, so padded lanes contribute nothing
        float dl[A];
#pragma unroll
        for (int a = 0; a < A; ++a) dl[a] = xs[XN + kTile + col * A + a];
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
                    float *t = tr_v + (2 * j + 8 * g + 4 * half) * kTS + col;  // stored [hidden][sample]
                    t[0] = z0 > 0.0f ? dz.x : 0.0f;
                    t[kTS] = z1 > 0.0f ? dz.y : 0.0f;
                }
            }
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
                    float *t = tr_p + (2 * j + 8 * g + 4 * half) * kTS + col;
                    t[0] = z0 > 0.0f ? dh[j].x : 0.0f;
                    t[kTS] = z1 > 0.0f ? dh[j].y : 0.0f;
                }
            }
        }
        // ---------------- dW0aug += dz^T xaug for both heads (samples are the contraction dimension)
        __builtin_amdgcn_wave_barrier();
        if (N16 > 0) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {  // samples 4 ks .. 4 ks + 3; this lane supplies sample 4 ks + q4
                float bx[N16R];
#pragma unroll
                for (int nt = 0; nt < N16; ++nt) {
                    bx[nt] = xs[(4 * ks + q4) * XS + nt * 16 + m16];
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const float av = tr_v[(mt * 16 + m16) * kTS + 4 * ks + q4], ap = tr_p[(mt * 16 + m16) * kTS + 4 * ks + q4];
#pragma unroll
                    for (int nt = 0; nt < N16; ++nt) {
                        gW0v[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bx[nt], gW0v[mt][nt], 0, 0, 0);
                        gW0p[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap, bx[nt], gW0p[mt][nt], 0, 0, 0);
                    }
                }
            }
        }
        if (LO > 0) {
            // block b = lane / 4 holds hidden rows 4 b .. 4 b + 3 of the value tile (b < 8) or the policy tile; column lane % 4 is
            // feature 16 N16 + lane % 4
            const float4 *a_src = reinterpret_cast<const float4 *>(lane < 32 ? tr_v + lane * kTS : tr_p + (lane - 32) * kTS);
            const float4 *b_src = reinterpret_cast<const float4 *>(xs + XLO + (lane & 3) * kTile);
#pragma unroll
            for (int s0 = 0; s0 < kTile / 4; s0 += 2) {  // eight samples per round: two 16-byte reads per operand
                const float4 a0 = a_src[s0], a1 = a_src[s0 + 1], b0 = b_src[s0], b1 = b_src[s0 + 1];
                gW0lo[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.x, b0.x, gW0lo[0], 0, 0, 0);
                gW0lo[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.y, b0.y, gW0lo[1], 0, 0, 0);
                gW0lo[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.z, b0.z, gW0lo[2], 0, 0, 0);
                gW0lo[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.w, b0.w, gW0lo[3], 0, 0, 0);
                gW0lo[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.x, b1.x, gW0lo[0], 0, 0, 0);
                gW0lo[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.y, b1.y, gW0lo[1], 0, 0, 0);
                gW0lo[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.z, b1.z, gW0lo[2], 0, 0, 0);
                gW0lo[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.w, b1.w, gW0lo[3], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (more) park(stage + (cur ^ 1) * STG);  // nobody reads that stage any more: every wave passed the last barrier
        __syncthreads();
        cur ^= 1;
    }

    // ---------------- write this block's partial gradients
    // layout: dW0aug [2W][FW] | dW1v [W] | dW1p [A][W] | db1v | db1p [A]
    constexpr int FW = bwd_feature_stride(K);
    float *out = partial + (int64_t)blockIdx.x * P;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < N16; ++nt) {
            const int f = nt * 16 + m16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = mt * 16 + 4 * q4 + i;
                if (f <= K) {
                    out[(tile_v * kTile + row) * FW + f] = gW0v[mt][nt][i];
                    out[(tile_p * kTile + row) * FW + f] = gW0p[mt][nt][i];
                }
            }
        }
    if (LO > 0 && (lane & 3) < LO) {
        const int b = lane >> 2, f = N16 * 16 + (lane & 3);
        const int tile = b < 8 ? tile_v : tile_p;
#pragma unroll
        for (int i = 0; i < 4; ++i) out[(tile * kTile + 4 * (b & 7) + i) * FW + f] = (gW0lo[0][i] + gW0lo[1][i]) + (gW0lo[2][i] + gW0lo[3][i]);
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
// blockDim = (64 outputs, kReduceSlices): slice s adds blocks s, s + kReduceSlices, ... of its output, then the slices are
// added in order -- the same grouping on every run, 1 / kReduceSlices of the dependent-load chain of one thread per output.
// FOLD (mlp_common.hpp "the legal fold"): the partials are those of the folded first layer -- columns ev [A^2] | indicator | (padding) |
// bias -- and the gradients of the eight ORIGINAL tensors follow from them exactly:
//   dW0[h][k < A^2] = column k;   db0[h] = bias column;   with d_abs = indicator column (the rows of the absorbing state):
//   dW0[h][A^2 + j] = sum_rows dz legal[row][j] = (db0 - d_abs) + d_abs e0[j]  ->  db0 for j = 0, db0 - d_abs for j >= 1.
constexpr int kReduceSlices = 16;
template <int A, bool FOLD>
__global__ __launch_bounds__(64 * kReduceSlices) void k_mlp_reduce(int nblocks, int W, int P, const float *__restrict__ partial,
                                                                   float *__restrict__ g_vw0, float *__restrict__ g_vb0,
                                                                   float *__restrict__ g_vw1, float *__restrict__ g_vb1,
                                                                   float *__restrict__ g_pw0, float *__restrict__ g_pb0,
                                                                   float *__restrict__ g_pw1, float *__restrict__ g_pb1) {
    constexpr int K = MlpShape<A, FOLD>::K, OBS = MlpShape<A, FOLD>::OBS, FW = bwd_feature_stride(K);
    __shared__ double part[kReduceSlices][64], part_ind[kReduceSlices][64];
    const int e = blockIdx.x * 64 + threadIdx.x;
    const int total = 2 * W * FW + W + A * W + 1 + A;
    double s = 0.0, s_ind = 0.0;
    const bool padding = e < 2 * W * FW && e % FW > K;  // columns of the dW0aug tiles beyond the bias column: never stored
    const bool bias_col = FOLD && e < 2 * W * FW && e % FW == K;  // (it also needs its row's indicator column)
    if (e < total && !padding)
        for (int b = threadIdx.y; b < nblocks; b += kReduceSlices) {
            s += (double)partial[(int64_t)b * P + e];
            if (bias_col) s_ind += (double)partial[(int64_t)b * P + e - (K - A * A)];
        }
    part[threadIdx.y][threadIdx.x] = s;
    if (FOLD) part_ind[threadIdx.y][threadIdx.x] = s_ind;
    __syncthreads();
    if (threadIdx.y != 0 || e >= total) return;
#pragma unroll
    for (int i = 1; i < kReduceSlices; ++i) {
        s += part[i][threadIdx.x];
        if (FOLD) s_ind += part_ind[i][threadIdx.x];
    }
    const float v = (float)s;
    const int n0 = 2 * W * FW;
    if (e < n0) {
        const int h = e / FW, k = e % FW;
        float *gw = h < W ? g_vw0 : g_pw0, *gb = h < W ? g_vb0 : g_pb0;
        const int hh = h < W ? h : h - W;
        if constexpr (FOLD) {
            if (k < A * A) {
                gw[hh * OBS + k] = v;
            } else if (k == K) {
                gb[hh] = v;
                gw[hh * OBS + A * A] = v;
                const float rest = (float)(s - s_ind);
#pragma unroll
                for (int j = 1; j < A * A; ++j) gw[hh * OBS + A * A + j] = rest;
            }
        } else {
            if (k < K) gw[hh * K + k] = v;
            else if (k == K) gb[hh] = v;
        }
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

}  // namespace

struct BwdPlan {
    int waves, groups, grid_x, P, total;
    size_t lds_bytes;
    bool resident;  // mlp_bwd_t.hip (dz and the wave's weights in registers) instead of the LDS-transpose kernel of this file
};

static bool use_resident_backward() {
    static const int v = [] {
        const char *e = getenv("RNAD_MLP_BWD");  // "lds": the kernel of this file; anything else / unset: the register-resident one
        return (e && e[0] == 'l') ? 0 : 1;
    }();
    return v != 0;
}

// One wave per hidden tile (of both heads).  With one feature tile (A <= 3) a wave needs ~230 VGPRs: 8 waves per block, two
// per SIMD.  With more feature tiles it needs up to ~400: 4 waves per block, one per SIMD, and blockIdx.y walks the tile groups.
static bool mlp_backward_plan(int64_t N, int W, int A, BwdPlan *p, bool fold = false) {
    const int K = fold ? mlp_fold_k(A) : 2 * A * A, T = W / kTile, FT = (K + 1 + kTile - 1) / kTile;
    p->total = 2 * W * bwd_feature_stride(K) + W + A * W + 1 + A;
    p->P = (p->total + 3) & ~3;
    int dev0 = 0, cus0 = 256;
    if (hipGetDevice(&dev0) == hipSuccess) (void)hipDeviceGetAttribute(&cus0, hipDeviceAttributeMultiprocessorCount, dev0);
    const int64_t n_tiles0 = (N + kTile - 1) / kTile;
    p->resident = fold || use_resident_backward();  // (the fold exists in the register-resident kernel only)
    if (p->resident) {
        // 4 waves per block (one per SIMD), as many blocks per CU as the registers allow; the LDS holds only the sample stage
        int wv = 4;  // measured: 4 waves per block 5.31 ms, 8: 5.41, 2: 5.72 (configs[1], 12.6 M samples)
        while (wv > 1 && T % wv) wv >>= 1;
        p->waves = wv;
        p->groups = T / wv;
        p->lds_bytes = mlp_backward_t_lds(A, fold);
        p->grid_x = (int)std::max<int64_t>(1, std::min<int64_t>(n_tiles0, std::max(1, cus0 * mlp_backward_t_blocks_per_cu(A, wv, fold) / p->groups)));
        return true;
    }
    int waves = FT == 1 ? 8 : 4;
    auto lds_for = [&](int wv) {
        return ((size_t)mlp_packed_floats(A, W) + (size_t)wv * 2 * kTile * kTS + 2 * (size_t)kTile * (bwd_stage_stride(K) + 1 + A + 4)) * sizeof(float);
    };
    while (waves > 1 && (T % waves || lds_for(waves) > 160 * 1024)) waves >>= 1;  // fewer waves per block: smaller dz scratch
    p->waves = waves;
    p->groups = T / waves;
    p->lds_bytes = lds_for(waves);
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
    int64_t bytes = (int64_t)p.grid_x * p.P * (int64_t)sizeof(float);
    BwdPlan f;  // (rnad_mlp_backward_fold shares the workspace: its plan may run more, smaller blocks)
    if (A >= 2 && mlp_backward_plan(N, W, A, &f, true)) bytes = std::max(bytes, (int64_t)f.grid_x * f.P * (int64_t)sizeof(float));
    return bytes;
}

static int mlp_backward_launch(int64_t N, const int32_t *rows, const int64_t *n_rows, int A, int W, const float *packed, const void *obs,
                               int obs_half, const float *dlogits, const float *dvalue, float *g_vw0, float *g_vb0, float *g_vw1,
                               float *g_vb1, float *g_pw0, float *g_pb0, float *g_pw1, float *g_pb1, float *workspace, void *stream_,
                               bool fold = false) {
    RNAD_REQUIRE(packed && obs && dlogits && dvalue && g_vw0 && g_vb0 && g_vw1 && g_vb1 && g_pw0 && g_pb0 && g_pw1 && g_pb1 && workspace,
                 "rnad_mlp_backward: null argument");
    RNAD_REQUIRE(W >= kTile && W % kTile == 0, "rnad_mlp_backward: width %d must be a positive multiple of %d", W, kTile);
    RNAD_REQUIRE(N >= 1, "rnad_mlp_backward: empty batch");
    hipStream_t stream = (hipStream_t)stream_;
    BwdPlan plan;
    RNAD_REQUIRE(!fold || A >= 2, "rnad_mlp_backward_fold: the legal fold needs at least two actions");
    RNAD_REQUIRE(A >= 1 && A <= RNAD_MAX_ACTIONS && mlp_backward_plan(N, W, A, &plan, fold),
                 "rnad_mlp_backward: weights do not fit the LDS (A=%d, width=%d)", A, W);
    const int grid = plan.grid_x, P = plan.P;
    const size_t lds_bytes = plan.lds_bytes;
    const int threads = 64 * plan.waves;
    if (plan.resident) {
        ProfScope prof(PROF_MLP_BWD, stream);
        if (int rc = mlp_backward_t_launch(A, plan.waves, dim3(grid, plan.groups), stream, N, W, packed, obs, obs_half, dlogits, dvalue,
                                           workspace, P, rows, n_rows, fold))
            return rc;
        RNAD_HIP_OK(hipGetLastError());
    } else {
        ProfScope prof(PROF_MLP_BWD, stream);
#define RNAD_MLPB_LAUNCH(T_, WV_)                                                                                                 \
    do {                                                                                                                           \
        auto kern = k_mlp_backward<kA, T_, WV_>;                                                                                   \
        if (lds_bytes > 64 * 1024)                                                                                                 \
            RNAD_HIP_OK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));      \
        hipLaunchKernelGGL(kern, dim3(grid, plan.groups), dim3(threads), lds_bytes, stream, N, W, packed, (const T_ *)obs, dlogits, \
                           dvalue, workspace, P, rows, n_rows);                                                                                  \
    } while (0)
#define RNAD_MLPB_WAVES(T_)                                                      \
    do {                                                                         \
        if (plan.waves == 8) { if constexpr (kOneFt) RNAD_MLPB_LAUNCH(T_, 8); }  \
        else if (plan.waves == 4) RNAD_MLPB_LAUNCH(T_, 4);                       \
        else if (plan.waves == 2) RNAD_MLPB_LAUNCH(T_, 2);                       \
        else RNAD_MLPB_LAUNCH(T_, 1);                                            \
    } while (0)
        RNAD_DISPATCH_A(A, {
            constexpr bool kOneFt = 2 * kA * kA + 1 <= kTile;
            if (obs_half) RNAD_MLPB_WAVES(__half);
            else RNAD_MLPB_WAVES(float);
        });
#undef RNAD_MLPB_WAVES
#undef RNAD_MLPB_LAUNCH
        RNAD_HIP_OK(hipGetLastError());
    }
    const int total = plan.total;
    const unsigned rgrid = (unsigned)((total + 63) / 64);
    if (fold) {
        RNAD_DISPATCH_A(A, hipLaunchKernelGGL((k_mlp_reduce<kA, true>), dim3(rgrid), dim3(64, kReduceSlices), 0, stream, grid, W, P, workspace,
                                              g_vw0, g_vb0, g_vw1, g_vb1, g_pw0, g_pb0, g_pw1, g_pb1));
    } else {
        RNAD_DISPATCH_A(A, hipLaunchKernelGGL((k_mlp_reduce<kA, false>), dim3(rgrid), dim3(64, kReduceSlices), 0, stream, grid, W, P, workspace,
                                              g_vw0, g_vb0, g_vw1, g_vb1, g_pw0, g_pb0, g_pw1, g_pb1));
    }
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int rnad_mlp_backward(int64_t N, int A, int W, const float *packed, const void *obs, int obs_half, const float *dlogits,
                                 const float *dvalue, float *g_vw0, float *g_vb0, float *g_vw1, float *g_vb1, float *g_pw0,
                                 float *g_pb0, float *g_pw1, float *g_pb1, float *workspace, void *stream) {
    return mlp_backward_launch(N, nullptr, nullptr, A, W, packed, obs, obs_half, dlogits, dvalue, g_vw0, g_vb0, g_vw1, g_vb1, g_pw0, g_pb0,
                               g_pw1, g_pb1, workspace, stream);
}

extern "C" int rnad_mlp_backward_rows(int64_t max_rows, const int32_t *rows, const int64_t *n_rows, int A, int W, const float *packed,
                                      const void *obs, int obs_half, const float *dlogits, const float *dvalue, float *g_vw0,
                                      float *g_vb0, float *g_vw1, float *g_vb1, float *g_pw0, float *g_pb0, float *g_pw1, float *g_pb1,
                                      float *workspace, void *stream) {
    RNAD_REQUIRE(rows && n_rows, "rnad_mlp_backward_rows: null row list");
    return mlp_backward_launch(max_rows, rows, n_rows, A, W, packed, obs, obs_half, dlogits, dvalue, g_vw0, g_vb0, g_vw1, g_vb1, g_pw0,
                               g_pb0, g_pw1, g_pb1, workspace, stream);
}

// The FOLD instantiation (mlp_common.hpp "the legal fold"; packed: rnad_mlp_pack_fold_multi; obs rows as in rnad_mlp_forward_fold): the
// gradients of the eight ORIGINAL Linear tensors.  rows / n_rows: NULL or a row list (N = its capacity).  workspace:
// rnad_mlp_backward_workspace(N, A, W) bytes are enough.
extern "C" int rnad_mlp_backward_fold(int64_t N, const int32_t *rows, const int64_t *n_rows, int A, int W, const float *packed, const void *obs,
                                      int obs_half, const float *dlogits, const float *dvalue, float *g_vw0, float *g_vb0, float *g_vw1,
                                      float *g_vb1, float *g_pw0, float *g_pb0, float *g_pw1, float *g_pb1, float *workspace, void *stream) {
    RNAD_REQUIRE(!rows == !n_rows, "rnad_mlp_backward_fold: rows and n_rows go together");
    return mlp_backward_launch(N, rows, n_rows, A, W, packed, obs, obs_half, dlogits, dvalue, g_vw0, g_vb0, g_vw1, g_vb1, g_pw0, g_pb0, g_pw1,
                               g_pb1, workspace, stream, true);
}
