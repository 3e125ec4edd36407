// mlp_bwd_t.hip -- backward of the fused policy/value MLP, register-resident variant.
//
// Same mathematics and the same per-block partial layout as mlp_bwd.hip (k_mlp_reduce sums both), different mapping.  The
// hidden layer is recomputed TRANSPOSED, z^T[sample, hidden] = xaug[sample, :] W0aug^T, on v_mfma_f32_16x16x4_f32 tiles
// (M = 16 samples, N = 16 hidden units, K = 4 input features): the accumulator of such a tile holds, in lane (hidden m, q),
// the four samples 4 q .. 4 q + 3 of hidden unit m -- which is exactly the A-operand layout (rows = hidden, K = four sample
// slots) of the weight-gradient product dW0aug[hidden, feature] += dz[hidden, sample] xaug[sample, feature] on the same
// instruction.  So dz never leaves the register file: no LDS transposes (mlp_bwd.hip: 64 LDS instructions per tile and wave),
// and because a wave owns one hidden tile for the whole launch its first-layer weights (B operands of the recompute) and
// second-layer weights live in registers too: the kernel keeps NO weight image in LDS, only the block's sample stage.
//   recompute   K / 4 k-steps per 16x16 tile; the first-layer bias of the lane's hidden unit, replicated in a resident
//               register quad, is the C operand of the first one (destination != source: no per-tile initialisation)
//   dW1         one scalar accumulator per lane and hidden unit (its lane IS the hidden unit), summed over q at the end
//   dW0aug      16-wide feature tiles on 16x16x4; <= 4 left-over features on v_mfma_f32_4x4x1_16b_f32, whose block
//               (q, m / 4) pairs hidden units 4 (m / 4) .. + 3 with the left-over features for ITS sample; the four q blocks
//               are summed at the end
#include "mlp_common.hpp"

using namespace rnad;
using namespace rnad_mlp;

namespace rnad_mlp {

template <int A, typename ObsT, int WAVES, bool FOLD>
__global__ __launch_bounds__(64 * WAVES) void k_mlp_backward_t(int64_t N, int W, const float *__restrict__ packed,
                                                              const ObsT *__restrict__ obs, const float *__restrict__ dlogit,
                                                              const float *__restrict__ dv, float *__restrict__ partial, int P,
                                                              const int32_t *__restrict__ rows, const int64_t *__restrict__ n_rows) {
    if (n_rows) N = *n_rows;  // row-list launch: sample s is row rows[s]; the count lives in device memory
    // FOLD (mlp_common.hpp "the legal fold"): K = A^2 + 1 (+ padding) input features -- the expected values and the absorbing-state
    // indicator --, the legal columns of the raw weight image folded into this wave's bias and indicator weight below
    constexpr int K = MlpShape<A, FOLD>::K, KS = K / 2, OBS = MlpShape<A, FOLD>::OBS;
    constexpr int FW = bwd_feature_stride(K);  // row stride of dW0aug in the partial buffer
    constexpr int REM = (K + 1) % 16, N16 = (K + 1) / 16 + (REM > 4 ? 1 : 0), LO = REM > 4 ? 0 : REM;
    constexpr int N16R = N16 > 0 ? N16 : 1;
    constexpr int KQ = (K + 3) / 4;  // k-steps of the recompute (four input features each; the stage rows are zero-padded)
    constexpr int XS = bwd_stage_stride(K);
    static_assert(XS >= 4 * KQ, "stage rows must cover the padded augmented input");
    constexpr int nthreads = 64 * WAVES;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // stage (floats): xaug [32][XS] | dv [32] | dlogits [A][32] | left-over feature columns [4][32]
    constexpr int XN = kTile * XS, XDL = XN + kTile, XLO = XDL + kTile * A, STG = XLO + 4 * kTile;
    float *stage = lds;  // [2][STG]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m16 = lane & 15, q = lane >> 4;
    const int own = blockIdx.y * WAVES + wave;  // this wave's hidden tile, in both heads
    const int tile_v = own, tile_p = W / kTile + own;

    // ---------------- resident operands
    float wB[2][2][KQ];  // [head][16-row half of the tile][k-step]: W0[hidden][4 kk + q], zero beyond K
    f32x4 bias4[2][2];   // b0[hidden] in all four components: the accumulator of a z tile starts from it
#pragma unroll
    for (int hd = 0; hd < 2; ++hd)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int h = (hd ? tile_p : tile_v) * kTile + 16 * mt + m16;  // row of the stacked [2W] first layer
            float b = packed[img_b0(K, W) + h], w_ind = 0.0f;
            if constexpr (FOLD) {
                const float *lc = packed + img_legal(K, W, A) + h;
                fold_hidden_unit<A>([&](int k) { return lc[k * 2 * W]; }, b, b, w_ind);
            }
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk) {
                const int k = 4 * kk + q;
                float w = 0.0f;
                if (k < K) w = packed[(h / kTile) * (KS * 64) + (k / 2) * 64 + (k % 2) * 32 + (h % kTile)];
                if (FOLD && k == A * A) w = w_ind;
                wB[hd][mt][kk] = w;
            }
            bias4[hd][mt] = f32x4{b, b, b, b};
        }
    float w1v_[2], w1p_[A][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        w1v_[mt] = packed[img_w1v(K, W) + tile_v * kTile + 16 * mt + m16];
#pragma unroll
        for (int a = 0; a < A; ++a) w1p_[a][mt] = packed[img_w1p(K, W) + a * W + own * kTile + 16 * mt + m16];
    }

    f32x4 gW0[2][2][N16R], gLo[2][2];
#pragma unroll
    for (int hd = 0; hd < 2; ++hd)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            gLo[hd][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < N16R; ++nt) gW0[hd][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    float gW1v[2] = {0.f, 0.f}, gW1p[A][2], gb1v = 0.0f, gb1p[A];
#pragma unroll
    for (int a = 0; a < A; ++a) gW1p[a][0] = gW1p[a][1] = gb1p[a] = 0.0f;

    // ---------------- the block's sample stage (as in mlp_bwd.hip: fetched once per block, one tile ahead, double-buffered)
    const int64_t n_tiles = (N + kTile - 1) / kTile;
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
            pre_x[u] = (in && k < K) ? obs_feature<A, FOLD, ObsT>(obs + row * OBS, k) : 0.0f;
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
            if (a < A) dst[XDL + a * kTile + smp] = pre_dl[u];
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

        // operands shared by both heads
        float xa[2][KQ];  // recompute A operand: xaug[sample 16 st + m16][4 kk + q]
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk) xa[st][kk] = xs[(16 * st + m16) * XS + 4 * kk + q];
        float bf[2][4][N16R];  // dW0 B operand: xaug[sample 16 st + 4 q + i][16 nt + m16]
        f32x4 blo[2];          // left-over B operand: xaug[sample 16 st + 4 q + i][16 N16 + m16 % 4], i = component
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int nt = 0; nt < N16; ++nt) bf[st][i][nt] = xs[(16 * st + 4 * q + i) * XS + 16 * nt + m16];
            if (LO > 0) {
                const float4 t = *reinterpret_cast<const float4 *>(xs + XLO + (m16 & 3) * kTile + 16 * st + 4 * q);
                blo[st] = f32x4{t.x, t.y, t.z, t.w};
            }
        }
        f32x4 dvq[2], dlq[A][2];  // dL/dvalue, dL/dlogits of this lane's samples 16 st + 4 q + i
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const float4 t = *reinterpret_cast<const float4 *>(xs + XN + 16 * st + 4 * q);
            dvq[st] = f32x4{t.x, t.y, t.z, t.w};
            gb1v += (t.x + t.y) + (t.z + t.w);
#pragma unroll
            for (int a = 0; a < A; ++a) {
                const float4 u = *reinterpret_cast<const float4 *>(xs + XDL + a * kTile + 16 * st + 4 * q);
                dlq[a][st] = f32x4{u.x, u.y, u.z, u.w};
                gb1p[a] += (u.x + u.y) + (u.z + u.w);
            }
        }

#pragma unroll
        for (int hd = 0; hd < 2; ++hd) {
            f32x4 dz[2][2];  // [st][mt]: dL/dz of hidden 16 mt + m16 at samples 16 st + 4 q + i
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    f32x4 z = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[st][0], wB[hd][mt][0], bias4[hd][mt], 0, 0, 0);
#pragma unroll
                    for (int kk = 1; kk < KQ; ++kk) z = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[st][kk], wB[hd][mt][kk], z, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float zz = z[i], h = fmaxf(zz, 0.0f);
                        float up;  // dL/dh = sum_o W1[o, hidden] dL/dout[o]
                        if (hd == 0) {
                            up = w1v_[mt] * dvq[st][i];
                            gW1v[mt] = __builtin_fmaf(dvq[st][i], h, gW1v[mt]);
                        } else {
                            up = 0.0f;
#pragma unroll
                            for (int a = 0; a < A; ++a) {
                                up = __builtin_fmaf(w1p_[a][mt], dlq[a][st][i], up);
                                gW1p[a][mt] = __builtin_fmaf(dlq[a][st][i], h, gW1p[a][mt]);
                            }
                        }
                        dz[st][mt][i] = zz > 0.0f ? up : 0.0f;
                    }
                }
            // dW0aug += dz xaug: the four sample slots of a lane group are the contraction dimension
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                        for (int nt = 0; nt < N16; ++nt)
                            gW0[hd][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dz[st][mt][i], bf[st][i][nt], gW0[hd][mt][nt], 0, 0, 0);
                        if (LO > 0) gLo[hd][mt] = __builtin_amdgcn_mfma_f32_4x4x1f32(dz[st][mt][i], blo[st][i], gLo[hd][mt], 0, 0, 0);
                    }
        }
        if (more) park(stage + (cur ^ 1) * STG);  // nobody reads that stage any more: every wave passed the last barrier
        __syncthreads();
        cur ^= 1;
    }

    // ---------------- write this block's partial gradients
    // layout: dW0aug [2W][FW] | dW1v [W] | dW1p [A][W] | db1v | db1p [A]
    float *out = partial + (int64_t)blockIdx.x * P;
#pragma unroll
    for (int hd = 0; hd < 2; ++hd) {
        const int tile = hd ? tile_p : tile_v;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int nt = 0; nt < N16; ++nt) {
                const int f = nt * 16 + m16;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (f <= K) out[(tile * kTile + 16 * mt + 4 * q + i) * FW + f] = gW0[hd][mt][nt][i];
            }
            if (LO > 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = gLo[hd][mt][i];  // block (q, m16 / 4): hidden 16 mt + 4 (m16 / 4) + i, feature 16 N16 + m16 % 4, its q's samples
                    v += __shfl_xor(v, 16, 64);
                    v += __shfl_xor(v, 32, 64);
                    if (q == 0 && (m16 & 3) < LO) out[(tile * kTile + 16 * mt + 4 * (m16 >> 2) + i) * FW + N16 * 16 + (m16 & 3)] = v;
                }
            }
        }
    }
    float *o1 = out + 2 * W * FW;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        float v = gW1v[mt];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (q == 0) o1[tile_v * kTile + 16 * mt + m16] = v;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            float p = gW1p[a][mt];
            p += __shfl_xor(p, 16, 64);
            p += __shfl_xor(p, 32, 64);
            if (q == 0) o1[W + a * W + own * kTile + 16 * mt + m16] = p;
        }
    }
    if (own == 0) {  // every wave saw the same samples: one of them reports the output-bias gradients
        gb1v += __shfl_xor(gb1v, 16, 64);
        gb1v += __shfl_xor(gb1v, 32, 64);
        if (lane == 0) o1[W + A * W] = gb1v;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            float p = gb1p[a];
            p += __shfl_xor(p, 16, 64);
            p += __shfl_xor(p, 32, 64);
            if (lane == 0) o1[W + A * W + 1 + a] = p;
        }
    }
}

// Launch: WAVES = 4 waves per block (a quarter of the hidden tiles of width 128 per head); the block's LDS is only the stage.
template <int A, typename ObsT, bool FOLD>
static void launch_t(int waves, dim3 grid, size_t lds_bytes, hipStream_t stream, int64_t N, int W, const float *packed, const void *obs,
                     const float *dlogits, const float *dvalue, float *workspace, int P, const int32_t *rows, const int64_t *n_rows) {
    switch (waves) {
        case 8: hipLaunchKernelGGL((k_mlp_backward_t<A, ObsT, 8, FOLD>), grid, dim3(512), lds_bytes, stream, N, W, packed, (const ObsT *)obs, dlogits, dvalue, workspace, P, rows, n_rows); break;
        case 4: hipLaunchKernelGGL((k_mlp_backward_t<A, ObsT, 4, FOLD>), grid, dim3(256), lds_bytes, stream, N, W, packed, (const ObsT *)obs, dlogits, dvalue, workspace, P, rows, n_rows); break;
        case 2: hipLaunchKernelGGL((k_mlp_backward_t<A, ObsT, 2, FOLD>), grid, dim3(128), lds_bytes, stream, N, W, packed, (const ObsT *)obs, dlogits, dvalue, workspace, P, rows, n_rows); break;
        default: hipLaunchKernelGGL((k_mlp_backward_t<A, ObsT, 1, FOLD>), grid, dim3(64), lds_bytes, stream, N, W, packed, (const ObsT *)obs, dlogits, dvalue, workspace, P, rows, n_rows); break;
    }
}

// Resident blocks per CU of the instantiation that a launch would use (register-limited: the LDS stage is small), asked of the
// runtime once per shape.  The grid is sized to exactly that many persistent blocks: more would run as a second, half-empty round.
template <int A, typename ObsT, bool FOLD>
static int occupancy_t(int waves, size_t lds_bytes) {
    int n = 0;
    hipError_t e;
    switch (waves) {
        case 8: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_mlp_backward_t<A, ObsT, 8, FOLD>, 512, lds_bytes); break;
        case 4: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_mlp_backward_t<A, ObsT, 4, FOLD>, 256, lds_bytes); break;
        case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_mlp_backward_t<A, ObsT, 2, FOLD>, 128, lds_bytes); break;
        default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_mlp_backward_t<A, ObsT, 1, FOLD>, 64, lds_bytes); break;
    }
    return (e == hipSuccess && n > 0) ? n : 1;
}

size_t mlp_backward_t_lds(int A, bool fold);

int mlp_backward_t_blocks_per_cu(int A, int waves, bool fold) {
    static int cache[2][RNAD_MAX_ACTIONS + 1][9] = {};
    if (A < 1 || A > RNAD_MAX_ACTIONS || waves < 1 || waves > 8) return 1;
    if (cache[fold][A][waves] == 0) {
        const size_t lds_bytes = mlp_backward_t_lds(A, fold);
        int n = 1;
        // (the fp16-observation instantiation uses the same registers)
        RNAD_DISPATCH_A(A, n = fold ? occupancy_t<kA, float, true>(waves, lds_bytes) : occupancy_t<kA, float, false>(waves, lds_bytes));
        cache[fold][A][waves] = n;
    }
    return cache[fold][A][waves];
}

size_t mlp_backward_t_lds(int A, bool fold) {
    const int K = fold ? mlp_fold_k(A) : 2 * A * A;
    return (size_t)2 * (kTile * bwd_stage_stride(K) + kTile + kTile * A + 4 * kTile) * sizeof(float);
}

int mlp_backward_t_launch(int A, int waves, dim3 grid, hipStream_t stream, int64_t N, int W, const float *packed, const void *obs,
                           int obs_half, const float *dlogits, const float *dvalue, float *workspace, int P, const int32_t *rows,
                           const int64_t *n_rows, bool fold) {
    const size_t lds_bytes = mlp_backward_t_lds(A, fold);
    RNAD_DISPATCH_A(A, {
        if (fold) {
            if (obs_half) launch_t<kA, __half, true>(waves, grid, lds_bytes, stream, N, W, packed, obs, dlogits, dvalue, workspace, P, rows, n_rows);
            else launch_t<kA, float, true>(waves, grid, lds_bytes, stream, N, W, packed, obs, dlogits, dvalue, workspace, P, rows, n_rows);
        } else {
            if (obs_half) launch_t<kA, __half, false>(waves, grid, lds_bytes, stream, N, W, packed, obs, dlogits, dvalue, workspace, P, rows, n_rows);
            else launch_t<kA, float, false>(waves, grid, lds_bytes, stream, N, W, packed, obs, dlogits, dvalue, workspace, P, rows, n_rows);
        }
    });
    return 0;
}

}  // namespace rnad_mlp
