// mlp_rows.hip -- the nets of a tabular update on the (player, state) rows of the tree AND the row records, in one launch (gfx950).
//
// The default step evaluates the learner net (both heads) and the target net (value head) on the 2S observations of the tree
// (learn/rnad.py:373,378 on every distinct input) and then turns the five net-output tables into the row records of the bucketed
// update (rnad_bucket_records: policy heads, process_policy, log_policy_reg, the fast records, the actor's policy rows).  At
// 2S = 132 862 rows both are mostly latency: k_mlp_forward gives every 4-wave workgroup a 45 KB weight image for ONE 64-row span per
// wave, and k_row_records is a launch of ~550 dependent instructions per row that waits for its loads and stores.  Here:
//   * one persistent workgroup per CU: W / 32 COMPUTE waves, one per hidden tile -- wave w owns hidden tile w of the learner's value
//     head, of the target's value head and of the learner's policy head; its first-layer weights (the A operands of its MFMA chains)
//     stay in REGISTERS for the whole launch: there is no weight image in LDS, only the folded first-layer biases and the second-layer
//     rows -- and 4 RECORD waves;
//   * a workgroup takes a contiguous run of 32-row tiles (their count differs by at most one between workgroups) in steps of 64 rows
//     (an odd last tile is a half step), 4 steps to a chunk.  Phase 1 (compute waves): the partial second-layer sums of the wave's tiles
//     for the rows of the chunk, parked in LDS.  Phase 2 (record waves, one step each, WHILE the compute waves are in phase 1 of the
//     next chunk: the matrix pipe never waits for the latency-bound record arithmetic): the W / 32 partial sums added up in wave order
//     (+ the output bias) complete logits / v / v_target; then the row's records with the very function k_row_records uses
//     (row_records.hpp) -- same bits for the same logits.  One barrier per chunk, two buffers of partial sums.
// MODE 1 is the lazy-rows variant (learn/rnad.py: _value_tables): the learner's logits already exist (the staged actor wrote them), only
// the two value heads are evaluated, on a listed subset of the rows.  MODE 2 is that staged ACTOR (rnad_mlp_rows_actor): the policy
// head of one net on a row list, its logits and -- from phase 2 -- its policy rows (net.py:45-46 under the mover's legal bits), no
// records: on configs[3] k_mlp_forward gives every workgroup a 112 KB weight image (one workgroup per CU) for each of the three
// staging launches.
// Summation order of the second layer (this kernel's own, as k_mlp_forward's is its own; nothing in the reference fixes it): per lane
// and hidden tile as epilogue_value / epilogue_policy, the two half-waves, then the tiles in ascending order, then the bias.
#include "mlp_common.hpp"
#include <type_traits>
#include "rollout_math.hpp"
#include "row_records.hpp"

using namespace rnad;
using namespace rnad::dev;
using namespace rnad_mlp;

namespace {

constexpr int kRowsMaxWaves = 8;  // compute waves: W <= 256
constexpr int kRecWaves = 4;      // record waves
constexpr int kChunkSteps = 4;    // 64-row steps per chunk: one per record wave
// A/B switch for tools/ (what the kernel's time is made of): 1 = no records in phase 2, 2 = no MFMA chains / epilogues in phase 1
#ifndef RNAD_ROWS_ABLATE
#define RNAD_ROWS_ABLATE 0
#endif

struct RowsArgs {
    const float *packed_net, *packed_target;  // weight images (rnad_mlp_pack / _pack_fold)
    float *logit, *v, *v_target;              // tables [2S, A], [2S], [2S]: written (logit: read when the policy head is not evaluated here)
    const float *logit_reg, *logit_reg_;      // [2S, A] logits of the two regularisation nets
    const uint8_t *mask_tab;
    float *rec, *fast, *pol_rows;             // record tables (rec / pol_rows may be NULL)
    const rnad_step_params_t *sp;
    const int32_t *rows;                      // optional row list (count in device memory)
    const int64_t *n_rows;
};

// First layer of one hidden tile for two (TWO) or one 32-row tiles: A operands from registers, the (folded) bias tile from LDS as the
// C operand of each chain's first MFMA.
template <int KS, bool TWO>
__device__ __forceinline__ void chain_reg(const float (&a)[KS], const float *__restrict__ brow, const float (&x0)[KS], const float (&x1)[KS],
                                          f32x16 &c0, f32x16 &c1) {
    f32x16 bias;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4 *>(brow + 8 * g);
        bias[4 * g + 0] = b.x; bias[4 * g + 1] = b.y; bias[4 * g + 2] = b.z; bias[4 * g + 3] = b.w;
    }
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], x0[0], bias, 0, 0, 0);
    if constexpr (TWO) c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], x1[0], bias, 0, 0, 0);
#pragma unroll
    for (int ks = 1; ks < KS; ++ks) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], x0[ks], c0, 0, 0, 0);
        if constexpr (TWO) c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], x1[ks], c1, 0, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------- r06: split-precision first layer
// fp32 MFMA runs at 1/16 of the bf16 rate on gfx950 (64 against 1024 FLOP per clock and SIMD) and shares the ALUs with the VALU work of
// the epilogues.  SPLIT: every fp32 operand as the exact sum of three bf16 numbers, x = x_h + x_m + x_l (round to nearest each time:
// x_h = bf16(x), x_m = bf16(x - x_h), x_l = bf16(x - x_h - x_m); the differences are exact, 3 x 9 significant bits cover the 24 of a
// float), and W x as SIX v_mfma_f32_32x32x16_bf16 products per 16 input features -- W_l x_h, W_h x_l, W_m x_m, W_m x_h, W_h x_m, W_h x_h,
// small terms first -- accumulated in fp32 by the matrix core.  Every bf16 x bf16 product is exact in fp32; what is dropped (W_m x_l,
// W_l x_m, W_l x_l) is below 2^-24 |W| |x| in total, i.e. below the rounding of the fp32 chain it replaces: the same function to fp32
// accuracy in another summation order (tests/test_hip_rows.py: 1e-5 / 3e-6 against k_mlp_forward as before).  6 x 32 cycles per 16
// features against 8 x 64 (A = 3 with the fold: 5 x 64), and the bf16 matrix pipe leaves the VALU free for the epilogues.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
struct Split8 {
    bf16x8 h, m, l;
};
__device__ __forceinline__ Split8 split8(const float (&v)[8]) {
    Split8 o;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const bf16x2 h = __builtin_convertvector(f32x2{v[i], v[i + 1]}, bf16x2);
        const float r0 = v[i] - (float)h[0], r1 = v[i + 1] - (float)h[1];  // exact
        const bf16x2 m = __builtin_convertvector(f32x2{r0, r1}, bf16x2);
        const float q0 = r0 - (float)m[0], q1 = r1 - (float)m[1];  // exact
        const bf16x2 l = __builtin_convertvector(f32x2{q0, q1}, bf16x2);
        o.h[i] = h[0]; o.h[i + 1] = h[1];
        o.m[i] = m[0]; o.m[i + 1] = m[1];
        o.l[i] = l[0]; o.l[i + 1] = l[1];
    }
    return o;
}

// chain_reg on split operands: KB blocks of 16 input features
template <int KB, bool TWO>
__device__ __forceinline__ void chain_split(const Split8 (&a)[KB], const float *__restrict__ brow, const Split8 (&x0)[KB], const Split8 (&x1)[KB],
                                            f32x16 &c0, f32x16 &c1) {
    f32x16 bias;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4 *>(brow + 8 * g);
        bias[4 * g + 0] = b.x; bias[4 * g + 1] = b.y; bias[4 * g + 2] = b.z; bias[4 * g + 3] = b.w;
    }
    c0 = bias;
    if constexpr (TWO) c1 = bias;
#if RNAD_ROWS_ABLATE & 4
    c0[0] += (float)x0[0].h[0]; if constexpr (TWO) c1[0] += (float)x1[0].h[0];
    return;
#endif
#define RNAD_SPLIT_MFMA(wa_, xb_)                                                                      \
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb].wa_, x0[kb].xb_, c0, 0, 0, 0);                  \
    if constexpr (TWO) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb].wa_, x1[kb].xb_, c1, 0, 0, 0)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        RNAD_SPLIT_MFMA(l, h);
        RNAD_SPLIT_MFMA(h, l);
        RNAD_SPLIT_MFMA(m, m);
    }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        RNAD_SPLIT_MFMA(m, h);
        RNAD_SPLIT_MFMA(h, m);
    }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        RNAD_SPLIT_MFMA(h, h);
    }
#undef RNAD_SPLIT_MFMA
}

__device__ __forceinline__ float lane_sum(const f32x2 (&acc)[2]) { return (acc[0].x + acc[0].y) + (acc[1].x + acc[1].y); }

template <bool B>
struct Flag { static constexpr bool value = B; };

// MODE 0: learner (both heads) + target (value head) -> tables + records;  1: the two value heads (logits from the table) -> tables +
// records;  2: the learner's policy head -> logits + policy rows
// WIDE (SPLIT with more than one block of 16 input features -- A >= 4 -- or with all three heads, MODE 0, whose 12-wave build needs scratch:
// a build with scratch wrote wrong records): no dedicated record waves -- the first kChunkSteps compute waves write the
// records of the previous chunk after their own phase 1 (r04 measured that arrangement 3 % behind dedicated record waves) -- so that a
// workgroup is 8 waves, two per SIMD, and a wave may hold 256 registers: split weights (12 per hidden tile and block) and the split inputs of
// two row tiles do not fit the 168 of a 12-wave workgroup.
template <int A, bool FOLD, bool SPLIT, int MODE>
constexpr bool rows_wide() { return SPLIT && ((MlpShape<A, FOLD>::K + 15) / 16 > 1 || MODE == 0); }

template <int A, typename ObsT, bool FOLD, int MODE, bool SPLIT = false>
__global__ __launch_bounds__(64 * (kRowsMaxWaves + (rows_wide<A, FOLD, SPLIT, MODE>() ? 0 : kRecWaves))) void k_rows_forward_records(int64_t N, int W, RowsArgs g, const ObsT *__restrict__ obs,
                                                                                           rnad_learn_params_t hp) {
    if (g.n_rows) N = *g.n_rows;
    if (g.sp) {
        hp.alpha = g.sp->alpha;
        hp.one_minus_alpha = g.sp->one_minus_alpha;
    }
    constexpr int K = MlpShape<A, FOLD>::K, KS = K / 2, OBS = MlpShape<A, FOLD>::OBS;
    constexpr bool POLICY = MODE != 1, VALUES = MODE != 2;
    constexpr int NV = VALUES ? 2 : 0;                 // value heads: learner, target
    constexpr int NOUT = NV + (POLICY ? A : 0);        // partial sums per row: learner value | target value | learner logits
    constexpr int U = NV + (POLICY ? 1 : 0);           // hidden tiles per compute wave
    constexpr int U0 = VALUES ? 0 : 2;                 // first of them in the order learner value (0), target value (1), learner policy (2)
    const int T = W / kTile;                  // hidden tiles per head = compute waves of this workgroup
    constexpr bool WIDE = rows_wide<A, FOLD, SPLIT, MODE>();
    constexpr bool STAGED = SPLIT;  // the chunk's observation rows go through LDS, loaded once per workgroup (below)
    const int nthreads = 64 * (T + (WIDE ? 0 : kRecWaves));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, half = lane >> 5;
    const bool computes = wave < T;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *b0f = lds;                // [3][W] first-layer biases (FOLD: with the legal columns folded in)
    float *w1 = lds + 3 * W;         // [2 + A][W] second-layer rows: learner value | target value | learner policy [A]
    float *part = w1 + (2 + A) * W;  // [2][kChunkSteps][T][NOUT][64] partial sums; before the first step: the indicator weights [U][W]
    const int part_buf = kChunkSteps * T * NOUT * 64;
    // STAGED: the observation rows of a chunk, staged ONCE per workgroup -- [2 buffers][kChunkRows][XS] floats behind the partial sums.  r06,
    // measured: every compute wave fetching its own copy of the rows (8 waves x 10 scattered 4-byte loads per step) kept the CU's vector L1
    // busy 0.87 of the launch (TCP_TOTAL_CACHE_ACCESSES 23.8 M per configs[3] launch, the most of any kernel of the step) and was what
    // the launch waited for: the split first layer alone changed nothing (profiles/r06_experiments.md).
    constexpr int kChunkRows = kChunkSteps * 2 * kTile;
    constexpr int KBx = (K + 15) / 16, XS = 16 * KBx + 4;  // floats per staged row: the features padded to whole blocks of 16, + 4 (bank spread of the 32-byte reads)
    float *xl = part + 2 * part_buf;

    constexpr int KB = (K + 15) / 16;          // SPLIT: blocks of 16 input features (zero beyond K)
    constexpr int KSr = SPLIT ? 1 : KS, KBs = SPLIT ? KB : 1;
    Split8 aw[U][KBs];  // SPLIT: a compute wave's A operands, W0[hidden = 32 tile + col][k = 16 kb + 8 half + j] as three bf16 each
    float a[U][KSr];  // a compute wave's A operands: W0[hidden = 32 tile + col][k = 2 ks + half]
    if (computes && !SPLIT) {
#pragma unroll
        for (int v = 0; v < U; ++v) {
            const int u = U0 + v;
            const float *img = u == 1 ? g.packed_target : g.packed_net;
            const int tile = u == 2 ? T + wave : wave;
            const float *wa = img + tile * (KS * 64) + half * 32 + col;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a[v][ks] = wa[ks * 64];
        }
    }
    for (int i = threadIdx.x + U0 * W; i < (U0 + U) * W; i += nthreads) {
        const int u = i / W, h = i % W;
        const float *img = u == 1 ? g.packed_target : g.packed_net;
        const int hs = u == 2 ? W + h : h;  // row of the stacked [2W] first layer
        float b = img[img_b0(K, W) + hs], wi = 0.0f;
        if constexpr (FOLD) {
            const float *lc = img + img_legal(K, W, A) + hs;
            fold_hidden_unit<A>([&](int k) { return lc[k * 2 * W]; }, b, b, wi);
        }
        b0f[i] = b;
        part[i] = wi;
    }
    for (int i = threadIdx.x + (VALUES ? 0 : 2 * W); i < (POLICY ? 2 + A : 2) * W; i += nthreads) {
        float x;
        if (i < W) x = g.packed_net[img_w1v(K, W) + i];
        else if (i < 2 * W) x = g.packed_target[img_w1v(K, W) + i - W];
        else x = g.packed_net[img_w1p(K, W) + i - 2 * W];
        w1[i] = x;
    }
    __syncthreads();
    if constexpr (FOLD && !SPLIT) {
        constexpr int kk = A * A;  // the indicator's input slot
        if (computes && half == (kk & 1)) {
#pragma unroll
            for (int v = 0; v < U; ++v) a[v][kk / 2] = part[(U0 + v) * W + wave * kTile + col];
        }
    }
    if constexpr (SPLIT) {
        if (computes) {
#pragma unroll
            for (int v = 0; v < U; ++v) {
                const int u = U0 + v;
                const float *img = u == 1 ? g.packed_target : g.packed_net;
                const int tile = u == 2 ? T + wave : wave;
                const float *wa = img + tile * (KS * 64) + col;
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    float w8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        // (k depends on the lane's half: both candidates are compile-time slots of the image)
                        const int k0 = 16 * kb + j, k1 = k0 + 8;
                        auto slot = [&](int k) -> float {
                            if (FOLD && k == A * A) return part[(U0 + v) * W + wave * kTile + col];  // the indicator's weight
                            if (k >= (FOLD ? A * A : K)) return 0.0f;
                            return wa[(k / 2) * 64 + (k % 2) * 32];
                        };
                        const float lo_half = slot(k0), hi_half = slot(k1);
                        w8[j] = half ? hi_half : lo_half;
                    }
                    aw[v][kb] = split8(w8);
                }
            }
        }
    }
    __syncthreads();  // the scratch in `part` is free

    // this workgroup's rows: a contiguous run of 32-row tiles
    const int64_t n_tiles = (N + kTile - 1) / kTile;
    const int64_t base = n_tiles / gridDim.x, rem = n_tiles % gridDim.x;
    const int64_t tile0 = (int64_t)blockIdx.x * base + ((int64_t)blockIdx.x < rem ? (int64_t)blockIdx.x : rem);
    const int64_t my_tiles = base + ((int64_t)blockIdx.x < rem ? 1 : 0);
    const int64_t s_begin = tile0 * kTile, s_end_ = (tile0 + my_tiles) * kTile, s_end = s_end_ < N ? s_end_ : N;
    const int n_steps = (int)((my_tiles + 1) / 2);
    const int n_chunks = (n_steps + kChunkSteps - 1) / kChunkSteps;

    // STAGED: chunk c's rows -> registers (stage_load, at the start of the iteration before: the loads travel while the chunk before is
    // computed) -> LDS buffer c & 1 (stage_store, at the end of that iteration, in front of its barrier).  Thread i takes the elements
    // i, i + nthreads, ... of the chunk's [kChunkRows][K] feature matrix: consecutive threads read consecutive floats of a row.
    constexpr int kStageThreads = 64 * (kRowsMaxWaves + (WIDE ? 0 : kRecWaves));  // (staged workgroups have kRowsMaxWaves compute waves: width 256)
    const bool stages = true;
    const int stage_tid = (int)threadIdx.x;
    constexpr int kStageMax = STAGED ? (kChunkRows * K + kStageThreads - 1) / kStageThreads : 1;
    float xst[kStageMax];
    auto stage_load = [&](int c) {
        const int64_t first = s_begin + (int64_t)c * kChunkRows;
#pragma unroll
        for (int e = 0; e < kStageMax; ++e) {
            const int i = stage_tid + e * kStageThreads;
            const int r = i / K, k = i % K;
            const int64_t sample = first + r;
            float v = 0.0f;
            if (i < kChunkRows * K && sample < s_end) {
                const int64_t row = g.rows ? (int64_t)g.rows[sample] : sample;
                v = obs_feature<A, FOLD, ObsT>(obs + row * OBS, k);
            }
            xst[e] = v;
        }
    };
    auto stage_store = [&](int c) {
        float *dst = xl + (c & 1) * (kChunkRows * XS);
#pragma unroll
        for (int e = 0; e < kStageMax; ++e) {
            const int i = stage_tid + e * kStageThreads;
            if (i < kChunkRows * K) dst[(i / K) * XS + (i % K)] = xst[e];
        }
    };
    if constexpr (STAGED) {
        for (int i = threadIdx.x; i < 2 * kChunkRows * XS; i += nthreads) xl[i] = 0.0f;  // (the features beyond K stay zero)
        __syncthreads();
        if (n_chunks > 0 && stages) {
            stage_load(0);
            stage_store(0);
        }
        __syncthreads();
    }

    // B operands of the next step, in flight during the current one: x[row = 32 s + col][2 ks + half]; SPLIT: [16 kb + 8 half + j]
    constexpr int XN = SPLIT ? 8 * KB : KS;
    float xn[2][XN];
    auto fetch = [&](int step) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int64_t sample = s_begin + (int64_t)step * (2 * kTile) + s * kTile + col;
            const bool in = sample < s_end;
            int64_t row = sample;
            if constexpr (!STAGED) row = (in && g.rows) ? (int64_t)g.rows[sample] : sample;
            if constexpr (STAGED) {
                // (the staged rows: 32 contiguous bytes per lane and block of 16 features)
                const int local = (step % kChunkSteps) * (2 * kTile) + s * kTile + col;
                const float *rp = xl + ((step / kChunkSteps) & 1) * (kChunkRows * XS) + local * XS + 8 * half;
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    const float4 lo4 = *reinterpret_cast<const float4 *>(rp + 16 * kb), hi4 = *reinterpret_cast<const float4 *>(rp + 16 * kb + 4);
                    xn[s][8 * kb + 0] = lo4.x; xn[s][8 * kb + 1] = lo4.y; xn[s][8 * kb + 2] = lo4.z; xn[s][8 * kb + 3] = lo4.w;
                    xn[s][8 * kb + 4] = hi4.x; xn[s][8 * kb + 5] = hi4.y; xn[s][8 * kb + 6] = hi4.z; xn[s][8 * kb + 7] = hi4.w;
                }
                (void)in; (void)row;
            } else if constexpr (SPLIT) {
                const ObsT *rp = obs + row * OBS;
                constexpr int EV = FOLD ? A * A : K;  // features read from the row as they are
                float ind = 0.0f;                     // FOLD: the indicator feature (obs_feature)
                if constexpr (FOLD) ind = in ? 1.0f - load_obs<ObsT>(rp + A * A + (A > 1 ? 1 : 0)) : 0.0f;
#pragma unroll
                for (int i = 0; i < XN; ++i) {
                    const int k0 = 16 * (i / 8) + (i % 8), k1 = k0 + 8;  // the two half-waves' features of this slot
                    if (k0 >= K && k1 >= K) {
                        xn[s][i] = 0.0f;  // (beyond K for both: a constant)
                        continue;
                    }
                    const int k = half ? k1 : k0;
                    float v = (in && k < EV) ? load_obs<ObsT>(rp + k) : 0.0f;
                    if constexpr (FOLD) v = k == A * A ? ind : v;
                    xn[s][i] = v;
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) xn[s][ks] = in ? obs_feature<A, FOLD, ObsT>(obs + row * OBS, 2 * ks + half) : 0.0f;
            }
        }
    };
    const float *bias_u = b0f + wave * kTile + 4 * half;
    const float *w1_u = w1 + wave * kTile + 4 * half;
    // phase 1 of one step: this wave's hidden tiles on the step's 64 (TWO) or 32 rows -> partial sums into `dst`
    // the heads of this wave's hidden tiles on one (PAIR = false: x0 / xs0, into out[.][o0]) or two row tiles (into out[.][0], out[.][1])
    auto heads = [&](const float (&x0)[KSr], const float (&x1)[KSr], const Split8 (&xs0)[KBs], const Split8 (&xs1)[KBs], float (&out)[NOUT][2],
                     auto pair_, int o0) {
        constexpr bool PAIR = decltype(pair_)::value;
#pragma unroll
        for (int u = 0; u < NV; ++u) {  // learner value, target value
            f32x16 c0_, c1_;
            f32x2 acc0[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}}, acc1[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
            if constexpr (SPLIT) chain_split<KBs, PAIR>(aw[u], bias_u + u * W, xs0, xs1, c0_, c1_);
            else chain_reg<KSr, PAIR>(a[u], bias_u + u * W, x0, x1, c0_, c1_);
            epilogue_value(c0_, w1_u + u * W, acc0);
            if constexpr (PAIR) epilogue_value(c1_, w1_u + u * W, acc1);
            out[u][o0] = lane_sum(acc0);
            if constexpr (PAIR) out[u][1] = lane_sum(acc1);
        }
        if constexpr (POLICY) {
            f32x16 c0_, c1_;
            f32x2 acc0[A][2], acc1[A][2];
#pragma unroll
            for (int a_ = 0; a_ < A; ++a_) acc0[a_][0] = acc0[a_][1] = acc1[a_][0] = acc1[a_][1] = f32x2{0.f, 0.f};
            if constexpr (SPLIT) chain_split<KBs, PAIR>(aw[U - 1], bias_u + 2 * W, xs0, xs1, c0_, c1_);
            else chain_reg<KSr, PAIR>(a[U - 1], bias_u + 2 * W, x0, x1, c0_, c1_);
            epilogue_policy<A>(c0_, w1_u + 2 * W, W, acc0);
            if constexpr (PAIR) epilogue_policy<A>(c1_, w1_u + 2 * W, W, acc1);
#pragma unroll
            for (int a_ = 0; a_ < A; ++a_) {
                out[NV + a_][o0] = lane_sum(acc0[a_]);
                if constexpr (PAIR) out[NV + a_][1] = lane_sum(acc1[a_]);
            }
        }
    };
    auto p1_step = [&](int step, float *dst, auto two_) {
        constexpr bool TWO = decltype(two_)::value;
        float out[NOUT][2];  // [output][row tile]
#pragma unroll
        for (int o = 0; o < NOUT; ++o) out[o][0] = out[o][1] = 0.0f;
        float x0[KSr], x1[KSr];
        Split8 xs0[KBs], xs1[KBs];
        if constexpr (STAGED) fetch(step);  // (from the staged rows in LDS: no prefetch across steps)
        {
            if constexpr (SPLIT) {
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    float t0[8], t1[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { t0[j] = xn[0][8 * kb + j]; t1[j] = xn[1][8 * kb + j]; }
                    xs0[kb] = split8(t0);
                    if constexpr (TWO) xs1[kb] = split8(t1);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) { x0[ks] = xn[0][ks]; x1[ks] = xn[1][ks]; }
            }
            if constexpr (!STAGED)
                if (step + 1 < n_steps) fetch(step + 1);
#if RNAD_ROWS_ABLATE & 2
#pragma unroll
            for (int o = 0; o < NOUT; ++o) out[o][0] = out[o][1] = (SPLIT ? (float)xs0[0].h[0] + (float)xs1[0].l[1] : x0[0] + x1[0]);
#else
            // (r06: issuing the next head's products ahead of this head's epilogue -- two more accumulators -- measured no gain and spilled)
            heads(x0, x1, xs0, xs1, out, two_, 0);
#endif
        }
        // the two half-waves hold complementary hidden rows of the same 32 + 32 rows: half h keeps row tile h and gets the other
        // half's share of it -- lane l ends up with the tile's sums for row 64 step + l
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const float keep = half ? out[o][1] : out[o][0], send = half ? out[o][0] : out[o][1];
            dst[o * 64] = keep + __shfl_xor(send, 32, 64);
        }
    };
    // phase 2 of one step: complete the sums of its 64 rows, then their records (MODE 2: their policy rows)
    auto p2_step = [&](int step, const float *src) {
        const int64_t sample = s_begin + (int64_t)step * (2 * kTile) + lane;
        if (sample >= s_end) return;
        const int64_t row = g.rows ? (int64_t)g.rows[sample] : sample;
        const uint32_t bits = g.mask_tab[row];
        float lr[A], lr2[A], lg[A];
        if constexpr (VALUES) {
#pragma unroll
            for (int a_ = 0; a_ < A; ++a_) {
                lr[a_] = g.logit_reg[row * A + a_];
                lr2[a_] = g.logit_reg_[row * A + a_];
                if (!POLICY) lg[a_] = g.logit[row * A + a_];
            }
        }
        float sum[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) sum[o] = src[o * 64];
        for (int w = 1; w < T; ++w) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) sum[o] += src[(w * NOUT + o) * 64];
        }
        const float *b1n = g.packed_net + img_b1(K, W, A);  // [1 + A]: value bias, policy biases
        if constexpr (POLICY) {
#pragma unroll
            for (int a_ = 0; a_ < A; ++a_) g.logit[row * A + a_] = lg[a_] = sum[NV + a_] + b1n[1 + a_];
        }
        if constexpr (VALUES) {
            const float *b1t = g.packed_target + img_b1(K, W, A);
            const float vr = sum[0] + b1n[0], vtr = sum[1] + b1t[0];
            g.v[row] = vr;
            g.v_target[row] = vtr;
#if RNAD_ROWS_ABLATE & 1
            if (bits == 0x12345u) g.fast[row] = lr[0] + lr2[0] + lg[0];
#else
            write_row_records<A>(row, lg, vr, vtr, lr, lr2, bits, hp, g.rec, g.fast, g.pol_rows);
#endif
        } else {  // the same function of the same logits as k_policy_rows / k_row_records
            constexpr int PSTR = (A + 3) & ~3;
            float pol[PSTR];
            rnad::dev::policy_head_ptr<A>(lg, bits, pol, nullptr);
#pragma unroll
            for (int a_ = A; a_ < PSTR; ++a_) pol[a_] = 0.0f;
            float4 *p4 = reinterpret_cast<float4 *>(g.pol_rows + row * PSTR);
#pragma unroll
            for (int u = 0; u < PSTR / 4; ++u) p4[u] = float4{pol[4 * u], pol[4 * u + 1], pol[4 * u + 2], pol[4 * u + 3]};
        }
    };

    if (!STAGED && computes && n_steps > 0) fetch(0);
    // iteration c: the compute waves are in phase 1 of chunk c, the record waves in phase 2 of chunk c - 1 (the other buffer)
    for (int c = 0; c <= n_chunks; ++c) {
        if constexpr (STAGED)
            if (c + 1 < n_chunks && stages) stage_load(c + 1);
        if (computes) {
            if (c < n_chunks) {
                float *buf = part + (c & 1) * part_buf + (int64_t)wave * NOUT * 64 + lane;
                const int first = c * kChunkSteps, last = first + kChunkSteps < n_steps ? first + kChunkSteps : n_steps;
                for (int step = first; step < last; ++step) {
                    float *dst = buf + (int64_t)(step - first) * T * NOUT * 64;
                    if (2 * step + 1 < my_tiles) p1_step(step, dst, Flag<true>{});
                    else p1_step(step, dst, Flag<false>{});
                }
            }
        } else if (c > 0) {
            const int step = (c - 1) * kChunkSteps + (wave - T);
            if (step < n_steps) p2_step(step, part + ((c - 1) & 1) * part_buf + (int64_t)(wave - T) * T * NOUT * 64 + lane);
        }
        if constexpr (WIDE) {  // (the buffer of chunk c - 1 is written again in iteration c + 1, behind the barrier below)
            if (c > 0 && wave < kChunkSteps) {
                const int step = (c - 1) * kChunkSteps + wave;
                if (step < n_steps) p2_step(step, part + ((c - 1) & 1) * part_buf + (int64_t)wave * T * NOUT * 64 + lane);
            }
        }
        if constexpr (STAGED)
            if (c + 1 < n_chunks && stages) stage_store(c + 1);
        __syncthreads();
    }
}

size_t rows_lds_bytes(int A, int W, int mode, int fold = 0, bool split = false) {
    const int T = W / kTile, nout = (mode != 2 ? 2 : 0) + (mode != 1 ? A : 0);
    const int K = fold ? ((A * A + 2) & ~1) : 2 * A * A;
    const size_t staged = split ? (size_t)2 * (kChunkSteps * 2 * kTile) * (16 * ((K + 15) / 16) + 4) : 0;  // (two chunks of staged rows)
    return ((size_t)3 * W + (size_t)(2 + A) * W + (size_t)2 * kChunkSteps * T * nout * 64 + staged) * sizeof(float);
}

}  // namespace

// Shapes whose instantiation keeps everything in registers under the 12-wave budget (168 VGPRs; hipcc 7.2 spills beyond): with the
// policy head A <= 3, without it A <= 5 (FOLD) / 4.  Other shapes take the two launches this one replaces.
extern "C" int rnad_mlp_rows_records_supported(int A, int W, int fold, int policy_from_table) {
    if (A < 1 || A > RNAD_MAX_ACTIONS || W < kTile || W % kTile != 0 || W / kTile > kRowsMaxWaves) return 0;
    if (fold && A < 2) return 0;
    if (rows_lds_bytes(A, W, policy_from_table ? 1 : 0) > 150 * 1024) return 0;
    return policy_from_table ? A <= (fold ? 5 : 4) : A <= 3;
}

// the staged actor's launch (MODE 2): A <= 5 with the fold, A <= 3 without (registers, as above)
extern "C" int rnad_mlp_rows_actor_supported(int A, int W, int fold) {
    if (A < 1 || A > RNAD_MAX_ACTIONS || W < kTile || W % kTile != 0 || W / kTile > kRowsMaxWaves) return 0;
    if (fold && A < 2) return 0;
    return A <= (fold ? 5 : 3);
}

static int rows_launch(const rnad_tree_t *tree, int W, int fold, const void *obs, int obs_half, int mode, const RowsArgs &g,
                       const rnad_learn_params_t &hp, hipStream_t stream) {
    const int A = tree->A, T = W / kTile;
    const int64_t N = 2 * tree->S;
    int dev = 0, cus = 256;
    RNAD_HIP_OK(hipGetDevice(&dev));
    RNAD_HIP_OK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t n_tiles = (N + kTile - 1) / kTile;
    const int64_t grid = std::max<int64_t>(1, std::min<int64_t>((n_tiles + 1) / 2, cus));  // one persistent workgroup per CU
    ProfScope prof(PROF_MLP, stream);
    // r06: the split-precision first layer on rows staged in LDS (chain_split, STAGED).
    // RNAD_MLP_SPLIT: 0 = the fp32 MFMA chains everywhere, 1 = the split first layer wherever an instantiation without scratch exists; unset:
    // where it pays -- first layers of more than 16 input features (A >= 4 with the fold: configs[3]'s value heads 61.6 -> 53.9 us).  At
    // A = 3 it is a tie on all 132 862 rows of configs[1] (44.7 against 45.4 us) and a loss on the 13 676 distinct ones (13.7 against 11.8:
    // one step per workgroup, all prologue), so the default step of configs[1] keeps the r05 kernel.
    const char *split_e = getenv("RNAD_MLP_SPLIT");
    const int K_in = fold ? ((A * A + 2) & ~1) : 2 * A * A;
    const bool split_wanted = split_e ? atoi(split_e) != 0 : (K_in + 15) / 16 > 1;
    const bool split = split_wanted && A >= 2 && A <= 5 && fold && T == kRowsMaxWaves &&
                       rows_lds_bytes(A, W, mode, fold, true) <= 160 * 1024;
    const size_t lds_bytes = rows_lds_bytes(A, W, mode, fold, split);
#define RNAD_ROWS_LAUNCH4(T_, F_, M_, S_)                                                                                              \
    do {                                                                                                                               \
        auto kern = k_rows_forward_records<kA, T_, F_, M_, S_>;                                                                        \
        if (lds_bytes > 64 * 1024)                                                                                                     \
            RNAD_HIP_OK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));          \
        constexpr bool wide_ = rows_wide<kA, F_, S_, M_>();                                                                                \
        RNAD_REQUIRE(!S_ || T == kRowsMaxWaves, "rnad_mlp_rows: the split first layer needs a width of %d", kRowsMaxWaves * kTile); \
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * (T + (wide_ ? 0 : kRecWaves))), lds_bytes, stream, N, W, g, (const T_ *)obs, hp); \
    } while (0)
#define RNAD_ROWS_LAUNCH3(T_, F_, M_)                                                   \
    do {                                                                                \
        if constexpr (kA >= 2 && kA <= 5) {                                             \
            if (split) RNAD_ROWS_LAUNCH4(T_, F_, M_, true);                             \
            else RNAD_ROWS_LAUNCH4(T_, F_, M_, false);                                  \
        } else {                                                                        \
            RNAD_ROWS_LAUNCH4(T_, F_, M_, false);                                       \
        }                                                                               \
    } while (0)
#define RNAD_ROWS_LAUNCH2(T_, F_)                           \
    do {                                                    \
        if (mode == 0) RNAD_ROWS_LAUNCH3(T_, F_, 0);        \
        else if (mode == 1) RNAD_ROWS_LAUNCH3(T_, F_, 1);   \
        else RNAD_ROWS_LAUNCH3(T_, F_, 2);                  \
    } while (0)
#define RNAD_ROWS_LAUNCH(T_)                           \
    do {                                               \
        if (fold) RNAD_ROWS_LAUNCH2(T_, true);         \
        else RNAD_ROWS_LAUNCH2(T_, false);             \
    } while (0)
    RNAD_DISPATCH_A(A, {
        if (obs_half)
            RNAD_ROWS_LAUNCH(__half);
        else
            RNAD_ROWS_LAUNCH(float);
    });
#undef RNAD_ROWS_LAUNCH4
#undef RNAD_ROWS_LAUNCH3
#undef RNAD_ROWS_LAUNCH2
#undef RNAD_ROWS_LAUNCH
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int rnad_mlp_rows_records(const rnad_tree_t *tree, int W, int fold, const float *packed_net, const float *packed_target,
                                     const void *obs, int obs_half, const int32_t *rows, const int64_t *n_rows, int policy_from_table,
                                     float *logit_tab, float *v_tab, float *v_target_tab, const float *logit_reg_tab,
                                     const float *logit_reg_tab_, const rnad_learn_params_t *hp, const rnad_step_params_t *device_params,
                                     float *records, float *fast_records, float *policy_rows, void *stream_) {
    RNAD_REQUIRE(tree && packed_net && packed_target && obs && logit_tab && v_tab && v_target_tab && logit_reg_tab && logit_reg_tab_ && hp && fast_records,
                 "rnad_mlp_rows_records: null argument");
    RNAD_REQUIRE(!rows == !n_rows, "rnad_mlp_rows_records: rows and n_rows go together");
    RNAD_REQUIRE(rnad_mlp_rows_records_supported(tree->A, W, fold, policy_from_table),
                 "rnad_mlp_rows_records: shape not supported (A=%d, width=%d, fold=%d, policy_from_table=%d): see rnad_mlp_rows_records_supported",
                 tree->A, W, fold, policy_from_table);
    RNAD_REQUIRE((((uintptr_t)records | (uintptr_t)fast_records | (uintptr_t)policy_rows) & 15) == 0, "rnad_mlp_rows_records: record tables must be 16-byte aligned");
    RowsArgs g{};
    g.packed_net = packed_net; g.packed_target = packed_target;
    g.logit = logit_tab; g.v = v_tab; g.v_target = v_target_tab;
    g.logit_reg = logit_reg_tab; g.logit_reg_ = logit_reg_tab_;
    g.mask_tab = tree->mask_tab;
    g.rec = records; g.fast = fast_records; g.pol_rows = policy_rows;
    g.sp = device_params; g.rows = rows; g.n_rows = n_rows;
    return rows_launch(tree, W, fold, obs, obs_half, policy_from_table ? 1 : 0, g, *hp, (hipStream_t)stream_);
}

// The staged actor of a tree that is large next to the batch (rnad_mlp_forward_actor's job with this file's mapping): the policy head of one
// net on the listed rows of the tree's observation table -> logits [2S, A] and policy rows [2S, rnad_bucket_policy_row_stride(A)].
extern "C" int rnad_mlp_rows_actor(const rnad_tree_t *tree, const int32_t *rows, const int64_t *n_rows, int W, int fold, const float *packed,
                                   const void *obs, int obs_half, float *logits, float *policy_rows, void *stream_) {
    RNAD_REQUIRE(tree && packed && obs && logits && policy_rows, "rnad_mlp_rows_actor: null argument");
    RNAD_REQUIRE(!rows == !n_rows, "rnad_mlp_rows_actor: rows and n_rows go together");
    RNAD_REQUIRE(rnad_mlp_rows_actor_supported(tree->A, W, fold), "rnad_mlp_rows_actor: shape not supported (A=%d, width=%d, fold=%d)", tree->A, W, fold);
    RNAD_REQUIRE(((uintptr_t)policy_rows & 15) == 0, "rnad_mlp_rows_actor: policy_rows must be 16-byte aligned");
    RowsArgs g{};
    g.packed_net = packed; g.packed_target = packed;
    g.logit = logits;
    g.mask_tab = tree->mask_tab;
    g.pol_rows = policy_rows;
    g.rows = rows; g.n_rows = n_rows;
    rnad_learn_params_t hp{};
    return rows_launch(tree, W, fold, obs, obs_half, 2, g, hp, (hipStream_t)stream_);
}
