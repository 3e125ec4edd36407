// mlp_fwd.hip -- fused policy/value MLP forward on the fp32 matrix cores (gfx950).
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
// Both heads share the B operand; a wave walks 2 * W / 32 hidden tiles per 64 samples (two MFMA sample tiles per LDS read).
// Matrix-pipe time per sample tile = 2 * (W / 32) * (K / 2) * 64 cycles.
//
// Where the time goes (profiles/r01e_mlp_pmc.md): fp32 MFMA and fp32 VALU share the ALUs on gfx950, so a hidden tile pair
// costs 18 * 64 MFMA cycles + 32 v_max (4.7 cycles each) + the second-layer packed FMAs (6.4 each); that sum is 72 % matrix
// pipe, and the kernel measures 72 % of the fp32 MFMA peak AT THE CLOCK IT RUNS AT -- 2.04 GHz under this load, not 2.4
// (GRBM_GUI_ACTIVE / wall time; a bare MFMA loop holds 2.39 GHz).  LDS traffic, LDS latency and the input loads were each
// removed in turn (two sample tiles per weight read, software-pipelined operand reads, prefetched inputs) without moving the
// wall time: the kernel is ALU- and power-bound, not memory- or latency-bound.
#include "mlp_common.hpp"
#include "rollout_math.hpp"

using namespace rnad;
using namespace rnad_mlp;

namespace {

// First layer for ONE hidden tile and TWO 32-sample tiles: the A operand (weights) and the bias tile are read from LDS once
// and feed two independent accumulator chains.  The first-layer bias enters as the C operand of each chain's first MFMA
// (destination != source, so no register copies and no tenth MFMA).
template <int KS>
__device__ __forceinline__ void mfma_chain2(const float *__restrict__ lds, int W, int tile, int col, int half, const float (&x0)[KS],
                                            const float (&x1)[KS], f32x16 &c0, f32x16 &c1) {
    constexpr int K = 2 * KS;
    const float *wa = lds + tile * (KS * 64) + half * 32 + col;
    const float *brow = lds + img_b0(K, W) + tile * kTile + 4 * half;
    float a[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[ks] = wa[ks * 64];
    f32x16 bias;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4 *>(brow + 8 * g);
        bias[4 * g + 0] = b.x; bias[4 * g + 1] = b.y; bias[4 * g + 2] = b.z; bias[4 * g + 3] = b.w;
    }
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], x0[0], bias, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], x1[0], bias, 0, 0, 0);
#pragma unroll
    for (int ks = 1; ks < KS; ++ks) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], x0[ks], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], x1[ks], c1, 0, 0, 0);
    }
}

// HEADS: 1 = value only, 2 = policy only, 3 = both.  A wave owns 64 consecutive samples (two MFMA sample tiles) and walks the
// hidden tiles of the wanted heads one at a time: every LDS read (first-layer weights, second-layer weights) serves both
// sample tiles, which keeps the LDS pipe (shared by the four SIMDs) well below saturation.  The next 64 samples' inputs are
// loaded while the current ones are in the matrix pipe.  (Measured on gfx950: fp32 MFMA and fp32 VALU share the ALUs --
// tools/micro/mfma_peak.hip: every v_pk_fma_f32 adds 6.4 cycles to a 64-cycle MFMA -- so the VALU epilogue is kernel time and
// is kept to relu + the second-layer FMAs.)
// Up to four nets evaluated on the same inputs by one launch (blockIdx.y picks the net): the tabular update evaluates the
// learner, target and two regularisation nets on the same 2S observations, and at that size a launch is mostly latency.
struct NetSet {
    const float *packed[4];
    float *logits[4];
    float *value[4];
    int first_block[5];  // workgroups [first_block[i], first_block[i + 1]) serve net i
    // rnad_mlp_forward_actor: net 0 is a tabular ACTOR -- besides its logits, the policy head of every row (net.py:45-46 under the mover's
    // legal bits mask_tab[row]) goes out as a padded policy row, the table the bucketed rollout kernels gather from
    float *policy_rows;
    const uint8_t *mask_tab;
};

template <int A, typename ObsT, int HEADS, bool FOLD>
__global__ __launch_bounds__(kFwdThreads) void k_mlp_forward(int64_t N, int W, NetSet nets, const ObsT *__restrict__ obs,
                                                             const int32_t *__restrict__ rows, const int64_t *__restrict__ n_rows) {
    int net = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) net += (int)blockIdx.x >= nets.first_block[i] ? 1 : 0;
    const float *__restrict__ packed = nets.packed[net];
    float *__restrict__ logits = nets.logits[net];
    float *__restrict__ value = nets.value[net];
    const int64_t block = (int)blockIdx.x - nets.first_block[net], n_blocks = nets.first_block[net + 1] - nets.first_block[net];
    // rows != null: sample s of this launch is row rows[s] of obs / logits / value, and the sample count comes from device
    // memory (rnad_compact_valid's output: no host round trip between the compaction and this launch)
    if (n_rows) N = *n_rows;
    constexpr int K = MlpShape<A, FOLD>::K, KS = K / 2, OBS = MlpShape<A, FOLD>::OBS;  // KS: MFMA k-steps per hidden tile
    extern __shared__ __attribute__((aligned(16))) float lds[];
    load_image<kFwdThreads>(packed, lds, (FOLD ? img_floats_fold(K, W, A) : img_floats(K, W, A)) / 4);  // the image rnad_mlp_pack laid out, copied as is
    __syncthreads();
    if constexpr (FOLD) {  // the legal columns of the raw weights -> bias and indicator column (mlp_common.hpp)
        for (int h = threadIdx.x; h < 2 * W; h += kFwdThreads) {
            const float *lc = lds + img_legal(K, W, A) + h;
            float b, wi;
            fold_hidden_unit<A>([&](int k) { return lc[k * 2 * W]; }, lds[img_b0(K, W) + h], b, wi);
            lds[img_b0(K, W) + h] = b;
            constexpr int kk = A * A;  // the indicator's input slot
            lds[(h / kTile) * (KS * 64) + (kk / 2) * 64 + (kk % 2) * 32 + (h % kTile)] = wi;
        }
        __syncthreads();
    }
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

    constexpr int kSpan = 2 * kTile;  // samples per wave iteration
    const int64_t n_spans = (N + kSpan - 1) / kSpan;
    const int64_t span0 = block * (kFwdThreads / 64) + wave, dspan = n_blocks * (kFwdThreads / 64);
    float xn[2][KS];  // inputs of the next span, in flight during the current one
    auto fetch = [&](int64_t span) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int64_t sample = span * kSpan + s * kTile + col;
            const int64_t row = (rows && sample < N) ? (int64_t)rows[sample] : sample;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xn[s][ks] = sample < N ? obs_feature<A, FOLD, ObsT>(obs + row * OBS, 2 * ks + half) : 0.0f;
        }
    };
    if (span0 < n_spans) fetch(span0);
    for (int64_t span = span0; span < n_spans; span += dspan) {
        float x0[KS], x1[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { x0[ks] = xn[0][ks]; x1[ks] = xn[1][ks]; }
        if (span + dspan < n_spans) fetch(span + dspan);

        f32x2 acc_v[2][2], acc_p[2][A][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            acc_v[s][0] = acc_v[s][1] = f32x2{0.f, 0.f};
#pragma unroll
            for (int a = 0; a < A; ++a) acc_p[s][a][0] = acc_p[s][a][1] = f32x2{0.f, 0.f};
        }
        if ((HEADS & 1) && value) {  // (uniform: a net of the launch that does not want this head skips it)
            for (int t = 0; t < T; ++t) {
                f32x16 c0, c1;
                mfma_chain2<KS>(lds, W, t, col, half, x0, x1, c0, c1);
                epilogue_value(c0, w1v + t * kTile + 4 * half, acc_v[0]);
                epilogue_value(c1, w1v + t * kTile + 4 * half, acc_v[1]);
            }
        }
        if ((HEADS & 2) && logits) {
            for (int t = 0; t < T; ++t) {
                f32x16 c0, c1;
                mfma_chain2<KS>(lds, W, T + t, col, half, x0, x1, c0, c1);
                epilogue_policy<A>(c0, w1p + t * kTile + 4 * half, W, acc_p[0]);
                epilogue_policy<A>(c1, w1p + t * kTile + 4 * half, W, acc_p[1]);
            }
        }
        // lane-local sums, then the two half-waves (complementary hidden rows of the same 32 samples)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int64_t sample = span * kSpan + s * kTile + col;
            float out_v = (acc_v[s][0].x + acc_v[s][0].y) + (acc_v[s][1].x + acc_v[s][1].y), out_p[A];
#pragma unroll
            for (int a = 0; a < A; ++a) out_p[a] = (acc_p[s][a][0].x + acc_p[s][a][0].y) + (acc_p[s][a][1].x + acc_p[s][a][1].y);
            if (HEADS & 1) out_v += __shfl_xor(out_v, 32, 64);
            if (HEADS & 2) {
#pragma unroll
                for (int a = 0; a < A; ++a) out_p[a] += __shfl_xor(out_p[a], 32, 64);
            }
            if (sample < N && half == 0) {
                const int64_t row = rows ? (int64_t)rows[sample] : sample;
                if ((HEADS & 1) && value) value[row] = out_v + bv;
                if ((HEADS & 2) && logits) {
                    float lg[A];
#pragma unroll
                    for (int a = 0; a < A; ++a) logits[row * A + a] = lg[a] = out_p[a] + bp[a];
                    if (nets.policy_rows && net == 0) {  // (uniform) the same function of the same logits as k_policy_rows / k_row_records
                        constexpr int PS = (A + 3) & ~3;
                        float pol[PS];
                        rnad::dev::policy_head_ptr<A>(lg, nets.mask_tab[row], pol, nullptr);
#pragma unroll
                        for (int a = A; a < PS; ++a) pol[a] = 0.0f;
                        float4 *p4 = reinterpret_cast<float4 *>(nets.policy_rows + row * PS);
#pragma unroll
                        for (int u = 0; u < PS / 4; ++u) p4[u] = float4{pol[4 * u], pol[4 * u + 1], pol[4 * u + 2], pol[4 * u + 3]};
                    }
                }
            }
        }
    }
}


// Lay the eight torch Linear tensors out as the LDS image described at the top of this file; blockIdx.y picks the net (up to 4 per launch).
struct PackSet {
    const float *w[4][8];  // vw0, vb0, vw1, vb1, pw0, pb0, pw1, pb1 of each net
    float *packed[4];
};

__global__ __launch_bounds__(kThreads) void k_mlp_pack(int A, int W, PackSet ps, int total, int fold) {
    const float *const *w = ps.w[blockIdx.y];
    const float *__restrict__ vw0 = w[0], *__restrict__ vb0 = w[1], *__restrict__ vw1 = w[2], *__restrict__ vb1 = w[3];
    const float *__restrict__ pw0 = w[4], *__restrict__ pb0 = w[5], *__restrict__ pw1 = w[6], *__restrict__ pb1 = w[7];
    float *__restrict__ packed = ps.packed[blockIdx.y];
    // fold: the first-layer region holds the A^2 expected-value columns (the indicator slot and the padding stay zero: the kernels
    // fill the slot when they load the image), the raw legal columns follow the output biases
    const int OBS = 2 * A * A, K = fold ? ((A * A + 2) & ~1) : OBS, KS = K / 2;
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= total) return;
    float x = 0.0f;
    if (i < img_b0(K, W)) {
        const int tile = i / (KS * 64), rem = i % (KS * 64);
        const int ks = rem / 64, half = (rem % 64) / 32, col = rem % 32;
        const int h = tile * kTile + col, k = 2 * ks + half;
        if (!fold || k < A * A) x = h < W ? vw0[h * OBS + k] : pw0[(h - W) * OBS + k];
    } else if (fold && i >= img_legal(K, W, A)) {
        const int r = i - img_legal(K, W, A), k = r / (2 * W), h = r % (2 * W);
        x = h < W ? vw0[h * OBS + A * A + k] : pw0[(h - W) * OBS + A * A + k];
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

extern "C" int64_t rnad_mlp_packed_size(int A, int W) { return mlp_packed_floats(A, W); }

static int mlp_pack_launch(int n_nets, int A, int W, const float *const *weights, float *const *packed, int fold, void *stream);

extern "C" int rnad_mlp_pack_multi(int n_nets, int A, int W, const float *const *weights, float *const *packed, void *stream) {
    return mlp_pack_launch(n_nets, A, W, weights, packed, 0, stream);
}

extern "C" int64_t rnad_mlp_fold_packed_size(int A, int W) { return mlp_packed_floats_fold(A, W); }

extern "C" int rnad_mlp_pack_fold_multi(int n_nets, int A, int W, const float *const *weights, float *const *packed, void *stream) {
    RNAD_REQUIRE(A >= 2, "rnad_mlp_pack_fold: the legal fold needs at least two actions");
    return mlp_pack_launch(n_nets, A, W, weights, packed, 1, stream);
}

static int mlp_pack_launch(int n_nets, int A, int W, const float *const *weights, float *const *packed, int fold, void *stream) {
    RNAD_REQUIRE(n_nets >= 1 && n_nets <= 4 && weights && packed, "rnad_mlp_pack_multi: 1..4 nets");
    RNAD_REQUIRE(A >= 1 && A <= RNAD_MAX_ACTIONS && W >= kTile && W % kTile == 0, "rnad_mlp_pack: bad shape (A=%d, width=%d)", A, W);
    PackSet ps{};
    for (int i = 0; i < n_nets; ++i) {
        for (int j = 0; j < 8; ++j) {
            RNAD_REQUIRE(weights[8 * i + j], "rnad_mlp_pack: null weight tensor %d of net %d", j, i);
            ps.w[i][j] = weights[8 * i + j];
        }
        RNAD_REQUIRE(packed[i], "rnad_mlp_pack: null output %d", i);
        ps.packed[i] = packed[i];
    }
    const int total = fold ? mlp_packed_floats_fold(A, W) : mlp_packed_floats(A, W);
    hipLaunchKernelGGL(k_mlp_pack, dim3((total + kThreads - 1) / kThreads, n_nets), dim3(kThreads), 0, (hipStream_t)stream, A, W, ps, total, fold);
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int rnad_mlp_pack(int A, int W, const float *vw0, const float *vb0, const float *vw1, const float *vb1, const float *pw0,
                             const float *pb0, const float *pw1, const float *pb1, float *packed, void *stream) {
    const float *w[8] = {vw0, vb0, vw1, vb1, pw0, pb0, pw1, pb1};
    float *out[1] = {packed};
    return rnad_mlp_pack_multi(1, A, W, w, out, stream);
}

static int mlp_forward_launch(int64_t N, const int32_t *rows, const int64_t *n_rows, int A, int W, int n_nets, const NetSet &nets,
                              const void *obs, int obs_half, void *stream_, bool fold = false) {
    RNAD_REQUIRE(obs && n_nets >= 1 && n_nets <= 4, "rnad_mlp_forward: null argument");
    for (int i = 0; i < n_nets; ++i)
        RNAD_REQUIRE(nets.packed[i] && (nets.logits[i] || nets.value[i]), "rnad_mlp_forward: null argument (net %d)", i);
    RNAD_REQUIRE(W >= kTile && W % kTile == 0, "rnad_mlp_forward: width %d must be a positive multiple of %d", W, kTile);
    RNAD_REQUIRE(N >= 0, "rnad_mlp_forward: negative batch");
    if (N == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    RNAD_REQUIRE(!fold || A >= 2, "rnad_mlp_forward_fold: the legal fold needs at least two actions");
    const size_t lds_bytes = (size_t)(fold ? mlp_packed_floats_fold(A, W) : mlp_packed_floats(A, W)) * sizeof(float);
    RNAD_REQUIRE(lds_bytes <= 160 * 1024, "rnad_mlp_forward: weights (%zu B) do not fit the 160 KiB LDS (A=%d, width=%d)", lds_bytes, A, W);
    int dev = 0, cus = 256;
    RNAD_HIP_OK(hipGetDevice(&dev));
    RNAD_HIP_OK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    constexpr int kWaves = kFwdThreads / 64;
    const int blocks_per_cu = std::max(1, std::min(12 / kWaves, (int)(160 * 1024 / lds_bytes)));
    const int64_t n_spans = (N + 2 * kTile - 1) / (2 * kTile);  // a wave iteration covers 64 samples
    // every net of the launch gets its own full round of workgroups, in one linear grid (a launch per head set cost ~6 us more at
    // the tabular update's 132 862 rows; one persistent round split evenly / by cost over the nets: 59.8 / 64.5 us instead of 53;
    // workgroups whose waves split the hidden tiles of one span: no faster there and 46 % slower at 1.9 M rows)
    const int64_t per_net = std::max<int64_t>(1, std::min<int64_t>((n_spans + kWaves - 1) / kWaves, (int64_t)cus * blocks_per_cu));
    NetSet launch = nets;
    launch.first_block[0] = 0;
    for (int i = 0; i < 4; ++i) launch.first_block[i + 1] = i < n_nets ? launch.first_block[i] + (int)per_net : 0x7fffffff;
    const unsigned grid = (unsigned)launch.first_block[n_nets];
    // only the heads that are wanted are computed: the kernel is instantiated for their union over the nets of the launch
    int heads = 0;
    for (int i = 0; i < n_nets; ++i) heads |= (nets.value[i] ? 1 : 0) | (nets.logits[i] ? 2 : 0);
    ProfScope prof(PROF_MLP, stream);
#define RNAD_MLP_LAUNCH2(T_, H_)                                                                                                  \
    do {                                                                                                                           \
        auto kern = fold ? k_mlp_forward<kA, T_, H_, true> : k_mlp_forward<kA, T_, H_, false>;                                     \
        if (lds_bytes > 64 * 1024)                                                                                       \
            RNAD_HIP_OK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kFwdThreads), lds_bytes, stream, N, W, launch, (const T_ *)obs, rows, n_rows);            \
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

extern "C" int rnad_mlp_forward(int64_t N, int A, int W, const float *packed, const void *obs, int obs_half, float *logits, float *value,
                                void *stream) {
    NetSet nets{};
    nets.packed[0] = packed; nets.logits[0] = logits; nets.value[0] = value;
    return mlp_forward_launch(N, nullptr, nullptr, A, W, 1, nets, obs, obs_half, stream);
}

extern "C" int rnad_mlp_forward_multi(int n_nets, int64_t N, int A, int W, const float *const *packed, const void *obs, int obs_half,
                                      float *const *logits, float *const *value, void *stream) {
    RNAD_REQUIRE(n_nets >= 1 && n_nets <= 4 && packed && logits && value, "rnad_mlp_forward_multi: 1..4 nets");
    // ONE launch for all nets, instantiated for the union of the wanted heads; a net that does not want a head skips it at run time
    NetSet nets{};
    for (int i = 0; i < n_nets; ++i) {
        RNAD_REQUIRE(logits[i] || value[i], "rnad_mlp_forward_multi: net %d wants no output", i);
        nets.packed[i] = packed[i]; nets.logits[i] = logits[i]; nets.value[i] = value[i];
    }
    return mlp_forward_launch(N, nullptr, nullptr, A, W, n_nets, nets, obs, obs_half, stream);
}

extern "C" int rnad_mlp_forward_rows(int64_t max_rows, const int32_t *rows, const int64_t *n_rows, int A, int W, const float *packed,
                                     const void *obs, int obs_half, float *logits, float *value, void *stream) {
    RNAD_REQUIRE(rows && n_rows, "rnad_mlp_forward_rows: null row list");
    NetSet nets{};
    nets.packed[0] = packed; nets.logits[0] = logits; nets.value[0] = value;
    return mlp_forward_launch(max_rows, rows, n_rows, A, W, 1, nets, obs, obs_half, stream);
}

// The FOLD instantiation (mlp_common.hpp "the legal fold"): `packed` images from rnad_mlp_pack_fold_multi; every row of `obs` must carry
// an all-ones legal plane or e0 = [1, 0, ..., 0] (the caller checks its table once).  rows / n_rows: NULL, or a row list as in
// rnad_mlp_forward_rows (N is then the capacity of the list).
extern "C" int rnad_mlp_forward_fold(int n_nets, int64_t N, const int32_t *rows, const int64_t *n_rows, int A, int W,
                                     const float *const *packed, const void *obs, int obs_half, float *const *logits, float *const *value,
                                     void *stream) {
    RNAD_REQUIRE(n_nets >= 1 && n_nets <= 4 && packed && logits && value, "rnad_mlp_forward_fold: 1..4 nets");
    RNAD_REQUIRE(!rows == !n_rows, "rnad_mlp_forward_fold: rows and n_rows go together");
    NetSet nets{};
    for (int i = 0; i < n_nets; ++i) {
        RNAD_REQUIRE(logits[i] || value[i], "rnad_mlp_forward_fold: net %d wants no output", i);
        nets.packed[i] = packed[i]; nets.logits[i] = logits[i]; nets.value[i] = value[i];
    }
    return mlp_forward_launch(N, rows, n_rows, A, W, n_nets, nets, obs, obs_half, stream, true);
}

// A tabular ACTOR on (a row list of) the tree's 2S observations: the policy head of one net -> its logits [2S, A] AND its policy rows
// [2S, rnad_bucket_policy_row_stride(A)] (the masked exp-normalise of net.py:45-46 under the mover's legal bits, in the kernel's
// epilogue: what rnad_bucket_sort / rnad_bucket_play otherwise take from the logits with a launch of their own per call).  fold != 0:
// `packed` is the FOLD image.  rows / n_rows: NULL = all 2S rows.
extern "C" int rnad_mlp_forward_actor(const rnad_tree_t *tree, const int32_t *rows, const int64_t *n_rows, int W, int fold, const float *packed,
                                      const void *obs, int obs_half, float *logits, float *policy_rows, void *stream) {
    RNAD_REQUIRE(tree && packed && obs && logits && policy_rows, "rnad_mlp_forward_actor: null argument");
    RNAD_REQUIRE(!rows == !n_rows, "rnad_mlp_forward_actor: rows and n_rows go together");
    RNAD_REQUIRE(((uintptr_t)policy_rows & 15) == 0, "rnad_mlp_forward_actor: policy_rows must be 16-byte aligned");
    NetSet nets{};
    nets.packed[0] = packed; nets.logits[0] = logits; nets.value[0] = nullptr;
    nets.policy_rows = policy_rows;
    nets.mask_tab = tree->mask_tab;
    return mlp_forward_launch(2 * tree->S, rows, n_rows, tree->A, W, 1, nets, obs, obs_half, stream, fold != 0);
}
