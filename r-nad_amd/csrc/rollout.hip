// rollout.hip -- policy head, sampler (K3), transition (K2) and the rollout driver (gfx950).
//
// Replaces: nn/net.py:45-49,74-77 (masked exp-normalise, log-policy, multinomial), environment/episode.py:96-125
// (States.step) and :175-230 (Episodes.generate).  Citations are baskuit/R-NaD file:line.
#include "common.hpp"
#include "rollout_math.hpp"

#include <algorithm>

using namespace rnad;
using namespace rnad::dev;

namespace rnad {
int launch_observe(const rnad_tree_t *tree, int64_t B, const int32_t *idx, int player, void *obs, int obs_half, uint8_t *mbits,
                   float *maskf, hipStream_t stream);
}

namespace {

constexpr int kThreads = 256;

template <int A>
__global__ __launch_bounds__(kThreads) void k_policy_head(int64_t N, const float *__restrict__ logits,
                                                          const uint8_t *__restrict__ mbits, const float *__restrict__ maskf,
                                                          float *__restrict__ policy, float *__restrict__ log_policy) {
    const int64_t n = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (n >= N) return;
    float l[A], p[A], lp[A];
    uint32_t bits = 0;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        l[a] = logits[n * A + a];
        if (!mbits) bits |= (maskf[n * A + a] != 0.0f ? 1u : 0u) << a;
    }
    if (mbits) bits = mbits[n];
    policy_head_ptr<A>(l, bits, p, log_policy ? lp : nullptr);
#pragma unroll
    for (int a = 0; a < A; ++a) {
        policy[n * A + a] = p[a];
        if (log_policy) log_policy[n * A + a] = lp[a];
    }
}

__global__ __launch_bounds__(kThreads) void k_sample(int64_t B, int n, const float *__restrict__ probs,
                                                     const float *__restrict__ noise, uint64_t seed, int64_t lane0, int step,
                                                     int stream_id, int32_t *__restrict__ out) {
    const int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (b >= B) return;
    float p[RNAD_MAX_ACTIONS];
    load_n<RNAD_MAX_ACTIONS>(probs + b * n, n, p);
    if (noise) {  // explicit Exp(1) noise: torch's race
        float q[RNAD_MAX_ACTIONS];
        load_n<RNAD_MAX_ACTIONS>(noise + b * n, n, q);
        out[b] = race_argmax_n<RNAD_MAX_ACTIONS>(n, p, q);
    } else {  // seeded: the uniform of this decision (stream 0: the action of env step `step`; 1: the chance draw of its transition)
        float u[3];
        rnad_decision_uniforms(seed, (uint64_t)(lane0 + b), (uint32_t)step, u);
        out[b] = pick_n<RNAD_MAX_ACTIONS>(n, p, stream_id ? u[2] : u[step & 1]);
    }
}

template <int A>
__global__ __launch_bounds__(kThreads) void k_transition(const Trans *__restrict__ trans, int C, int64_t B,
                                                         const int32_t *__restrict__ idx, const int32_t *__restrict__ row_a,
                                                         const int32_t *__restrict__ col_a, const float *__restrict__ noise,
                                                         uint64_t seed, int64_t lane0, int step, int32_t *__restrict__ idx_out,
                                                         float *__restrict__ reward, int32_t *__restrict__ alive) {
    const int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    bool live = false;
    if (b < B) {
        int next;
        float rew;
        float u[3] = {0.0f, 0.0f, 0.0f};
        if (!noise && C > 1) rnad_decision_uniforms(seed, (uint64_t)(lane0 + b), (uint32_t)step, u);
        transition_lane<A>(trans, C, idx[b], row_a[b], col_a[b], noise ? noise + b * C : nullptr, u[2], next, rew);
        idx_out[b] = next;
        reward[b] = rew;
        live = next != 0;
    }
    if (alive) {  // one atomic per block
        __shared__ int part[kThreads / 64];
        const unsigned long long m = __ballot(live);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = (int)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            int s = 0;
#pragma unroll
            for (int i = 0; i < kThreads / 64; ++i) s += part[i];
            if (s) atomicAdd(alive, s);
        }
    }
}

// One env step of Episodes.generate for one lane (episode.py:196-212), everything except the net forward and
// the observation of the next step.  MODE 0: logits -> policy head -> sample; 1: policy given -> sample;
// 2: policy and action given (a net that samples for itself, net.py:49).
template <int A, int MODE>
__global__ __launch_bounds__(kThreads) void k_act(const Trans *__restrict__ trans, int C, int64_t B, int t,
                                                  const float *__restrict__ net_out, const int32_t *__restrict__ actions_in,
                                                  const float *__restrict__ value, const float *__restrict__ noise_a,
                                                  const float *__restrict__ noise_c, uint64_t seed, int64_t lane0,
                                                  const int32_t *__restrict__ idx_t, const uint8_t *__restrict__ mbits_t,
                                                  const int32_t *__restrict__ act_prev, float *__restrict__ policy_t,
                                                  int32_t *__restrict__ act_t, float *__restrict__ rewards_t,
                                                  float *__restrict__ values_t, int32_t *__restrict__ idx_next, int64_t table_S) {
    const int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (b < B) {
        float in[A], pol[A];
        // table_S != 0: net_out is a [2, S, A] table of the actor's logits per (player to move, state); the lane gathers its row
        const int64_t src = table_S ? (int64_t)(t & 1) * table_S + idx_t[b] : b;
#pragma unroll
        for (int a = 0; a < A; ++a) in[a] = net_out[src * A + a];
        if (MODE == 0) {
            policy_head_ptr<A>(in, mbits_t[b], pol, nullptr);
        } else {
#pragma unroll
            for (int a = 0; a < A; ++a) pol[a] = in[a];
        }
        int action;
        float u[3] = {0.0f, 0.0f, 0.0f};  // the seeded uniforms of this step's game transition (include/rnad_rng.h)
        if ((MODE != 2 && !noise_a) || ((t & 1) && !noise_c && C > 1)) rnad_decision_uniforms(seed, (uint64_t)(lane0 + b), (uint32_t)t, u);
        if (MODE == 2) {
            action = actions_in[b];
        } else if (noise_a) {  // explicit Exp(1) noise: torch's race
            float q[A];
#pragma unroll
            for (int a = 0; a < A; ++a) q[a] = noise_a[b * A + a];
            action = race_argmax<A>(pol, q);
        } else {
            action = pick<A>(pol, u[t & 1]);
        }
#pragma unroll
        for (int a = 0; a < A; ++a) policy_t[b * A + a] = pol[a];
        act_t[b] = action;
        values_t[b] = value ? value[src] : 0.0f;  // no value head in this rollout: the slot is defined, and unread (rnad.py never reads it)
        const int s = idx_t[b];
        int next = s;
        float rew = 0.0f;  // row turn: torch.zeros (episode.py:101)
        if (t & 1)
            transition_lane<A>(trans, C, s, act_prev[b], action, noise_c ? noise_c + b * C : nullptr, u[2], next, rew);
        idx_next[b] = next;
        rewards_t[b] = rew;
    }
}

// alive[t] = #lanes with indices[t, :] != 0.  grid = (chunks, T_cap + 1): each block counts one chunk of one row and
// issues ONE atomic, so a counter word sees at most `chunks` (<= 64) updates.
__global__ __launch_bounds__(kThreads) void k_count_alive(int64_t B, const int32_t *__restrict__ indices, int32_t *__restrict__ alive) {
    const int t = blockIdx.y;
    const int32_t *row = indices + (int64_t)t * B;
    int cnt = 0;
    for (int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x; b < B; b += (int64_t)gridDim.x * kThreads) cnt += row[b] != 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    __shared__ int part[kThreads / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
#pragma unroll
        for (int i = 0; i < kThreads / 64; ++i) s += part[i];
        if (s) atomicAdd(alive + t, s);
    }
}

// ---------------------------------------------------------------------------------------- live-row lists
// rows[] = the positions r with indices[r] != 0, in increasing order, and their count -- so that the MLP kernels can skip the
// (t, b) slots of absorbed episodes (ragged trees: rnad_mlp_forward_rows / rnad_mlp_backward_rows).  Small launches: per-chunk counts,
// [beyond kCompactOwnPrefix chunks: a one-block exclusive scan of the counts,] ordered write.  The order is fixed (not an atomic append), which
// keeps the weight-gradient sums of the backward pass reproducible.
constexpr int kCompactRows = 8;                           // rows of kThreads elements per block
constexpr int kCompactChunk = kCompactRows * kThreads;    // 2048 positions per block

__device__ __forceinline__ int wave_rank(bool flag, int &wave_total) {
    const uint64_t m = __ballot(flag);
    wave_total = __popcll(m);
    return __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
}

__global__ __launch_bounds__(kThreads) void k_compact_count(int64_t N, const int32_t *__restrict__ indices, int32_t *__restrict__ counts) {
    const int64_t base = (int64_t)blockIdx.x * kCompactChunk;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < kCompactRows; ++j) {
        const int64_t r = base + j * kThreads + threadIdx.x;
        cnt += (r < N && indices[r] != 0) ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    __shared__ int part[kThreads / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
#pragma unroll
        for (int i = 0; i < kThreads / 64; ++i) s += part[i];
        counts[blockIdx.x] = s;
    }
}

// counts[nb] -> exclusive prefix sums in place, total -> *n_rows.  One block; nb is N / 2048 (8192 for 2^24 positions).
__global__ __launch_bounds__(1024) void k_compact_scan(int nb, int32_t *__restrict__ counts, int64_t *__restrict__ n_rows) {
    __shared__ int64_t wave_sum[16];
    __shared__ int64_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nb ? counts[i] : 0;
        int incl = v;  // inclusive scan within the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if ((threadIdx.x & 63) >= off) incl += o;
        }
        if ((threadIdx.x & 63) == 63) wave_sum[threadIdx.x >> 6] = incl;
        __syncthreads();
        int64_t before = carry_s;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += wave_sum[w];
        if (i < nb) counts[i] = (int32_t)(before + incl - v);
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_rows = carry_s;
}

// OWN_PREFIX: `offsets` still holds the per-chunk COUNTS and every workgroup adds up the counts of the chunks before its own (a few
// loads per thread while there are at most kCompactOwnPrefix chunks) -- k_compact_scan's work without its launch; the last workgroup
// writes the total.
constexpr int kCompactOwnPrefix = 2048;
template <bool OWN_PREFIX>
__global__ __launch_bounds__(kThreads) void k_compact_write(int64_t N, const int32_t *__restrict__ indices,
                                                            const int32_t *__restrict__ offsets, int32_t *__restrict__ rows,
                                                            int64_t *__restrict__ n_rows) {
    __shared__ int totals[kCompactRows][kThreads / 64];
    __shared__ int prefix_s[kThreads / 64];
    if (OWN_PREFIX) {
        int sum = 0;
        for (int b = threadIdx.x; b < (int)blockIdx.x; b += kThreads) sum += offsets[b];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
        if ((threadIdx.x & 63) == 0) prefix_s[threadIdx.x >> 6] = sum;
    }
    const int64_t base = (int64_t)blockIdx.x * kCompactChunk;
    bool flag[kCompactRows];
    int rank[kCompactRows];
#pragma unroll
    for (int j = 0; j < kCompactRows; ++j) {
        const int64_t r = base + j * kThreads + threadIdx.x;
        flag[j] = r < N && indices[r] != 0;
        int tot;
        rank[j] = wave_rank(flag[j], tot);
        if ((threadIdx.x & 63) == 0) totals[j][threadIdx.x >> 6] = tot;
    }
    __syncthreads();
    int before;
    if (OWN_PREFIX) {
        before = 0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) before += prefix_s[w];
    } else {
        before = offsets[blockIdx.x];
    }
#pragma unroll
    for (int j = 0; j < kCompactRows; ++j) {
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) {
            if (w == (int)(threadIdx.x >> 6) && flag[j]) rows[before + rank[j]] = (int32_t)(base + j * kThreads + threadIdx.x);
            before += totals[j][w];
        }
    }
    if (OWN_PREFIX && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *n_rows = before;
}

__global__ void k_fill_i32(int64_t n, int32_t *p, int32_t v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + kThreads - 1) / kThreads); }

}  // namespace

extern "C" int rnad_policy_head(int64_t N, int A, const float *logits, const uint8_t *mask_bits, const float *mask,
                                float *policy, float *log_policy, void *stream) {
    RNAD_REQUIRE(logits && policy && (mask_bits || mask), "rnad_policy_head: null argument");
    if (N == 0) return 0;
    RNAD_DISPATCH_A(A, hipLaunchKernelGGL((k_policy_head<kA>), dim3(blocks_for(N)), dim3(kThreads), 0, (hipStream_t)stream, N,
                                          logits, mask_bits, mask, policy, log_policy));
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int rnad_sample(int64_t B, int n, const float *probs, const float *noise, uint64_t seed, int64_t lane0, int step,
                           int stream_id, int32_t *out, void *stream) {
    RNAD_REQUIRE(probs && out, "rnad_sample: null argument");
    RNAD_REQUIRE(n >= 1 && n <= RNAD_MAX_ACTIONS, "rnad_sample: %d categories out of range [1,%d]", n, RNAD_MAX_ACTIONS);
    if (B == 0) return 0;
    hipLaunchKernelGGL(k_sample, dim3(blocks_for(B)), dim3(kThreads), 0, (hipStream_t)stream, B, n, probs, noise, seed, lane0,
                       step, stream_id, out);
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int rnad_transition(const rnad_tree_t *tree, int64_t B, const int32_t *idx, const int32_t *row_actions,
                               const int32_t *col_actions, const float *noise, uint64_t seed, int64_t lane0, int step,
                               int32_t *idx_out, float *reward, int32_t *alive, void *stream) {
    RNAD_REQUIRE(tree && idx && row_actions && col_actions && idx_out && reward, "rnad_transition: null argument");
    if (B == 0) return 0;
    RNAD_DISPATCH_A(tree->A, hipLaunchKernelGGL((k_transition<kA>), dim3(blocks_for(B)), dim3(kThreads), 0, (hipStream_t)stream,
                                                tree->trans, tree->C, B, idx, row_actions, col_actions, noise, seed, lane0,
                                                step, idx_out, reward, alive));
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

static int check_traj(const rnad_tree_t *tree, const rnad_traj_t *tr, const char *who) {
    RNAD_REQUIRE(tree && tr, "%s: null argument", who);
    RNAD_REQUIRE(tr->indices && tr->observations && tr->mask_bits && tr->policy && tr->actions && tr->rewards && tr->values &&
                     tr->alive,
                 "%s: trajectory has a null buffer", who);
    RNAD_REQUIRE(tr->T_cap >= 1 && tr->B >= 1, "%s: bad trajectory shape T_cap=%d B=%lld", who, tr->T_cap, (long long)tr->B);
    return 0;
}

extern "C" int rnad_rollout_begin(const rnad_tree_t *tree, const rnad_traj_t *tr, void *stream_) {
    if (int rc = check_traj(tree, tr, "rnad_rollout_begin")) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    RNAD_REQUIRE(tr->B < ((int64_t)1 << 31), "rnad_rollout_begin: batch %lld too large", (long long)tr->B);
    hipLaunchKernelGGL(k_fill_i32, dim3(blocks_for(tr->B)), dim3(kThreads), 0, stream, tr->B, tr->indices, 1);  // root, episode.py:22
    RNAD_HIP_OK(hipGetLastError());
    return launch_observe(tree, tr->B, tr->indices, 0, tr->observations, tr->obs_half, tr->mask_bits, nullptr, stream);
}

static int rollout_step_impl(const rnad_tree_t *tree, const rnad_traj_t *tr, int t, int mode, const float *logits,
                             const float *policy_in, const int32_t *actions_in, const float *value, const float *noise_action,
                             const float *noise_chance, uint64_t seed, int64_t lane0, int64_t table_S, void *stream_) {
    if (int rc = check_traj(tree, tr, "rnad_rollout_step")) return rc;
    RNAD_REQUIRE(t >= 0 && t < tr->T_cap, "rnad_rollout_step: step %d outside [0,%d)", t, tr->T_cap);
    RNAD_REQUIRE(mode >= 0 && mode <= 2, "rnad_rollout_step: mode %d", mode);
    RNAD_REQUIRE(mode != 0 || logits, "rnad_rollout_step: mode 0 needs logits");
    RNAD_REQUIRE(mode == 0 || policy_in, "rnad_rollout_step: mode %d needs policy_in", mode);
    RNAD_REQUIRE(mode != 2 || actions_in, "rnad_rollout_step: mode 2 needs actions_in");
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t B = tr->B;
    const int A = tree->A;
    const float *net_out = mode == 0 ? logits : policy_in;
    const int32_t *idx_t = tr->indices + (int64_t)t * B;
    const int32_t *act_prev = t > 0 ? tr->actions + (int64_t)(t - 1) * B : tr->actions;
    {
        ProfScope prof(PROF_ACT, stream);
#define RNAD_ACT(MODE_)                                                                                                       \
    hipLaunchKernelGGL((k_act<kA, MODE_>), dim3(blocks_for(B)), dim3(kThreads), 0, stream, tree->trans, tree->C, B, t, net_out, \
                       actions_in, value, noise_action, noise_chance, seed, lane0, idx_t, tr->mask_bits + (int64_t)t * B,      \
                       act_prev, tr->policy + (int64_t)t * B * A, tr->actions + (int64_t)t * B, tr->rewards + (int64_t)t * B,  \
                       tr->values + (int64_t)t * B, tr->indices + (int64_t)(t + 1) * B, table_S)
        RNAD_DISPATCH_A(A, {
            if (mode == 0) RNAD_ACT(0);
            else if (mode == 1) RNAD_ACT(1);
            else RNAD_ACT(2);
        });
#undef RNAD_ACT
        RNAD_HIP_OK(hipGetLastError());
    }
    if (t + 1 < tr->T_cap) {
        const size_t esz = tr->obs_half ? 2 : 4;
        void *obs_next = (char *)tr->observations + (size_t)(t + 1) * B * 2 * A * A * esz;
        return launch_observe(tree, B, tr->indices + (int64_t)(t + 1) * B, (t + 1) & 1, obs_next, tr->obs_half,
                              tr->mask_bits + (int64_t)(t + 1) * B, nullptr, stream);
    }
    return 0;
}

extern "C" int rnad_rollout_step(const rnad_tree_t *tree, const rnad_traj_t *tr, int t, int mode, const float *logits,
                                 const float *policy_in, const int32_t *actions_in, const float *value,
                                 const float *noise_action, const float *noise_chance, uint64_t seed, int64_t lane0,
                                 void *stream) {
    return rollout_step_impl(tree, tr, t, mode, logits, policy_in, actions_in, value, noise_action, noise_chance, seed, lane0, 0, stream);
}

extern "C" int rnad_rollout_end(const rnad_tree_t *tree, const rnad_traj_t *tr, void *stream_) {
    if (int rc = check_traj(tree, tr, "rnad_rollout_end")) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = zero_async(tr->alive, sizeof(int32_t) * (tr->T_cap + 1), stream)) return rc;
    const unsigned chunks = (unsigned)std::min<int64_t>(64, blocks_for(tr->B));
    hipLaunchKernelGGL(k_count_alive, dim3(chunks, tr->T_cap + 1), dim3(kThreads), 0, stream, tr->B, tr->indices, tr->alive);
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

// Episodes.generate (episode.py:194-212) for a net that IS the fused MLP: all T_cap steps enqueued from one call.
// live_rows / n_live / block_counts (all three or none): from step 1 on the actor runs on the lanes that are still in the tree
// only (rnad_compact_valid of indices[t], then rnad_mlp_forward_rows).  Absorbed lanes keep the logits / value of their last
// live step: finite, and every consumer masks them (they sit in state 0, whose only transition is back to state 0).
extern "C" int rnad_rollout_run(const rnad_tree_t *tree, const rnad_traj_t *tr, int W, const float *packed, float *logits_ws,
                                int64_t logits_step_stride, float *value_ws, uint64_t seed, int64_t lane0, int32_t *live_rows,
                                int64_t *n_live, int32_t *block_counts, void *stream) {
    if (int rc = check_traj(tree, tr, "rnad_rollout_run")) return rc;
    RNAD_REQUIRE(packed && logits_ws, "rnad_rollout_run: null argument");
    const bool skip = live_rows != nullptr;
    RNAD_REQUIRE(skip == (n_live != nullptr) && skip == (block_counts != nullptr), "rnad_rollout_run: incomplete live-row workspace");
    RNAD_REQUIRE(!skip || logits_step_stride == 0, "rnad_rollout_run: keeping every step's logits needs the dense actor");
    if (int rc = rnad_rollout_begin(tree, tr, stream)) return rc;
    const size_t esz = tr->obs_half ? 2 : 4;
    const size_t step_bytes = (size_t)tr->B * 2 * tree->A * tree->A * esz;
    for (int t = 0; t < tr->T_cap; ++t) {
        const void *obs_t = (const char *)tr->observations + (size_t)t * step_bytes;
        float *logits_t = logits_ws + (int64_t)t * logits_step_stride;  // stride 0: one scratch row; B*A: keep every step's logits
        if (skip && t > 0) {
            if (int rc = rnad_compact_valid(tr->B, tr->indices + (int64_t)t * tr->B, live_rows, n_live, block_counts, stream)) return rc;
            if (int rc = rnad_mlp_forward_rows(tr->B, live_rows, n_live, tree->A, W, packed, obs_t, tr->obs_half, logits_t, value_ws, stream))
                return rc;
        } else {
            if (int rc = rnad_mlp_forward(tr->B, tree->A, W, packed, obs_t, tr->obs_half, logits_t, value_ws, stream)) return rc;
        }
        if (int rc = rnad_rollout_step(tree, tr, t, 0, logits_t, nullptr, nullptr, value_ws, nullptr, nullptr, seed, lane0, stream)) return rc;
    }
    return rnad_rollout_end(tree, tr, stream);
}

// ---------------------------------------------------------------------------------------- rnad_compact_valid
extern "C" int64_t rnad_compact_workspace(int64_t N) { return N < 0 ? -1 : (N + kCompactChunk - 1) / kCompactChunk + 1; }

extern "C" int rnad_compact_valid(int64_t N, const int32_t *indices, int32_t *rows, int64_t *n_rows, int32_t *block_counts, void *stream_) {
    RNAD_REQUIRE(N >= 0 && N < ((int64_t)1 << 31), "rnad_compact_valid: %lld positions do not fit an int32 row list", (long long)N);
    RNAD_REQUIRE(n_rows && (N == 0 || (indices && rows && block_counts)), "rnad_compact_valid: null argument");
    hipStream_t stream = (hipStream_t)stream_;
    if (N == 0) {
        if (int rc = zero_async(n_rows, sizeof(int64_t), stream)) return rc;
        return 0;
    }
    const int nb = (int)((N + kCompactChunk - 1) / kCompactChunk);
    hipLaunchKernelGGL(k_compact_count, dim3(nb), dim3(kThreads), 0, stream, N, indices, block_counts);
    if (nb <= kCompactOwnPrefix) {
        hipLaunchKernelGGL(k_compact_write<true>, dim3(nb), dim3(kThreads), 0, stream, N, indices, block_counts, rows, n_rows);
    } else {
        hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(1024), 0, stream, nb, block_counts, n_rows);
        hipLaunchKernelGGL(k_compact_write<false>, dim3(nb), dim3(kThreads), 0, stream, N, indices, block_counts, rows, n_rows);
    }
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------- tabular actor
// The observation of a lane is a function of (state, player to move) alone (episode.py:62-68), so an actor whose weights are
// fixed for the rollout can be evaluated ONCE on the 2S distinct observations (rnad_observe_all + rnad_mlp_forward) instead of
// on B lanes at each of the T steps; every step then gathers its logits row.  Same observation bits, same kernel: the policies,
// actions and trajectories are those of rnad_rollout_run bit for bit.  value_table [2S] (optional): the actor's value head on
// the same rows, gathered into traj->values; NULL fills it with zeros.
extern "C" int rnad_rollout_run_tabular(const rnad_tree_t *tree, const rnad_traj_t *tr, const float *logits_table,
                                        const float *value_table, uint64_t seed, int64_t lane0, void *stream) {
    if (int rc = check_traj(tree, tr, "rnad_rollout_run_tabular")) return rc;
    RNAD_REQUIRE(logits_table, "rnad_rollout_run_tabular: null table");
    if (int rc = rnad_rollout_begin(tree, tr, stream)) return rc;
    for (int t = 0; t < tr->T_cap; ++t)
        if (int rc = rollout_step_impl(tree, tr, t, 0, logits_table, nullptr, nullptr, value_table, nullptr, nullptr, seed, lane0, tree->S,
                                       stream))
            return rc;
    return rnad_rollout_end(tree, tr, stream);
}
