// tree.hip -- tree tables in HBM and K1, the episode-gather kernel (gfx950).
//
// Replaces: environment/tree.py:115-146 (the per-state tensor schema, as the thing kernels read) and
// environment/episode.py:62-68,208 (States.observations + masks).  Citations are baskuit/R-NaD file:line.
#include "common.hpp"

#include <algorithm>
#include <cstring>
#include <mutex>

namespace rnad {

static thread_local std::string g_error;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
}

// ---------------------------------------------------------------------------------------- profiling
static unsigned g_prof_mask = 0;  // bit k: bracket launches of kernel k with events
struct ProfPair {
    hipEvent_t a, b;
};
static std::vector<ProfPair> g_prof[PROF_COUNT];

namespace {
__global__ __launch_bounds__(256) void k_zero_words(uint32_t *__restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
}  // namespace

int zero_async(void *ptr, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return 0;
    if (((uintptr_t)ptr & 3) || (bytes & 3)) {
        set_error("zero_async: pointer / size not 4-byte aligned");
        return 2;
    }
    const size_t n = bytes / 4;
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_zero_words, dim3(grid), dim3(256), 0, stream, (uint32_t *)ptr, n);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

ProfScope::ProfScope(int which_, hipStream_t stream_) : which(which_), stream(stream_) {
    if (!((g_prof_mask >> which_) & 1u)) return;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;  // timing events do not belong in a captured graph
    if (hipStreamIsCapturing(stream_, &capturing) != hipSuccess || capturing != hipStreamCaptureStatusNone) return;
    if (hipEventCreate(&start) != hipSuccess) {
        start = nullptr;
        return;
    }
    (void)hipEventRecord(start, stream);
}

ProfScope::~ProfScope() {
    if (!start) return;
    hipEvent_t stop;
    if (hipEventCreate(&stop) != hipSuccess) return;
    (void)hipEventRecord(stop, stream);
    g_prof[which].push_back({start, stop});
}

}  // namespace rnad

using namespace rnad;

extern "C" const char *rnad_last_error(void) { return g_error.c_str(); }
extern "C" int rnad_version(void) { return 1; }

extern "C" int rnad_prof_enable(int on /* bit mask, -1 = all */) {
    for (auto &v : g_prof) {
        for (auto &p : v) {
            (void)hipEventDestroy(p.a);
            (void)hipEventDestroy(p.b);
        }
        v.clear();
    }
    g_prof_mask = (unsigned)on;
    return 0;
}

extern "C" int rnad_prof_read(int which, int64_t *launches, double *total_ms) {
    RNAD_REQUIRE(which >= 0 && which < PROF_COUNT, "rnad_prof_read: kernel id %d out of range", which);
    RNAD_HIP_OK(hipDeviceSynchronize());
    double ms = 0.0;
    for (auto &p : g_prof[which]) {
        float x = 0.f;
        RNAD_HIP_OK(hipEventElapsedTime(&x, p.a, p.b));
        ms += x;
    }
    if (launches) *launches = (int64_t)g_prof[which].size();
    if (total_ms) *total_ms = ms;
    return 0;
}

// ---------------------------------------------------------------------------------------- tree tables
extern "C" int rnad_tree_create(rnad_tree_t **out, int64_t S, int C, int A, const int64_t *index, const float *value,
                                const float *chance, const float *expected_value, const float *legal, int device) {
    RNAD_REQUIRE(out && index && value && chance && expected_value && legal, "rnad_tree_create: null argument");
    RNAD_REQUIRE(A >= 1 && A <= RNAD_MAX_ACTIONS, "rnad_tree_create: max_actions %d out of range [1,%d]", A, RNAD_MAX_ACTIONS);
    RNAD_REQUIRE(C >= 1 && C <= RNAD_MAX_TRANSITIONS, "rnad_tree_create: max_transitions %d out of range [1,%d]", C,
                 RNAD_MAX_TRANSITIONS);
    RNAD_REQUIRE(S >= 2 && S < (int64_t)1 << 31, "rnad_tree_create: tree size %lld must be in [2, 2^31)", (long long)S);
    const int AA = A * A, NS = node_stride_floats(A);

    std::vector<float> node((size_t)S * NS, 0.0f);
    std::vector<Trans> trans((size_t)S * AA * C);
    for (int64_t s = 0; s < S; ++s) {
        uint64_t bits = 0;
        for (int k = 0; k < AA; ++k) {
            node[(size_t)s * NS + k] = expected_value[s * AA + k];
            if (legal[s * AA + k] != 0.0f) bits |= (uint64_t)1 << k;
        }
        const uint32_t lo = (uint32_t)bits, hi = (uint32_t)(bits >> 32);
        memcpy(&node[(size_t)s * NS + AA], &lo, 4);
        memcpy(&node[(size_t)s * NS + AA + 1], &hi, 4);
        for (int t = 0; t < C; ++t)
            for (int rc = 0; rc < AA; ++rc) {
                const int64_t src = (s * C + t) * AA + rc;
                RNAD_REQUIRE(index[src] >= 0 && index[src] < S, "rnad_tree_create: index[%lld] = %lld outside [0,%lld)",
                             (long long)src, (long long)index[src], (long long)S);
                trans[((size_t)s * AA + rc) * C + t] = Trans{(int32_t)index[src], chance[src], value[src]};
            }
    }

    // depth levels below the root (state 1); the index tensor is increasing (tree.py:368-383), so one ascending
    // pass sees every parent before its children.
    auto *tree = new rnad_tree;
    tree->S = S; tree->C = C; tree->A = A; tree->NS = NS; tree->device = device;
    tree->level_of.assign((size_t)S, -1);
    tree->level_of[1] = 0;
    int max_level = 0;
    for (int64_t s = 1; s < S; ++s) {
        const int lv = tree->level_of[s];
        if (lv < 0) continue;
        for (int k = 0; k < AA * C; ++k) {
            const Trans &e = trans[(size_t)s * AA * C + k];
            if (e.next != 0 && e.chance > 0.0f) {
                if (e.next <= s) {
                    delete tree;
                    set_error("rnad_tree_create: index tensor is not increasing at state %lld", (long long)s);
                    return 2;
                }
                tree->level_of[e.next] = lv + 1;
                max_level = std::max(max_level, lv + 1);
            }
        }
    }
    tree->n_levels = max_level + 1;
    tree->max_depth = max_level + 1;
    // uniform_length: the only way out of the tree is from its deepest level.  Then every lane is live for exactly
    // 2 * max_depth env steps and there is nothing for the live-row lists to skip.
    tree->uniform_length = true;
    for (int64_t s = 1; s < S && tree->uniform_length; ++s) {
        const int lv = tree->level_of[s];
        if (lv < 0) continue;
        for (int k = 0; k < AA * C; ++k) {
            const Trans &e = trans[(size_t)s * AA * C + k];
            if (e.chance > 0.0f && (e.next == 0) != (lv == max_level)) { tree->uniform_length = false; break; }
        }
    }
    std::vector<int64_t> count(tree->n_levels + 1, 0);
    for (int64_t s = 1; s < S; ++s)
        if (tree->level_of[s] >= 0) count[tree->level_of[s] + 1]++;
    for (int l = 0; l < tree->n_levels; ++l) count[l + 1] += count[l];
    tree->level_offsets = count;
    std::vector<int32_t> order((size_t)std::max<int64_t>(count[tree->n_levels], 1));
    {
        std::vector<int64_t> cur(count.begin(), count.end() - 1);
        for (int64_t s = 1; s < S; ++s)
            if (tree->level_of[s] >= 0) order[cur[tree->level_of[s]]++] = (int32_t)s;
    }

    auto fail = [&](hipError_t e, const char *what) {
        set_error("rnad_tree_create: %s failed: %s", what, hipGetErrorString(e));
        rnad_tree_destroy(tree);
        return 1;
    };
    hipError_t e;
    DeviceGuard guard(device);  // the caller's current device is restored on return
    if (!guard.ok) return fail(hipErrorInvalidDevice, "hipSetDevice");
    if ((e = hipMalloc((void **)&tree->node, node.size() * sizeof(float))) != hipSuccess) return fail(e, "hipMalloc(node)");
    if ((e = hipMalloc((void **)&tree->trans, trans.size() * sizeof(Trans))) != hipSuccess) return fail(e, "hipMalloc(trans)");
    if ((e = hipMalloc((void **)&tree->level_order, order.size() * sizeof(int32_t))) != hipSuccess)
        return fail(e, "hipMalloc(level_order)");
    if ((e = hipMemcpy(tree->node, node.data(), node.size() * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess)
        return fail(e, "hipMemcpy(node)");
    if ((e = hipMemcpy(tree->trans, trans.data(), trans.size() * sizeof(Trans), hipMemcpyHostToDevice)) != hipSuccess)
        return fail(e, "hipMemcpy(trans)");
    if ((e = hipMemcpy(tree->level_order, order.data(), order.size() * sizeof(int32_t), hipMemcpyHostToDevice)) != hipSuccess)
        return fail(e, "hipMemcpy(level_order)");
    // hot states: whole levels from the root down while 2 * n_hot * (A + 1) doubles fit 96 KiB of LDS (rnad_learn_fused_tabular)
    {
        const int64_t cap = (96 * 1024) / (2 * (A + 1) * (int64_t)sizeof(double));
        int64_t n_hot = 0;
        for (int l = 0; l < tree->n_levels && count[l + 1] <= cap; ++l) n_hot = count[l + 1];
        tree->n_hot = (int)n_hot;
        std::vector<int32_t> slot((size_t)S, -1);
        for (int64_t k = 0; k < n_hot; ++k) slot[(size_t)order[(size_t)k]] = (int32_t)k;
        if ((e = hipMalloc((void **)&tree->hot_slot, slot.size() * sizeof(int32_t))) != hipSuccess) return fail(e, "hipMalloc(hot_slot)");
        if ((e = hipMemcpy(tree->hot_slot, slot.data(), slot.size() * sizeof(int32_t), hipMemcpyHostToDevice)) != hipSuccess)
            return fail(e, "hipMemcpy(hot_slot)");
    }
    // bucketed tabular pipeline: level-order position, depth and mover's legal bits per state; subtree sizes per level
    {
        std::vector<int32_t> pos((size_t)S, -1);
        std::vector<uint8_t> lev((size_t)S, 255);
        for (int64_t k = 0; k < count[tree->n_levels]; ++k) pos[(size_t)order[(size_t)k]] = (int32_t)k;
        for (int64_t s = 1; s < S; ++s)
            if (tree->level_of[s] >= 0) lev[(size_t)s] = (uint8_t)std::min(tree->level_of[s], 254);
        std::vector<uint8_t> mtab((size_t)2 * S, 0);
        for (int64_t s = 0; s < S; ++s)
            for (int P = 0; P < 2; ++P) {
                uint32_t mb = 0;
                for (int i = 0; i < A; ++i) {
                    const int src = P ? i : i * A;  // observations[:, 1, i, 0] of the mover's view (k_observe)
                    if (legal[s * AA + src] != 0.0f) mb |= 1u << i;
                }
                mtab[(size_t)P * S + s] = (uint8_t)mb;
            }
        // subtree extents: children have larger ids, so a descending pass sees every child before its parent
        std::vector<int64_t> end((size_t)S), desc((size_t)S, 0);
        for (int64_t s = 0; s < S; ++s) end[(size_t)s] = s + 1;
        bool contiguous = true;
        for (int64_t s = S - 1; s >= 1; --s) {
            if (tree->level_of[s] < 0) continue;
            for (int k = 0; k < AA * C; ++k) {
                const Trans &e = trans[(size_t)s * AA * C + k];
                if (e.next != 0 && e.chance > 0.0f) {
                    end[(size_t)s] = std::max(end[(size_t)s], end[(size_t)e.next]);
                    desc[(size_t)s] += 1 + desc[(size_t)e.next];
                }
            }
            if (end[(size_t)s] - s != 1 + desc[(size_t)s]) contiguous = false;
        }
        tree->contiguous_subtrees = contiguous;
        tree->subtree_size.assign((size_t)S, 0);
        tree->child_offsets.assign((size_t)S + 1, 0);
        for (int64_t s = 1; s < S; ++s) {
            if (tree->level_of[s] >= 0) tree->subtree_size[(size_t)s] = (int32_t)(1 + desc[(size_t)s]);
            int64_t n = 0;
            if (tree->level_of[s] >= 0)
                for (int k = 0; k < AA * C; ++k) {
                    const Trans &e = trans[(size_t)s * AA * C + k];
                    n += (e.next != 0 && e.chance > 0.0f) ? 1 : 0;
                }
            tree->child_offsets[(size_t)s + 1] = n;
        }
        for (int64_t s = 0; s < S; ++s) tree->child_offsets[(size_t)s + 1] += tree->child_offsets[(size_t)s];
        tree->children.resize((size_t)tree->child_offsets[(size_t)S]);
        for (int64_t s = 1; s < S; ++s) {
            if (tree->level_of[s] < 0) continue;
            int64_t w = tree->child_offsets[(size_t)s];
            for (int k = 0; k < AA * C; ++k) {
                const Trans &e = trans[(size_t)s * AA * C + k];
                if (e.next != 0 && e.chance > 0.0f) tree->children[(size_t)w++] = e.next;
            }
            std::sort(tree->children.begin() + tree->child_offsets[(size_t)s], tree->children.begin() + w);
        }
        tree->level_max_subtree.assign((size_t)tree->n_levels, 1);
        for (int64_t s = 1; s < S; ++s)
            if (tree->level_of[s] >= 0)
                tree->level_max_subtree[(size_t)tree->level_of[s]] = std::max(tree->level_max_subtree[(size_t)tree->level_of[s]], 1 + desc[(size_t)s]);
        if ((e = hipMalloc((void **)&tree->order_pos, pos.size() * sizeof(int32_t))) != hipSuccess) return fail(e, "hipMalloc(order_pos)");
        if ((e = hipMalloc((void **)&tree->level_dev, lev.size())) != hipSuccess) return fail(e, "hipMalloc(level_dev)");
        if ((e = hipMalloc((void **)&tree->mask_tab, mtab.size())) != hipSuccess) return fail(e, "hipMalloc(mask_tab)");
        if ((e = hipMemcpy(tree->order_pos, pos.data(), pos.size() * sizeof(int32_t), hipMemcpyHostToDevice)) != hipSuccess)
            return fail(e, "hipMemcpy(order_pos)");
        if ((e = hipMemcpy(tree->level_dev, lev.data(), lev.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy(level_dev)");
        if ((e = hipMemcpy(tree->mask_tab, mtab.data(), mtab.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy(mask_tab)");
    }
    tree->bytes = (size_t)S * (sizeof(int32_t) + 3) + node.size() * sizeof(float) + trans.size() * sizeof(Trans) + order.size() * sizeof(int32_t) + (size_t)S * sizeof(int32_t);
    *out = tree;
    return 0;
}

extern "C" void rnad_tree_destroy(rnad_tree_t *tree) {
    if (!tree) return;
    if (tree->node) (void)hipFree(tree->node);
    if (tree->trans) (void)hipFree(tree->trans);
    if (tree->level_order) (void)hipFree(tree->level_order);
    if (tree->hot_slot) (void)hipFree(tree->hot_slot);
    if (tree->order_pos) (void)hipFree(tree->order_pos);
    if (tree->level_dev) (void)hipFree(tree->level_dev);
    if (tree->mask_tab) (void)hipFree(tree->mask_tab);
    for (auto &kv : tree->cuts) {
        BucketCut &c = kv.second;
        if (c.bucket_of) (void)hipFree(c.bucket_of);
        if (c.bucket_lo) (void)hipFree(c.bucket_lo);
        if (c.bucket_path) (void)hipFree(c.bucket_path);
        if (c.path_states) (void)hipFree(c.path_states);
        if (c.bucket_span) (void)hipFree(c.bucket_span);
        if (c.group_by_lo) (void)hipFree(c.group_by_lo);
        if (c.upper_list) (void)hipFree(c.upper_list);
        if (c.upper_walk) (void)hipFree(c.upper_walk);
        if (c.anchor1) (void)hipFree(c.anchor1);
        if (c.hot_list) (void)hipFree(c.hot_list);
        if (c.hot_of) (void)hipFree(c.hot_of);
    }
    delete tree;
}

extern "C" int64_t rnad_tree_info(const rnad_tree_t *tree, int which) {
    if (!tree) return -1;
    switch (which) {
        case 0: return tree->S;
        case 1: return tree->C;
        case 2: return tree->A;
        case 3: return tree->max_depth;
        case 4: return tree->NS;
        case 5: return (int64_t)tree->bytes;
        case 6: return tree->uniform_length ? 1 : 0;
        default: return -1;
    }
}

// ---------------------------------------------------------------------------------------- K1 observe
//
// One lane (thread) per episode reads its state's node row -- NS floats at a 16-byte aligned address, as NS/4
// dwordx4 loads; the whole node table (48 B/state for A = 3) is L2-resident -- builds the mover's view in
// registers (transpose + negate for the column player, legal bits -> 1.0/0.0), and parks it in LDS.  The
// block then streams the 256 x 2A^2 tile to HBM with full-width, fully coalesced 16-byte stores: per-lane
// rows are 72 B (A = 3) / 200 B (A = 5), so storing them straight from registers would scatter 8-byte
// pieces at a 72-byte stride.  LDS rows use an odd stride so that neither the strided row writes nor the
// linear read-back conflict.
//
// Algorithmic bytes per env step: 4 (idx) + 8A^2 (ev + legal rows) + 2A^2*sizeof(obs) + 4A (mask).
template <typename OutT>
struct Pack16;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <>
struct Pack16<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void store(float *dst, const float *v) {
        const v4f x = {v[0], v[1], v[2], v[3]};
        // streaming output, never re-read by this kernel: non-temporal keeps the node table resident in L2 (measured
        // 18.8 -> 17.8 us per launch inside the rollout)
        __builtin_nontemporal_store(x, reinterpret_cast<v4f *>(dst));
    }
    static __device__ __forceinline__ float cvt(float x) { return x; }
};
template <>
struct Pack16<__half> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void store(__half *dst, const float *v) {
        union {
            __half h[8];
            uint4 u;
        } p;
#pragma unroll
        for (int i = 0; i < 8; ++i) p.h[i] = __float2half(v[i]);
        const v4u x = {p.u.x, p.u.y, p.u.z, p.u.w};
        __builtin_nontemporal_store(x, reinterpret_cast<v4u *>(dst));
    }
    static __device__ __forceinline__ __half cvt(float x) { return __float2half(x); }
};

constexpr int kObsLanes = 256;

template <int A, int P, typename OutT, bool VEC>
__global__ __launch_bounds__(kObsLanes) void k_observe(const float *__restrict__ node, const int32_t *__restrict__ idx,
                                                       int64_t B, OutT *__restrict__ obs, uint8_t *__restrict__ mbits,
                                                       float *__restrict__ maskf) {
    constexpr int AA = A * A, ROW = 2 * AA, NS = (AA + 2 + 3) & ~3, RS = ROW | 1;
    __shared__ float tile[kObsLanes * RS];
    const int64_t b0 = (int64_t)blockIdx.x * kObsLanes;
    const int lane = threadIdx.x;
    const int64_t b = b0 + lane;
    if (b < B) {
        const int64_t s = idx ? (int64_t)idx[b] : b;
        const float4 *row4 = reinterpret_cast<const float4 *>(node + s * NS);
        float r[NS];
#pragma unroll
        for (int q = 0; q < NS / 4; ++q) {
            const float4 v = row4[q];
            r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
        }
        const uint64_t bits = (uint64_t)__float_as_uint(r[AA]) | ((uint64_t)__float_as_uint(r[AA + 1]) << 32);
        float *o = tile + lane * RS;
#pragma unroll
        for (int i = 0; i < A; ++i)
#pragma unroll
            for (int j = 0; j < A; ++j) {
                const int src = P ? j * A + i : i * A + j;             // swapaxes(2,3) for the column player
                o[i * A + j] = P ? -r[src] : r[src];                   // -ev keeps the sign bit of -0.0
                o[AA + i * A + j] = ((bits >> src) & 1) ? 1.0f : 0.0f;
            }
        uint32_t mb = 0;
#pragma unroll
        for (int i = 0; i < A; ++i) {
            const int src = P ? i : i * A;                             // observations[:, 1, i, 0]
            const uint32_t bit = (uint32_t)((bits >> src) & 1);
            mb |= bit << i;
            if (maskf) maskf[b * A + i] = bit ? 1.0f : 0.0f;
        }
        if (mbits) mbits[b] = (uint8_t)mb;
    }
    __syncthreads();
    const int64_t left = B - b0;
    const int total = (int)(left < kObsLanes ? left : kObsLanes) * ROW;
    OutT *out = obs + b0 * ROW;
    if (VEC) {
        constexpr int N = Pack16<OutT>::N;
        for (int k = lane * N; k + N <= total; k += kObsLanes * N) {
            float v[N];
#pragma unroll
            for (int u = 0; u < N; ++u) {
                const int e = k + u;
                v[u] = tile[(e / ROW) * RS + (e % ROW)];
            }
            Pack16<OutT>::store(out + k, v);
        }
        const int tail0 = total - total % N;
        if (lane < total - tail0) out[tail0 + lane] = Pack16<OutT>::cvt(tile[((tail0 + lane) / ROW) * RS + (tail0 + lane) % ROW]);
    } else {
        for (int e = lane; e < total; e += kObsLanes) out[e] = Pack16<OutT>::cvt(tile[(e / ROW) * RS + (e % ROW)]);
    }
}

template <int A, typename OutT>
static int launch_observe_t(const rnad_tree_t *tree, int64_t B, const int32_t *idx, int player, OutT *obs, uint8_t *mbits,
                            float *maskf, hipStream_t stream) {
    const unsigned grid = (unsigned)((B + kObsLanes - 1) / kObsLanes);
    const bool vec = ((uintptr_t)obs % 16) == 0;
    ProfScope prof(PROF_OBSERVE, stream);
#define RNAD_OBS_LAUNCH(P_, V_) \
    hipLaunchKernelGGL((k_observe<A, P_, OutT, V_>), dim3(grid), dim3(kObsLanes), 0, stream, tree->node, idx, B, obs, mbits, maskf)
    if (player == 0) {
        if (vec) RNAD_OBS_LAUNCH(0, true); else RNAD_OBS_LAUNCH(0, false);
    } else {
        if (vec) RNAD_OBS_LAUNCH(1, true); else RNAD_OBS_LAUNCH(1, false);
    }
#undef RNAD_OBS_LAUNCH
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

namespace rnad {
int launch_observe(const rnad_tree_t *tree, int64_t B, const int32_t *idx, int player, void *obs, int obs_half, uint8_t *mbits,
                   float *maskf, hipStream_t stream) {
    if (B == 0) return 0;
    RNAD_DISPATCH_A(tree->A, {
        if (obs_half) return launch_observe_t<kA, __half>(tree, B, idx, player, (__half *)obs, mbits, maskf, stream);
        return launch_observe_t<kA, float>(tree, B, idx, player, (float *)obs, mbits, maskf, stream);
    });
    return 0;
}
}  // namespace rnad

extern "C" int rnad_observe(const rnad_tree_t *tree, int64_t B, const int32_t *idx, int player, void *obs, int obs_half,
                            uint8_t *mask_bits, float *mask, void *stream) {
    RNAD_REQUIRE(tree && idx && obs, "rnad_observe: null argument");
    RNAD_REQUIRE(B >= 0, "rnad_observe: negative batch");
    RNAD_REQUIRE(player == 0 || player == 1, "rnad_observe: player must be 0 or 1, got %d", player);
    return launch_observe(tree, B, idx, player, obs, obs_half, mask_bits, mask, (hipStream_t)stream);
}

extern "C" int rnad_observe_all(const rnad_tree_t *tree, float *obs_row, float *obs_col, void *stream) {
    RNAD_REQUIRE(tree && obs_row && obs_col, "rnad_observe_all: null argument");
    int rc = launch_observe(tree, tree->S, nullptr, 0, obs_row, 0, nullptr, nullptr, (hipStream_t)stream);
    if (rc) return rc;
    return launch_observe(tree, tree->S, nullptr, 1, obs_col, 0, nullptr, nullptr, (hipStream_t)stream);
}
