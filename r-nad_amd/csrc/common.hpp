// common.hpp -- shared host/device plumbing of librnad_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include "rnad_hip.h"
#include "rnad_rng.h"

namespace rnad {

void set_error(const char *fmt, ...);

#define RNAD_HIP_OK(expr)                                                                            \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            rnad::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

#define RNAD_REQUIRE(cond, ...)           \
    do {                                  \
        if (!(cond)) {                    \
            rnad::set_error(__VA_ARGS__); \
            return 2;                     \
        }                                 \
    } while (0)

// One (int32 next, f32 chance, f32 value) per chance outcome; C of them are contiguous per joint action.
struct Trans {
    int32_t next;
    float chance;
    float value;
};

// node row: expected_value[A*A] | legal bits lo | legal bits hi | pad to a multiple of 4 floats (16 B)
inline int node_stride_floats(int A) { return (A * A + 2 + 3) & ~3; }

// Makes `device` current for the scope and restores the caller's device afterwards (entry points that own a device id).
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) ok = hipSetDevice(device) == hipSuccess;
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// Zero `bytes` (a multiple of 4) at a 4-byte aligned device pointer with a KERNEL on `stream`.  Not hipMemsetAsync: the memset node of a
// captured hipGraph was seen to write garbage after ~57 replays (ROCm 7.2; tests/test_hip_graph.py::test_many_replays_stay_finite),
// so nothing on a path that may be captured clears memory that way.
int zero_async(void *ptr, size_t bytes, hipStream_t stream);

// Event bracketing for bench.py's roofline leg.
enum ProfKernel {
    PROF_OBSERVE = 0, PROF_ACT = 1, PROF_LEARN = 2, PROF_MLP = 3, PROF_MLP_BWD = 4,
    PROF_BUCKET_KEYS = 5, PROF_BUCKET_SORT = 6, PROF_BUCKET_ROLLOUT = 7, PROF_BUCKET_LEARN = 8, PROF_BUCKET_FINISH = 9, PROF_COUNT = 10
};
struct ProfScope {
    int which;
    hipStream_t stream;
    hipEvent_t start = nullptr;
    ProfScope(int which, hipStream_t stream);
    ~ProfScope();
};

}  // namespace rnad

// One partition of the tree into bucket subtrees (bucket.hip, "cut by subtree size"): states whose subtree holds more than `rows`
// states are UPPER; the non-upper children of an upper state, packed into runs of consecutive siblings spanning at most `rows`
// state ids, are the GROUPS.  Bucket ids: groups 0 .. n_groups - 1, then one terminal bucket per upper state (lanes that leave
// the tree from it).  Built on the host on first use for a given `rows` and kept with the tree handle.
struct BucketCut {
    int rows = 0, n_groups = 0, n_upper = 0, n_buckets = 0, max_path = 0;
    int32_t *bucket_of = nullptr;    // device [S]: bucket of a state (group for every state of a group's id range, n_groups + slot for upper states, -1 else)
    int32_t *bucket_lo = nullptr;    // device [n_buckets]: first state id of the group / the upper state itself
    int32_t *bucket_path = nullptr;  // device [n_buckets]: env steps a lane of the bucket spends above the group (terminal buckets: all of them)
    int32_t *group_by_lo = nullptr;  // device [n_groups]: the groups in ascending order of their first state id
    int32_t *bucket_span = nullptr;  // device [n_buckets]: state ids a group spans (bucket_lo .. bucket_lo + span - 1); 0 for terminal buckets
    int32_t *path_states = nullptr;  // device [n_buckets][max(max_path, 1)]: the state every lane of the bucket sits in at env step t, for the steps
                                     // it shares with the whole bucket (above the group; and at the root of a group that is one subtree)
    int32_t *upper_list = nullptr;   // device [max(n_upper, 1)]: the upper states in slot order
    void *upper_walk = nullptr;      // device [n_upper][A][A][C] {next state, its bucket, chance}: the transition table of the upper states,
                                     // compact, for k_bucket_keys to stage in LDS
    // hybrid keys walk (trees whose upper tables exceed the LDS, configs[3]): the upper states of the top levels that fit are staged
    int n_hot = 0;                   // upper slots staged in LDS by k_bucket_keys_hybrid (the first n_hot entries of hot_list)
    int32_t *hot_list = nullptr;     // device [max(n_hot, 1)]: upper slots (indices into upper_list), top levels first
    int32_t *hot_of = nullptr;       // device [max(n_upper, 1)]: position in hot_list, or -1
    int32_t *anchor1 = nullptr;      // device [S]: for a state strictly below the root of a group subtree, its ancestor one level below that
                                     // root (itself for the root's children); 0 elsewhere.  A staged actor's second level (rnad_bucket_stage_*)
    std::vector<int32_t> host_bucket_of;  // the same map on the host (rnad_bucket_map: tests and tools)
};

struct rnad_tree {
    int64_t S = 0;
    int C = 0, A = 0, NS = 0, device = 0, max_depth = 0;
    bool uniform_length = false;  // every episode takes exactly max_depth transitions (no early terminal, nothing pruned)
    float *node = nullptr;        // [S][NS]
    rnad::Trans *trans = nullptr; // [S][A][A][C]
    // NashConv support: states grouped by depth below the root (level 0 = {1}); ids of one level are contiguous
    // in level_order, parent/child relation is in `trans`.
    int n_levels = 0;
    int32_t *level_order = nullptr;      // device [S]
    std::vector<int64_t> level_offsets;  // host [n_levels + 1]
    std::vector<int32_t> level_of;       // host [S] (-1: unreachable)
    // tabular learner: the states of the top levels (the first n_hot entries of level_order) get a slot in a per-block LDS table
    int n_hot = 0;
    int32_t *hot_slot = nullptr;         // device [S]: position in level_order if < n_hot, else -1
    // bucketed tabular pipeline (bucket.hip)
    int32_t *order_pos = nullptr;        // device [S]: position of the state in level_order (-1: unreachable / state 0)
    uint8_t *level_dev = nullptr;        // device [S]: depth below the root (255: unreachable / state 0)
    uint8_t *mask_tab = nullptr;         // device [2][S]: the mover's legal-action bits at (player to move, state) (episode.py:208)
    std::vector<int64_t> level_max_subtree;  // host [n_levels]: largest subtree (states, the root of it included) below a state of that level
    bool contiguous_subtrees = false;    // ids are DFS pre-order: the subtree of s is exactly [s, s + size(s))  (tree.py:311-330)
    std::vector<int32_t> subtree_size;   // host [S]: states in the subtree of s, s included (0: unreachable / state 0)
    std::vector<int64_t> child_offsets;  // host CSR of the live children (index != 0, chance > 0) of every state, ascending ids
    std::vector<int32_t> children;
    mutable std::map<int, BucketCut> cuts;  // by rows; handles are used from one host thread (include/rnad_hip.h)
    mutable std::map<std::pair<int64_t, int>, int> plan_rows;  // (lanes, forced rows or 0) -> rows of the cut the planner chose (0: none fits)
    mutable std::set<std::tuple<int64_t, int, int, int>> plan_sized;  // (lanes, forced rows, sort tile, chunk) rnad_bucket_plan handed buffer sizes out for
    size_t bytes = 0;
};

// Dispatch a functor templated on A (1..RNAD_MAX_ACTIONS).
#define RNAD_DISPATCH_A(A_, ...)                                            \
    switch (A_) {                                                           \
        case 1: { constexpr int kA = 1; __VA_ARGS__; } break;               \
        case 2: { constexpr int kA = 2; __VA_ARGS__; } break;               \
        case 3: { constexpr int kA = 3; __VA_ARGS__; } break;               \
        case 4: { constexpr int kA = 4; __VA_ARGS__; } break;               \
        case 5: { constexpr int kA = 5; __VA_ARGS__; } break;               \
        case 6: { constexpr int kA = 6; __VA_ARGS__; } break;               \
        case 7: { constexpr int kA = 7; __VA_ARGS__; } break;               \
        case 8: { constexpr int kA = 8; __VA_ARGS__; } break;               \
        default: rnad::set_error("max_actions %d out of range [1,%d]", A_, RNAD_MAX_ACTIONS); return 2; \
    }
