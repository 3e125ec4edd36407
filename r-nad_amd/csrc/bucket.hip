// bucket.hip -- the bucketed tabular pipeline: rollout in subtree order + per-(player, state) gradient sums in LDS (gfx950).
//
// Replaces, on trees that are small next to the batch (2S distinct net inputs for T*B slots): environment/episode.py:175-230
// (Episodes.generate) and the tensor program of learn/rnad.py:365-425 including the reduction that loss.backward() performs over
// the slots that share a net input.  Citations are baskuit/R-NaD file:line.
//
// Why buckets.  The weight gradient is linear in dL/dout, and a net's input depends on (player to move, state) only
// (episode.py:62-68), so the update needs, per row (P, s), the SUM of the per-slot gradients of the slots that sit in s at a step
// of parity P.  Summing 12.6 M slots into 132 862 rows with global atomics costs ~1 ms (16.8 M spread 64-bit atomics, round 1).
// Here the tree is cut by subtree size (ids are DFS pre-order, tree.py:311-330, so the subtree of s is the id range
// [s, s + size(s))): states whose subtree exceeds R rows are "upper"; the other children of an upper state, packed into runs of
// consecutive siblings spanning at most R ids, are the "groups".  Lanes are sorted by the group they descend into (their
// bucket): one workgroup then owns ALL slots of the rows of its group and adds them up in an LDS table indexed by
// (state - first id of the group); the rows above are upper states every lane of the bucket went through, one row per step
// for the whole workgroup, and are added into 16 LDS copies of that row.  No global atomic on the common path, results in 64-bit fixed
// point (integer sums: any order, same bits).  On a regular tree the cut is a level; on a pruned one it follows the subtrees.
//
//   k_bucket_keys      lane-ordered: plays the env steps above the cut only, key = the group reached (or the upper state the lane ends in)
//   k_bucket_hist/scan/items/scatter   stable counting sort of the lanes by key (deterministic), work items per bucket
//   k_bucket_rollout   bucket-ordered: thread j replays lane lane_ids[j] from the root (counter-based draws keyed by the GLOBAL lane
//                      id, include/rnad_rng.h) and records the trajectory -- column j of every [T, B] buffer, or (COMPACT) of
//                      `indices` alone plus 12 bytes per lane: packed actions and the episode's one reward
//   k_bucket_expand    the dense [T, B] buffers of a compact trajectory, when something asks for them
//   k_bucket_learn     one workgroup per work item: backward-in-time V-trace / NeuRD pass per lane (learn_math.hpp; COMPACT:
//                      fast_slot on row-precomputed operands), sums in LDS
//   k_policy_rows / k_row_records   everything that depends on the (player, state) row alone, once per row instead of per slot
//   k_bucket_finish    fixed point -> fp32 tables dL/dlogit [2S, A], dL/dv [2S], normalised by the batch-global N_P
#include "learn_math.hpp"
#include "row_records.hpp"
#include "rollout_math.hpp"

#include <algorithm>
#include <cstdlib>

using namespace rnad;
using namespace rnad::dev;

#ifndef RNAD_ABLATE
#define RNAD_ABLATE 0  // bit mask of k_bucket_learn ablations (tools/ablate_learn.sh: what the kernel's time is made of; results are wrong)
#endif

namespace {

constexpr int kThreads = 256;
constexpr int kMaxPath = 32;         // env steps above a group (path rows of the LDS table)
#ifndef RNAD_PATH_SLOTS
#define RNAD_PATH_SLOTS 16
#endif
constexpr int kPathSlots = RNAD_PATH_SLOTS;  // copies of a path row in LDS: lane l adds into copy l & 15 (4-way instead of 64-way same-address adds)
constexpr int kMaxBuckets = 12288;   // LDS histogram of the sort passes: 48 KiB of int32
constexpr int kMaxUpper = 8192;      // upper states (64 replicas of their rows are kept)
constexpr int kSortThreads = 1024;   // threads per block of the sort passes
constexpr int kSortLanes = 4096;     // lanes per block of the sort passes (4 per thread, 256 contiguous per wave)
constexpr int kChunkDefault = 256;   // lanes per work item of the learner: one pass of a 256-thread workgroup (no long tail items)
constexpr int kReplicas = 64;        // copies of the upper-row table the workgroups spread their path sums over
constexpr int kMaxSteps = 64;        // T_cap bound of the alive counters in LDS
constexpr int kLaneBits = 22;        // B <= 2^22 lanes per call: a row receives at most one addend per lane
constexpr int kLearnLds = 64 * 1024; // LDS budget of one learner workgroup
constexpr int kKeysLds = 144 * 1024;  // LDS budget of k_bucket_keys_lds (the upper states' tables + the tile's histogram): one 1024-thread workgroup per CU still fits
constexpr int kPackedSteps = 10;      // env steps whose decisions k_bucket_keys hands to k_bucket_rollout (6 bits each)
constexpr int kCompactSteps = 21;     // env steps of a compact trajectory: 3 bits of action per step in one 64-bit word
constexpr int kSharedRoot = 256;      // flag in bucket_path: the group is one subtree (its root row is shared by the bucket's lanes)
constexpr int kFinishRows = 4;        // rows per thread of k_bucket_finish
constexpr int kTicketGroups = 64;     // first-level tickets of k_bucket_finish's last-workgroup election
constexpr int kTargetLanes = 1024;   // finest cut whose groups still hold this many lanes on average (configs[1]: full work items win)
constexpr int kMinLanesLevelCut = 128;  // a level cut (one group per subtree) is taken while its groups hold this many lanes on average

// lanes per thread of the walks with tables in LDS (k_bucket_keys_lds, k_bucket_keys_hybrid).  r04: one lane per thread -- 18.9 -> 17.7 us on
// configs[1], 42.9 -> 37.8 on configs[3]'s hybrid walk (two: what the global-table walk above keeps; four: 18.3 / 46.1)
#ifndef RNAD_PLAY_LDS
#define RNAD_PLAY_LDS 1
#endif
constexpr int kPlayLds = RNAD_PLAY_LDS;
constexpr int kMinSortTile = kSortThreads * kPlayLds > 1024 ? kSortThreads * kPlayLds : 1024;  // a tile is a whole number of passes of the LDS walks
// `kTile` = the plan's sort tile as a constant (tiles below kMinSortTile are never planned)
#define RNAD_DISPATCH_TILE(p_, ...)                                                          \
    switch ((p_).tile) {                                                                     \
        case 1024: { constexpr int kTile = kMinSortTile > 1024 ? kMinSortTile : 1024; __VA_ARGS__; } break; \
        case 2048: { constexpr int kTile = kMinSortTile > 2048 ? kMinSortTile : 2048; __VA_ARGS__; } break; \
        case 8192: { constexpr int kTile = 8192; __VA_ARGS__; } break;                        \
        default: { constexpr int kTile = kSortLanes; __VA_ARGS__; } break;                    \
    }

inline unsigned blocks_for(int64_t n, int per = kThreads) { return (unsigned)((n + per - 1) / per); }

struct Plan {
    const BucketCut *cut = nullptr;
    int lds = 0, path_words = 0, sort_blocks = 0, chunk = kChunkDefault;
    int tile = kSortLanes;  // lanes per workgroup of the sort passes (keys walk, histogram, scatter): kSortLanes, or a fraction of it on small batches
    int rel_bytes = 1;  // width of a relative state of the compact trajectory: states of a group are bucket_lo + (0 .. rows - 1)
    int forced = 0;     // RNAD_BUCKET_ROWS at the time of the call (0: the planner's choice)
    int64_t max_items = 0;
};

// One outcome of one joint action of an upper state, as k_bucket_keys walks it: where it leads, which bucket that state belongs to
// (group, upper slot + n_groups, or -1 for state 0), and the outcome's probability for the chance draw.
struct UpperWalk {
    int32_t next, key;
    float chance;
};

__global__ __launch_bounds__(kThreads) void k_upper_walk(int n, int AAC, const int32_t *__restrict__ upper_list, const Trans *__restrict__ trans,
                                                         const int32_t *__restrict__ bucket_of, UpperWalk *__restrict__ out) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n * AAC) return;
    const Trans t = trans[(int64_t)upper_list[i / AAC] * AAC + i % AAC];
    out[i] = UpperWalk{t.next, bucket_of[t.next], t.chance};
}

// The cut of `tree` for tables of `rows` rows, on the host (O(S)).
struct HostCut {
    int rows = 0, n_groups = 0, n_upper = 0, n_buckets = 0, max_path = 0;
    std::vector<int32_t> bucket_of, lo, span, path, upper_list, path_states;  // path_states: [n_buckets][max(max_path, 1)]
    std::vector<int32_t> group_by_lo;  // the groups sorted by lo (group ids follow the upper states they hang below, not the state ids)
    std::vector<int32_t> anchor1;      // [S] BucketCut::anchor1
};

HostCut build_cut(const rnad_tree_t *tree, int rows) {
    HostCut h;
    h.rows = rows;
    const int64_t S = tree->S;
    std::vector<int32_t> &bucket_of = h.bucket_of, &lo = h.lo, &span = h.span, &path = h.path, &upper_list = h.upper_list;
    bucket_of.assign((size_t)S, -1);
    h.anchor1.assign((size_t)S, 0);
    // the subtree of a group's root c, one level down: every state below a child d of c is anchored at d
    auto anchor_below = [&](int64_t c) {
        for (int64_t i = tree->child_offsets[(size_t)c]; i < tree->child_offsets[(size_t)c + 1]; ++i) {
            const int64_t d = tree->children[(size_t)i], d_hi = d + tree->subtree_size[(size_t)d];
            for (int64_t x = d; x < d_hi; ++x) h.anchor1[(size_t)x] = (int32_t)d;
        }
    };
    std::vector<int32_t> parent_upper((size_t)S, 0);  // upper state -> the upper state above it (0: the root)
    std::vector<int32_t> group_parent;                // bucket -> the upper state it hangs below (0: none)
    auto is_upper = [&](int64_t s) { return tree->subtree_size[(size_t)s] > rows; };
    if (!is_upper(1)) {  // the whole tree fits one table: a single group, nothing above it
        lo.push_back(1);
        span.push_back((int32_t)tree->subtree_size[1]);
        path.push_back(kSharedRoot);  // (every lane starts in state 1)
        group_parent.push_back(0);
        anchor_below(1);
        for (int64_t s = 1; s < S; ++s)
            if (tree->level_of[(size_t)s] >= 0) bucket_of[(size_t)s] = 0;
    } else {
        for (int64_t u = 1; u < S; ++u) {
            if (tree->level_of[(size_t)u] < 0 || !is_upper(u)) continue;
            upper_list.push_back((int32_t)u);
            const int level = tree->level_of[(size_t)u];
            int64_t g_lo = -1, g_hi = -1;
            int g_subtrees = 0;
            auto close = [&]() {
                if (g_lo < 0) return;
                const int32_t gid = (int32_t)lo.size();
                lo.push_back((int32_t)g_lo);
                span.push_back((int32_t)(g_hi - g_lo));
                // a group of ONE subtree: every lane of the bucket also shares the group's root for two more steps (kSharedRoot)
                path.push_back(2 * (level + 1) | (g_subtrees == 1 ? kSharedRoot : 0));
                group_parent.push_back((int32_t)u);
                for (int64_t x = g_lo; x < g_hi; ++x) bucket_of[(size_t)x] = gid;
                g_lo = g_hi = -1;
            };
            for (int64_t i = tree->child_offsets[(size_t)u]; i < tree->child_offsets[(size_t)u + 1]; ++i) {
                const int64_t c = tree->children[(size_t)i];
                if (is_upper(c)) {
                    parent_upper[(size_t)c] = (int32_t)u;
                    close();
                    continue;
                }
                const int64_t c_hi = c + tree->subtree_size[(size_t)c];
                anchor_below(c);
                if (g_lo >= 0 && c == g_hi && c_hi - g_lo <= rows) {
                    g_hi = c_hi;
                    ++g_subtrees;
                } else {
                    close();
                    g_lo = c;
                    g_hi = c_hi;
                    g_subtrees = 1;
                }
            }
            close();
        }
    }
    h.n_groups = (int)lo.size();
    h.n_upper = (int)upper_list.size();
    h.n_buckets = h.n_groups + h.n_upper;
    for (int i = 0; i < h.n_upper; ++i) {  // terminal buckets: lanes that leave the tree from an upper state
        const int32_t u = upper_list[(size_t)i];
        bucket_of[(size_t)u] = h.n_groups + i;
        lo.push_back(u);
        span.push_back(0);
        path.push_back(2 * tree->level_of[(size_t)u] + 2);
        group_parent.push_back(u);
    }
    int max_path = 0;
    for (int32_t v : path) max_path = std::max(max_path, (int)(v & (kSharedRoot - 1)) + ((v & kSharedRoot) ? 2 : 0));
    h.max_path = max_path;
    // the states a bucket's lanes share: the upper states from the root down to the one the bucket hangs below (a state holds both
    // players' steps of its level), then -- a group that is one subtree -- its root
    const int stride = std::max(max_path, 1);
    h.path_states.assign((size_t)h.n_buckets * stride, 0);
    if (max_path <= kMaxPath) {  // (deeper cuts are rejected by cut_fits)
        for (int b = 0; b < h.n_buckets; ++b) {
            int32_t *row = h.path_states.data() + (size_t)b * stride;
            for (int32_t u = group_parent[(size_t)b]; u != 0; u = parent_upper[(size_t)u]) {
                const int l = tree->level_of[(size_t)u];
                row[2 * l] = row[2 * l + 1] = u;
            }
            if (path[(size_t)b] & kSharedRoot) {
                const int n_path = path[(size_t)b] & (kSharedRoot - 1);
                row[n_path] = row[n_path + 1] = lo[(size_t)b];
            }
        }
    }
    if (upper_list.empty()) upper_list.push_back(0);
    h.group_by_lo.resize((size_t)h.n_groups);
    for (int g = 0; g < h.n_groups; ++g) h.group_by_lo[(size_t)g] = g;
    std::sort(h.group_by_lo.begin(), h.group_by_lo.end(), [&](int32_t a, int32_t b) { return lo[(size_t)a] < lo[(size_t)b]; });
    return h;
}

bool cut_fits(const HostCut &c) { return c.n_buckets <= kMaxBuckets && c.n_upper <= kMaxUpper && c.max_path <= kMaxPath; }

// The cut on the device, cached with the handle.  nullptr: HIP allocation failed.  (Called from make_plan for the CHOSEN cut only; the
// planner's candidates live on the host.  k_upper_walk runs on the null stream: rnad_bucket_plan is what a caller invokes before it
// captures a step.)
const BucketCut *get_cut(const rnad_tree_t *tree, int rows) {
    auto it = tree->cuts.find(rows);
    if (it != tree->cuts.end()) return &it->second;
    HostCut h = build_cut(tree, rows);
    BucketCut cut;
    cut.rows = rows;
    cut.n_groups = h.n_groups;
    cut.n_upper = h.n_upper;
    cut.n_buckets = h.n_buckets;
    cut.max_path = h.max_path;
    DeviceGuard guard(tree->device);
    auto up = [&](int32_t **dst, const std::vector<int32_t> &src) {
        if (hipMalloc((void **)dst, std::max<size_t>(src.size(), 1) * sizeof(int32_t)) != hipSuccess) return false;
        return src.empty() || hipMemcpy(*dst, src.data(), src.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess;
    };
    if (!guard.ok || !up(&cut.bucket_of, h.bucket_of) || !up(&cut.bucket_lo, h.lo) || !up(&cut.bucket_path, h.path) ||
        !up(&cut.upper_list, h.upper_list) || !up(&cut.path_states, h.path_states) || !up(&cut.bucket_span, h.span) ||
        !up(&cut.group_by_lo, h.group_by_lo) || !up(&cut.anchor1, h.anchor1)) {
        for (int32_t *ptr : {cut.bucket_of, cut.bucket_lo, cut.bucket_path, cut.upper_list, cut.path_states, cut.bucket_span, cut.group_by_lo, cut.anchor1})
            if (ptr) (void)hipFree(ptr);
        return nullptr;
    }
    if (cut.n_upper > 0) {  // (built on the device: the transition table has no host copy)
        const int AAC = tree->A * tree->A * tree->C, n = cut.n_upper * AAC;
        if (hipMalloc(&cut.upper_walk, (size_t)n * sizeof(UpperWalk)) == hipSuccess) {
            hipLaunchKernelGGL(k_upper_walk, dim3(blocks_for(n)), dim3(kThreads), 0, (hipStream_t)0, cut.n_upper, AAC, (const int32_t *)cut.upper_list,
                               (const Trans *)tree->trans, (const int32_t *)cut.bucket_of, (UpperWalk *)cut.upper_walk);
            if (hipStreamSynchronize((hipStream_t)0) != hipSuccess) {
                (void)hipFree(cut.upper_walk);
                cut.upper_walk = nullptr;  // (k_bucket_keys then walks the global tables)
            }
        }
    }
    if (cut.upper_walk && cut.n_upper > 0) {
        // hybrid keys walk: whole levels of upper states, top down, while their tables fit the LDS budget (next to the slot map)
        const int AAC = tree->A * tree->A * tree->C;
        const size_t per_state = (size_t)AAC * sizeof(UpperWalk) + 2 * (size_t)((tree->A + 3) & ~3) * sizeof(float);
        size_t budget = kKeysLds;
        if (const char *e = getenv("RNAD_KEYS_STAGE_BYTES")) budget = (size_t)std::max(0, atoi(e));  // (tests: partial staging on small trees)
        const size_t fixed = (size_t)cut.n_upper * sizeof(int32_t) + 16;
        std::vector<std::vector<int32_t>> by_level;
        for (int i = 0; i < cut.n_upper; ++i) {
            const int l = tree->level_of[(size_t)h.upper_list[(size_t)i]];
            if (l < 0) continue;
            if ((int)by_level.size() <= l) by_level.resize((size_t)l + 1);
            by_level[(size_t)l].push_back(i);
        }
        std::vector<int32_t> hot_list, hot_of((size_t)cut.n_upper, -1);
        for (const auto &lv : by_level) {
            if (fixed + (hot_list.size() + lv.size()) * per_state > budget) break;
            for (int32_t slot : lv) {
                hot_of[(size_t)slot] = (int32_t)hot_list.size();
                hot_list.push_back(slot);
            }
        }
        cut.n_hot = (int)hot_list.size();
        if (cut.n_hot > 0 && (!up(&cut.hot_list, hot_list) || !up(&cut.hot_of, hot_of))) cut.n_hot = 0;
    }
    cut.host_bucket_of = std::move(h.bucket_of);
    return &tree->cuts.emplace(rows, std::move(cut)).first->second;
}

// Table size: the finest cut (rows halved from the LDS budget down) whose groups still hold kTargetLanes lanes on average;
// if even the coarsest is finer than that, the coarsest.  false: this tree cannot be bucketed (ids not DFS pre-order, or no
// cut fits the limits above).  RNAD_BUCKET_ROWS forces a table size (tuning / tests).  Candidates are evaluated on the host alone
// (counts of a HostCut that is dropped again); the choice is kept per (lanes, forced rows), the chosen cut's device tables per rows.
int choose_rows(const rnad_tree_t *tree, int64_t B, int forced, int rows_max) {
    if (forced) return (forced >= 1 && forced <= rows_max && cut_fits(build_cut(tree, forced))) ? forced : 0;
    int chosen = 0;
    if (2 * tree->S > B) {
        // A tree that is large next to the batch (lazy rows, staged actor: learn/rnad.py): the finer the cut, the fewer states sit in
        // the groups the batch descends into -- the rows the policy head has to be evaluated on (configs[3]: 384 -> 307 rows per
        // table, 0.846 -> 0.81 ms per step).  The finest cut that fits the limits, in steps of 0.8.
        for (int rows = rows_max; rows >= 4; rows = rows * 4 / 5) {
            if (!cut_fits(build_cut(tree, rows))) {
                if (chosen) break;  // finer cuts only have more buckets
                continue;
            }
            chosen = rows;
        }
    } else {
        // r05, measured (tools/cut_sweep.sh, profiles/r05_cut_sweep.txt: the configs[1] tree at 2^18 .. 2^22 lanes, tables of 35 .. 563
        // rows): what k_bucket_play_learn pays for is not the number of steps below the cut but (1) groups that are runs of SEVERAL
        // sibling subtrees -- their lanes keep decision words of their own and replay the steps above the cut per lane -- and (2) two-byte
        // relative states; the cut "one level of the tree = one group per subtree" at one byte per state won at every batch size (2^19:
        // 142.7 us per step against 156.2 for the runs of three the halving below arrives at; 2^22: 420.9 against 507.0 for its 6 561
        // groups of ten states, whose keys pass walks a level more in lane order).  So: the COARSEST level cut -- a table the size of the
        // largest subtree of a level -- whose groups are all single subtrees with one-byte states, as long as a group still receives a
        // few wavefronts of lanes on average; trees without such a level (ragged subtrees packed into runs) fall through to the halving.
        for (int l = 0; l < tree->n_levels && !chosen; ++l) {
            const int64_t rows = tree->level_max_subtree[(size_t)l];
            if (rows < 1 || rows > rows_max || rows > 255) continue;
            const HostCut c = build_cut(tree, (int)rows);
            if (!cut_fits(c) || c.n_groups < 1 || B / c.n_groups < kMinLanesLevelCut) break;  // (deeper levels only have more groups)
            bool single = true;
            for (int g = 0; g < c.n_groups && single; ++g) single = (c.path[(size_t)g] & kSharedRoot) != 0;
            if (single) chosen = (int)rows;
        }
        if (chosen) return chosen;
        for (int rows = rows_max; rows >= 4; rows /= 2) {
            const HostCut c = build_cut(tree, rows);
            if (!cut_fits(c)) {
                if (chosen) break;  // finer cuts only have more buckets
                continue;
            }
            if (chosen && B / std::max(c.n_groups, 1) < kTargetLanes) break;
            chosen = rows;
        }
    }
    return chosen;
}

bool make_plan(const rnad_tree_t *tree, int64_t B, Plan &p) {
    if (!tree->contiguous_subtrees || B < 1 || B > ((int64_t)1 << kLaneBits)) return false;
    const int path_words = kMaxPath * kPathSlots * ((tree->A + 1) | 1);  // u64 words of the path region (slot stride odd: distinct banks)
    const int rows_max = (kLearnLds / 8 - path_words) / (2 * ((tree->A + 1) | 1));
    const char *force = getenv("RNAD_BUCKET_ROWS");
    const int forced = force ? std::max(atoi(force), -1) : 0;
    if (force && forced < 1) return false;
    const auto key = std::make_pair(B, forced);
    auto it = tree->plan_rows.find(key);
    if (it == tree->plan_rows.end()) it = tree->plan_rows.emplace(key, choose_rows(tree, B, forced, rows_max)).first;
    if (it->second == 0) return false;
    const BucketCut *chosen = get_cut(tree, it->second);
    if (!chosen) return false;
    p.cut = chosen;
    // the path region holds the rows of THIS cut's upper steps (not kMaxPath of them: LDS per workgroup bounds the resident waves)
    p.path_words = std::min(kMaxPath, std::max(chosen->max_path, 1)) * kPathSlots * ((tree->A + 1) | 1);
    p.lds = (p.path_words + 2 * chosen->rows * ((tree->A + 1) | 1)) * 8;
    p.rel_bytes = chosen->rows <= 255 ? 1 : 2;
    // Sort tile.  A workgroup of the keys pass / the scatter works through its tile in tile / 1024 passes, one after the other, and below
    // 2^20 lanes there are fewer tiles of 4096 than CUs: the launch then lasts as long as ONE workgroup's four passes whatever the batch
    // (r05, measured at 2^18 / 2^19 lanes: keys 15.0 / 15.5 us, scatter 9.4 / 10.6).  Smaller tiles while they still fill the chip at one
    // workgroup per CU -- not beyond: every workgroup stages the upper tables and takes the prefix of all bucket totals for itself
    // (2048-lane tiles at 2^20 lanes: keys 21.0 -> 23.3 us, scatter 13.1 -> 17.3, r04) -- and not on cuts with thousands of buckets
    // (configs[3]: the scatter's per-workgroup walk over the counters is what its time is).
    p.tile = kSortLanes;
    if (chosen->n_buckets <= 2048) {
        while (p.tile > kMinSortTile && (B + p.tile - 1) / p.tile < 256) p.tile /= 2;
        // ... and larger ones once every CU would get more than two of 4096: a workgroup's fixed part (the upper tables staged, the prefix
        // of all bucket totals: ~9 us of the scatter whatever the tile) is then paid half as often
        if ((B + p.tile - 1) / p.tile > 512) p.tile = 8192;
    }
    if (const char *t = getenv("RNAD_SORT_TILE")) {  // tuning knob / tests
        const int v = atoi(t);
        if ((v == 1024 || v == 2048 || v == 4096 || v == 8192) && v >= kMinSortTile) p.tile = v;
    }
    p.sort_blocks = (int)((B + p.tile - 1) / p.tile);
    if (const char *c = getenv("RNAD_BUCKET_CHUNK")) p.chunk = std::max(64, atoi(c));  // tuning knob
    p.max_items = (int64_t)chosen->n_buckets + B / p.chunk + 1 + 7;  // (+ 7: the XCD-aware item mapping needs 8 * ceil(n / 8) workgroups)
    p.forced = forced;
    return true;
}

// The sort tile and the chunk are read from the environment on every call (tuning knobs), and the caller's scratch buffer was sized by
// rnad_bucket_plan as hist[sort_blocks][n_buckets] + items: a knob that changed in between would make carve_scratch lay out more than was
// allocated (r05 advisor: a silent out-of-bounds write).  rnad_bucket_plan notes every (tile, chunk) it handed sizes out for; the entry
// points that write the scratch refuse a combination nobody asked the sizes of.
void note_sized(const rnad_tree_t *tree, int64_t B, const Plan &p) { tree->plan_sized.insert(std::make_tuple(B, p.forced, p.tile, p.chunk)); }
bool sized_for(const rnad_tree_t *tree, int64_t B, const Plan &p) {
    return tree->plan_sized.count(std::make_tuple(B, p.forced, p.tile, p.chunk)) != 0;
}
#define RNAD_REQUIRE_SIZED(tree_, B_, p_)                                                                                                  \
    RNAD_REQUIRE(sized_for(tree_, B_, p_), "RNAD_SORT_TILE / RNAD_BUCKET_CHUNK / RNAD_BUCKET_ROWS differ from what rnad_bucket_plan sized " \
                                            "this batch's scratch buffers for (tile %d, chunk %d): call rnad_bucket_plan again and reallocate", (p_).tile, (p_).chunk)

// Sum over the 64 lanes of a wave in integer arithmetic on the VALU's data-parallel primitives (no LDS traffic): an inclusive
// scan within each row of 16 lanes (row_shr 1, 2, 4, 8), then row 15 -> next row (row_bcast15) and lane 31 -> rows 2, 3
// (row_bcast31); lane 63 ends up with the total.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ long long dpp_moved(long long v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(unsigned long long)v, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)((unsigned long long)v >> 32), CTRL, ROW_MASK, 0xf, false);
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}

__device__ __forceinline__ long long wave_total_in_lane63(long long v) {
    v += dpp_moved<0x111, 0xf>(v);  // row_shr:1
    v += dpp_moved<0x112, 0xf>(v);  // row_shr:2
    v += dpp_moved<0x114, 0xf>(v);  // row_shr:4
    v += dpp_moved<0x118, 0xf>(v);  // row_shr:8
    v += dpp_moved<0x142, 0xa>(v);  // row_bcast15 into rows 1 and 3
    v += dpp_moved<0x143, 0xc>(v);  // row_bcast31 into rows 2 and 3
    return v;
}

// ---------------------------------------------------------------------------------------- 0. per-row tables
// Everything of the rollout and of the update that depends on the (player, state) row alone is evaluated once per row (2S rows)
// instead of once per slot (T * B): same functions on the same inputs, hence the same bits.
//
// k_policy_rows: policy[row] = policy head (net.py:45-46) of the actor's logits row under the mover's legal mask.
template <int A>
__global__ __launch_bounds__(kThreads) void k_policy_rows(int64_t rows, const float *__restrict__ logits, int64_t stride,
                                                          const uint8_t *__restrict__ mask_tab, float *__restrict__ policy,
                                                          const int32_t *__restrict__ row_list, const int64_t *__restrict__ n_rows,
                                                          const int32_t *__restrict__ upper_list, int n_upper, int64_t S) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    int64_t r = i;
    if (upper_list) {  // the rows of the upper states (both players) and of the absorbing state
        if (i >= 2 * ((int64_t)n_upper + 1)) return;
        const int64_t P = i / (n_upper + 1), k = i % (n_upper + 1);
        r = P * S + (k < n_upper ? upper_list[k] : 0);
    } else if (row_list) {  // a listed subset (count in device memory)
        if (i >= *n_rows) return;
        r = row_list[i];
    } else if (i >= rows) {
        return;
    }
    float in[A], pol[A];
#pragma unroll
    for (int a = 0; a < A; ++a) in[a] = logits[r * stride + a];
    policy_head_ptr<A>(in, mask_tab[r], pol, nullptr);
    constexpr int PS = (A + 3) & ~3;  // rows as the rollout kernels gather them: 16-byte loads
    float4 *p4 = reinterpret_cast<float4 *>(policy + r * PS);
#pragma unroll
    for (int u = 0; u < PS / 4; ++u)
        p4[u] = float4{4 * u < A ? pol[4 * u] : 0.0f, 4 * u + 1 < A ? pol[4 * u + 1] : 0.0f, 4 * u + 2 < A ? pol[4 * u + 2] : 0.0f,
                       4 * u + 3 < A ? pol[4 * u + 3] : 0.0f};
}

// flags[P * S + s] = 1 for both rows of every state that lies in a group some lane of the batch descends into (totals[group] > 0),
// 0 elsewhere -- also for the upper states, whose rows the caller evaluated before the keys pass.  What a staged actor still has to
// evaluate between the sort and the rollout (rnad_bucket_sort / rnad_bucket_play).
__global__ __launch_bounds__(kThreads) void k_group_flags(int64_t S, const int32_t *__restrict__ bucket_of, int n_groups,
                                                          const int32_t *__restrict__ totals, int32_t *__restrict__ flags) {
    const int64_t s = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (s >= S) return;
    const int b = bucket_of[s];
    const int32_t on = (b >= 0 && b < n_groups && totals[b] > 0) ? 1 : 0;
    flags[s] = on;
    flags[S + s] = on;
}

// (record layouts -- kRowStride / kRowLearn / kFastStride / kPolStride -- and the function that fills them: row_records.hpp)
template <int A>
__device__ __forceinline__ void load_policy_row(const float *__restrict__ tab, int64_t row, int64_t stride, bool vec4, float (&out)[A]) {
    if (vec4) {  // rows of a multiple of 4 floats, 16-byte aligned (k_row_records' policy rows)
        const float4 *p = reinterpret_cast<const float4 *>(tab + row * stride);
#pragma unroll
        for (int u = 0; u < (A + 3) / 4; ++u) {
            const float4 v = p[u];
            if (4 * u < A) out[4 * u] = v.x;
            if (4 * u + 1 < A) out[4 * u + 1] = v.y;
            if (4 * u + 2 < A) out[4 * u + 2] = v.z;
            if (4 * u + 3 < A) out[4 * u + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int a = 0; a < A; ++a) out[a] = tab[row * stride + a];
    }
}

template <int A>
__global__ __launch_bounds__(kThreads) void k_row_records(int64_t rows, const float *__restrict__ logit, const float *__restrict__ v,
                                                          const float *__restrict__ vt, const float *__restrict__ lr_,
                                                          const float *__restrict__ lr2_, const uint8_t *__restrict__ mask_tab,
                                                          rnad_learn_params_t hp, const rnad_step_params_t *__restrict__ sp,
                                                          float *__restrict__ rec, float *__restrict__ fast, float *__restrict__ pol_rows,
                                                          const int32_t *__restrict__ row_list, const int64_t *__restrict__ n_rows) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= (row_list ? *n_rows : rows)) return;
    const int64_t r = row_list ? (int64_t)row_list[i] : i;  // row list: only the rows a batch visited (lazy rows, learn/rnad.py)
    if (sp) {
        hp.alpha = sp->alpha;
        hp.one_minus_alpha = sp->one_minus_alpha;
    }
    const uint32_t bits = mask_tab[r];
    float lg[A], lr[A], lr2[A];
#pragma unroll
    for (int a = 0; a < A; ++a) {
        lg[a] = logit[r * A + a];
        lr[a] = lr_[r * A + a];
        lr2[a] = lr2_[r * A + a];
    }
    write_row_records<A>(r, lg, v[r], vt[r], lr, lr2, bits, hp, rec, fast, pol_rows);
}

// ---------------------------------------------------------------------------------------- 1. keys
// Lane b (lane order) plays env steps exactly as k_bucket_rollout will, for as long as it sits in an upper state: its key is the
// group it descends into, or the terminal bucket of the upper state it leaves the tree from.
// kPlay lanes per thread, in lock step: the walk is a chain of dependent gathers (policy rows -> transition record -> next state) with
// little arithmetic between them since the draws became one uniform each, so a thread keeps several independent chains in flight.
// A lane that has stopped keeps walking state 0 (its loads hit one line, its results are dropped), which keeps the code branch-free.
// Second staging level of a tabular actor (rnad_bucket_stage_*; trees that are large next to the batch).  The keys pass knows, for every
// lane that descends into a group, the subtree root it enters there: with `root` set it leaves that state per lane and stamps it in
// `mark0`; k_stage_rows<0> turns the stamped states into a row list, the caller evaluates its actor on it, k_stage_walk draws the
// transition at the root exactly as the rollout will (same counters, same rows: same outcome) and stamps the state it leads to in
// `mark1`, and k_stage_rows<1> lists the rows of the subtrees below the stamped states (BucketCut::anchor1).  A stamp is a function of the
// step's seed: nothing is cleared between steps, and a stale stamp that happens to match only lists rows nobody needed.
struct StageOut {
    unsigned long long *counts = nullptr;  // [2]: the lengths of the two row lists (cleared by the keys pass)
    uint32_t *root = nullptr;              // [B] lane order: the group subtree root a lane enters (0: it leaves the tree above the cut) in the
                                           // low kStageRootBits bits, the env steps it spent above the cut in the bits above
    uint32_t *sorted = nullptr;            // [B] the same words in bucket order (k_bucket_scatter), what k_stage_walk reads
    uint32_t *mark0 = nullptr, *mark1 = nullptr;  // [S] stamps
};
constexpr int kStageRootBits = 26;  // S <= 2^26 states; kMaxPath = 32 steps fit the 6 bits above
__host__ __device__ inline uint32_t stage_stamp(uint64_t seed) { return ((uint32_t)seed ^ (uint32_t)(seed >> 32)) | 1u; }
StageOut carve_stage(void *stage, int64_t B, int64_t S) {
    StageOut st;
    if (!stage) return st;
    st.counts = (unsigned long long *)stage;
    st.root = (uint32_t *)(st.counts + 2);
    st.sorted = st.root + B;
    st.mark0 = st.sorted + B;
    st.mark1 = st.mark0 + S;
    return st;
}

#ifndef RNAD_PLAY
#define RNAD_PLAY 2
#endif
constexpr int kPlay = RNAD_PLAY;

template <int A, int L>
__global__ __launch_bounds__(kThreads) void k_bucket_keys(const Trans *__restrict__ trans, int C, int64_t S, int64_t B, int n_steps,
                                                          const float *__restrict__ policy_tab, int64_t tab_stride, int vec4,
                                                          const int32_t *__restrict__ bucket_of, int n_groups, uint64_t seed,
                                                          const rnad_step_params_t *__restrict__ sp, int64_t lane0,
                                                          int32_t *__restrict__ keys, unsigned long long *__restrict__ decisions,
                                                          double *__restrict__ norm, StageOut stage) {
    const int64_t b0 = (int64_t)blockIdx.x * (kThreads * L) + threadIdx.x;  // this thread's lanes: b0 + l * kThreads
    if (b0 == 0 && norm) norm[0] = norm[1] = 0.0;  // summed up by k_bucket_alive at the end of this rollout
    if (b0 == 0 && stage.counts) stage.counts[0] = stage.counts[1] = 0ull;
    if (sp) seed = sp->seed;  // per-step scalars in device memory: a captured graph of the step replays with new values
    const int key_root = bucket_of[1];
    // decisions: what a lane drew at its first kPackedSteps env steps -- 3 bits of action and 3 bits of chance outcome per step,
    // the number of steps recorded in the top 4 bits -- so that k_bucket_rollout replays them without drawing again
    int state[L], key[L], steps[L];
    unsigned long long packed[L];
    bool on[L];  // still in an upper state
#pragma unroll
    for (int l = 0; l < L; ++l) {
        state[l] = 1;
        key[l] = key_root;
        steps[l] = 0;
        packed[l] = 0ull;
        on[l] = b0 + (int64_t)l * kThreads < B && key_root >= n_groups;
    }
    for (int t = 0; t < n_steps; t += 2) {  // one game transition per iteration: both players' steps in `state`, then the chance draw
        bool any = false;
#pragma unroll
        for (int l = 0; l < L; ++l) any |= on[l];
        if (!any) break;
        const bool two = t + 1 < n_steps;
        float u[L][3], pol0[L][A], pol1[L][A];
        int a0[L];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int64_t st = on[l] ? state[l] : 0;
            load_policy_row<A>(policy_tab, st, tab_stride, vec4 != 0, pol0[l]);
            if (two) load_policy_row<A>(policy_tab, S + st, tab_stride, vec4 != 0, pol1[l]);
        }
#pragma unroll
        for (int l = 0; l < L; ++l) rnad_decision_uniforms(seed, (uint64_t)(lane0 + b0 + (int64_t)l * kThreads), (uint32_t)t, u[l]);
#pragma unroll
        for (int l = 0; l < L; ++l) {
            a0[l] = pick<A>(pol0[l], u[l][0]);
            if (on[l]) {
                if (t < kPackedSteps) packed[l] |= (unsigned long long)a0[l] << (6 * t);
                steps[l] = t + 1;
            }
        }
        if (!two) break;
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int a1 = pick<A>(pol1[l], u[l][1]);
            int next, chosen = 0;
            float rew;
            transition_lane<A>(trans, C, on[l] ? state[l] : 0, a0[l], a1, nullptr, u[l][2], next, rew, &chosen);
            const int key_next = bucket_of[next];
            if (on[l]) {
                if (t + 1 < kPackedSteps) packed[l] |= (unsigned long long)(a1 | (chosen << 3)) << (6 * (t + 1));
                steps[l] = t + 2;
                state[l] = next;
                if (next != 0) key[l] = key_next;  // (a lane that leaves the tree from an upper state keeps that state's bucket)
                on[l] = next != 0 && key_next >= n_groups;
            }
        }
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const int64_t b = b0 + (int64_t)l * kThreads;
        if (b < B) {
            keys[b] = key[l];
            decisions[b] = packed[l] | ((unsigned long long)min(steps[l], kPackedSteps) << 60);
            if (stage.root) {  // (state: where the walk stopped -- the root of a group subtree when the key is a group)
                const int root = key[l] < n_groups ? state[l] : 0;
                stage.root[b] = (uint32_t)root | ((uint32_t)steps[l] << kStageRootBits);
                if (root) stage.mark0[root] = stage_stamp(seed);
            }
        }
    }
}

// The same walk with the upper states' tables in LDS.  Lanes arrive here in lane order, so at depth d a wave's lanes sit in up to
// min(64, (A^2 C)^d) different states and every gather of the global-table walk (two policy rows, the transition record, the bucket of
// the next state) touches that many cache lines: the walk is bound by the L1's line rate (counters: DESIGN.md section 5.1), not by
// arithmetic or HBM.  The upper states are few (configs[1]: 91), so each workgroup copies their policy rows and the compact
// transition table of the cut (BucketCut::upper_walk) into LDS once and walks there; a lane is at an upper SLOT instead of a state id.
// Same draws, same arithmetic, same keys and decisions as k_bucket_keys.
// One workgroup per sort tile (kSortLanes consecutive lanes, kSortThreads threads): it also counts its lanes per bucket in LDS and
// writes the tile's histogram row -- k_bucket_hist's work without its launch and without reading the keys back.
// Distinct observations (csrc/rows_dedup.hip): the copies of the representatives' records into the other rows of their groups, carried by
// the keys pass instead of a launch of their own (rnad_rows_expand: 7.4 us on configs[1]) -- the walk waits on LDS and arithmetic, the
// copies are 19 MB of independent loads and stores that fill those waits.  The walk itself reads the actor's policy rows of the UPPER
// states through rep_of (a representative's row is never written here: no race with the copies of other workgroups).
struct KeysExpand {
    const int32_t *rep_of = nullptr;  // [rows]: the representative of every row; NULL: nothing to copy
    int64_t rows = 0;
    int n = 0, max_quads = 0;
    float4 *tab[4] = {nullptr, nullptr, nullptr, nullptr};
    int quads[4] = {0, 0, 0, 0};
};
__device__ __forceinline__ void keys_expand_share(const KeysExpand &ex, int n_threads, int block, int n_blocks) {
    // workgroup b of n_blocks copies the rows [b * per, (b + 1) * per); a thread per (row, 16-byte chunk of the widest table)
    const int64_t per = (ex.rows + n_blocks - 1) / n_blocks, r0 = (int64_t)block * per;
    const int64_t r1 = r0 + per < ex.rows ? r0 + per : ex.rows;
    for (int64_t i = r0 * ex.max_quads + threadIdx.x; i < r1 * ex.max_quads; i += n_threads) {
        const int64_t r = i / ex.max_quads;
        const int q = (int)(i % ex.max_quads);
        const int64_t rp = ex.rep_of[r];
        if (rp == r) continue;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < ex.n && q < ex.quads[k]) ex.tab[k][r * ex.quads[k] + q] = ex.tab[k][rp * ex.quads[k] + q];
    }
}

inline size_t keys_lds_bytes(int n_upper, int n_buckets, int A, int C) {
    return (((size_t)n_upper * A * A * C * sizeof(UpperWalk) + 15) & ~(size_t)15) + (size_t)n_upper * 2 * ((A + 3) & ~3) * sizeof(float) +
           (size_t)n_buckets * sizeof(int32_t);
}

template <int A, int L, int TILE>
__global__ __launch_bounds__(kSortThreads) void k_bucket_keys_lds(const UpperWalk *__restrict__ walk, const int32_t *__restrict__ upper_list,
                                                                  int n_upper, int n_buckets, int C, int64_t S, int64_t B, int n_steps,
                                                                  const float *__restrict__ policy_tab, int64_t tab_stride, int key_root,
                                                                  int n_groups, uint64_t seed, const rnad_step_params_t *__restrict__ sp,
                                                                  int64_t lane0, int32_t *__restrict__ keys,
                                                                  unsigned long long *__restrict__ decisions, int32_t *__restrict__ hist,
                                                                  double *__restrict__ norm, StageOut stage, KeysExpand ex) {
    extern __shared__ __attribute__((aligned(16))) unsigned char keys_smem[];
    constexpr int PS = kPolStride<A>;
    const int AAC = A * A * C;
    UpperWalk *w = reinterpret_cast<UpperWalk *>(keys_smem);                                                             // [n_upper][A][A][C]
    float *pol = reinterpret_cast<float *>(keys_smem + (((size_t)n_upper * AAC * sizeof(UpperWalk) + 15) & ~(size_t)15));  // [n_upper][2][PS]
    int32_t *cnt = reinterpret_cast<int32_t *>(pol + (size_t)n_upper * 2 * PS);                                            // [n_buckets]
    for (int i = threadIdx.x; i < n_upper * AAC; i += kSortThreads) w[i] = walk[i];
    for (int i = threadIdx.x; i < n_upper * 2 * PS; i += kSortThreads) {
        const int slot = i / (2 * PS), player = (i / PS) & 1, a = i % PS;
        int64_t row = (int64_t)player * S + upper_list[slot];
        if (ex.rep_of) row = ex.rep_of[row];  // (the row's own copy may not have been written yet)
        pol[i] = a < A ? policy_tab[row * tab_stride + a] : 0.0f;
    }
    // (r04 made the copies here, "in flight during the walk": 19 MB through the B / 4096 workgroups of this launch -- 64 of them at 2^18
    // lanes, 27 us where the walk alone takes 9.  r05: extra workgroups of the scan launch that follows, k_bucket_scan)
    for (int i = threadIdx.x; i < n_buckets; i += kSortThreads) cnt[i] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0 && norm) norm[0] = norm[1] = 0.0;  // summed up by k_bucket_alive at the end of this rollout
    if (blockIdx.x == 0 && threadIdx.x == 0 && stage.counts) stage.counts[0] = stage.counts[1] = 0ull;
    if (sp) seed = sp->seed;
    __syncthreads();
    static_assert(TILE % (kSortThreads * L) == 0, "a sort tile is a whole number of passes");
    for (int pass = 0; pass < TILE / (kSortThreads * L); ++pass) {
        const int64_t b0 = (int64_t)blockIdx.x * TILE + (int64_t)pass * (kSortThreads * L) + threadIdx.x;  // lanes b0 + l * kSortThreads
        int slot[L], key[L], steps[L], root[L];
        unsigned long long packed[L];
        bool on[L];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            root[l] = key_root < n_groups ? 1 : 0;  // (a tree that is one group: every lane enters it at state 1)
            slot[l] = key_root - n_groups;
            key[l] = key_root;
            steps[l] = 0;
            packed[l] = 0ull;
            on[l] = b0 + (int64_t)l * kSortThreads < B && key_root >= n_groups;
        }
        for (int t = 0; t < n_steps; t += 2) {
            bool any = false;
#pragma unroll
            for (int l = 0; l < L; ++l) any |= on[l];
            if (!any) break;
            const bool two = t + 1 < n_steps;
            float u[L][3];
            int a0[L];
#pragma unroll
            for (int l = 0; l < L; ++l) rnad_decision_uniforms(seed, (uint64_t)(lane0 + b0 + (int64_t)l * kSortThreads), (uint32_t)t, u[l]);
#pragma unroll
            for (int l = 0; l < L; ++l) {
                float p0[A];
                load_policy_row<A>(pol, (int64_t)(on[l] ? slot[l] : 0) * 2, PS, true, p0);
                a0[l] = pick<A>(p0, u[l][0]);
                if (on[l]) {
                    if (t < kPackedSteps) packed[l] |= (unsigned long long)a0[l] << (6 * t);
                    steps[l] = t + 1;
                }
            }
            if (!two) break;
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int sl = on[l] ? slot[l] : 0;
                float p1[A];
                load_policy_row<A>(pol, (int64_t)sl * 2 + 1, PS, true, p1);
                const int a1 = pick<A>(p1, u[l][1]);
                const UpperWalk *e = w + ((sl * A + a0[l]) * A + a1) * C;
                int chosen = 0;
                if (C > 1) {
                    float ch[RNAD_MAX_TRANSITIONS];
#pragma unroll
                    for (int k = 0; k < RNAD_MAX_TRANSITIONS; ++k) ch[k] = k < C ? e[k].chance : 0.0f;
                    chosen = pick_n<RNAD_MAX_TRANSITIONS>(C, ch, u[l][2]);
                }
                const UpperWalk hit = e[chosen];
                if (on[l]) {
                    if (t + 1 < kPackedSteps) packed[l] |= (unsigned long long)(a1 | (chosen << 3)) << (6 * (t + 1));
                    steps[l] = t + 2;
                    if (hit.next != 0) key[l] = hit.key;  // (a lane that leaves the tree from an upper state keeps that state's bucket)
                    on[l] = hit.next != 0 && hit.key >= n_groups;
                    slot[l] = on[l] ? hit.key - n_groups : 0;
                    if (hit.next != 0 && hit.key < n_groups) root[l] = hit.next;
                }
            }
        }
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int64_t b = b0 + (int64_t)l * kSortThreads;
            if (b < B) {
                keys[b] = key[l];
                decisions[b] = packed[l] | ((unsigned long long)min(steps[l], kPackedSteps) << 60);
                atomicAdd(&cnt[key[l]], 1);
                if (stage.root) {
                    stage.root[b] = (uint32_t)root[l] | ((uint32_t)steps[l] << kStageRootBits);
                    if (root[l]) stage.mark0[root[l]] = stage_stamp(seed);
                }
            }
        }
    }
    __syncthreads();
    int32_t *row = hist + (int64_t)blockIdx.x * n_buckets;
    for (int i = threadIdx.x; i < n_buckets; i += kSortThreads) row[i] = cnt[i];
}

// Hybrid walk for trees whose upper tables exceed the LDS (configs[3]: 2 616 upper states, 3.1 MB of walk table): the upper states of the
// TOP levels that fit (BucketCut::hot_list: whole levels, root first; configs[3]: the root and its 100 children, 128 KB) are staged as in
// k_bucket_keys_lds -- every lane passes through them, in lane order, i.e. with no locality a cache could use beyond one line per lane --
// and a lane that walks on into an upper state below them continues on the global tables as k_bucket_keys does.  Same draws, same
// arithmetic, same keys and decisions as both.  The tile's histogram does not fit NEXT to the tables, so it is built in their place once
// the walk is over (every thread keeps the keys of its lanes): k_bucket_hist's work without its launch.
inline size_t keys_hybrid_lds_bytes(int n_hot, int n_upper, int A, int C) {
    return (((size_t)n_hot * A * A * C * sizeof(UpperWalk) + 15) & ~(size_t)15) + (size_t)n_hot * 2 * ((A + 3) & ~3) * sizeof(float) +
           (size_t)n_upper * sizeof(int32_t);
}

template <int A, int L, int TILE>
__global__ __launch_bounds__(kSortThreads) void k_bucket_keys_hybrid(const UpperWalk *__restrict__ walk, const int32_t *__restrict__ upper_list,
                                                                     const int32_t *__restrict__ hot_list, const int32_t *__restrict__ hot_of_g,
                                                                     int n_hot, int n_upper, const Trans *__restrict__ trans,
                                                                     const int32_t *__restrict__ bucket_of, int C, int64_t S, int64_t B,
                                                                     int n_steps, const float *__restrict__ policy_tab, int64_t tab_stride, int vec4,
                                                                     int key_root, int n_groups, uint64_t seed,
                                                                     const rnad_step_params_t *__restrict__ sp, int64_t lane0,
                                                                     int32_t *__restrict__ keys, unsigned long long *__restrict__ decisions,
                                                                     double *__restrict__ norm, StageOut stage, int n_buckets,
                                                                     int32_t *__restrict__ hist) {
    extern __shared__ __attribute__((aligned(16))) unsigned char keys_smem[];
    constexpr int PS = kPolStride<A>;
    constexpr int kPasses = TILE / (kSortThreads * L);
    int my_keys[kPasses][L];
    const int AAC = A * A * C;
    UpperWalk *w = reinterpret_cast<UpperWalk *>(keys_smem);                                                           // [n_hot][A][A][C]
    float *pol = reinterpret_cast<float *>(keys_smem + (((size_t)n_hot * AAC * sizeof(UpperWalk) + 15) & ~(size_t)15));  // [n_hot][2][PS]
    int32_t *hot_of = reinterpret_cast<int32_t *>(pol + (size_t)n_hot * 2 * PS);                                         // [n_upper]
    for (int i = threadIdx.x; i < n_hot * AAC; i += kSortThreads) w[i] = walk[(int64_t)hot_list[i / AAC] * AAC + i % AAC];
    for (int i = threadIdx.x; i < n_hot * 2 * PS; i += kSortThreads) {
        const int h = i / (2 * PS), player = (i / PS) & 1, a = i % PS;
        pol[i] = a < A ? policy_tab[((int64_t)player * S + upper_list[hot_list[h]]) * tab_stride + a] : 0.0f;
    }
    for (int i = threadIdx.x; i < n_upper; i += kSortThreads) hot_of[i] = hot_of_g[i];
    if (blockIdx.x == 0 && threadIdx.x == 0 && norm) norm[0] = norm[1] = 0.0;  // summed up by k_bucket_alive at the end of this rollout
    if (blockIdx.x == 0 && threadIdx.x == 0 && stage.counts) stage.counts[0] = stage.counts[1] = 0ull;
    if (sp) seed = sp->seed;
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < kPasses; ++pass) {
        const int64_t b0 = (int64_t)blockIdx.x * TILE + (int64_t)pass * (kSortThreads * L) + threadIdx.x;  // lanes b0 + l * kSortThreads
        int state[L], hot[L], key[L], steps[L];  // hot: position of the lane's upper state in the staged tables, or -1
        unsigned long long packed[L];
        bool on[L];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            state[l] = 1;
            key[l] = key_root;
            steps[l] = 0;
            packed[l] = 0ull;
            on[l] = b0 + (int64_t)l * kSortThreads < B && key_root >= n_groups;
            hot[l] = on[l] ? hot_of[key_root - n_groups] : 0;
        }
        for (int t = 0; t < n_steps; t += 2) {
            bool any = false;
#pragma unroll
            for (int l = 0; l < L; ++l) any |= on[l];
            if (!any) break;
            const bool two = t + 1 < n_steps;
            float u[L][3];
            int a0[L];
#pragma unroll
            for (int l = 0; l < L; ++l) rnad_decision_uniforms(seed, (uint64_t)(lane0 + b0 + (int64_t)l * kSortThreads), (uint32_t)t, u[l]);
#pragma unroll
            for (int l = 0; l < L; ++l) {
                float p0[A];
                if (!on[l] || hot[l] >= 0) load_policy_row<A>(pol, (int64_t)(on[l] ? hot[l] : 0) * 2, PS, true, p0);
                else load_policy_row<A>(policy_tab, state[l], tab_stride, vec4 != 0, p0);
                a0[l] = pick<A>(p0, u[l][0]);
                if (on[l]) {
                    if (t < kPackedSteps) packed[l] |= (unsigned long long)a0[l] << (6 * t);
                    steps[l] = t + 1;
                }
            }
            if (!two) break;
#pragma unroll
            for (int l = 0; l < L; ++l) {
                int next, key_next, chosen = 0, a1;
                if (!on[l] || hot[l] >= 0) {
                    const int sl = on[l] ? hot[l] : 0;
                    float p1[A];
                    load_policy_row<A>(pol, (int64_t)sl * 2 + 1, PS, true, p1);
                    a1 = pick<A>(p1, u[l][1]);
                    const UpperWalk *e = w + ((sl * A + a0[l]) * A + a1) * C;
                    if (C > 1) {
                        float ch[RNAD_MAX_TRANSITIONS];
#pragma unroll
                        for (int k = 0; k < RNAD_MAX_TRANSITIONS; ++k) ch[k] = k < C ? e[k].chance : 0.0f;
                        chosen = pick_n<RNAD_MAX_TRANSITIONS>(C, ch, u[l][2]);
                    }
                    const UpperWalk hit = e[chosen];
                    next = hit.next;
                    key_next = hit.key;
                } else {
                    float p1[A], rew;
                    load_policy_row<A>(policy_tab, S + state[l], tab_stride, vec4 != 0, p1);
                    a1 = pick<A>(p1, u[l][1]);
                    transition_lane<A>(trans, C, state[l], a0[l], a1, nullptr, u[l][2], next, rew, &chosen);
                    key_next = bucket_of[next];
                }
                if (on[l]) {
                    if (t + 1 < kPackedSteps) packed[l] |= (unsigned long long)(a1 | (chosen << 3)) << (6 * (t + 1));
                    steps[l] = t + 2;
                    state[l] = next;
                    if (next != 0) key[l] = key_next;  // (a lane that leaves the tree from an upper state keeps that state's bucket)
                    on[l] = next != 0 && key_next >= n_groups;
                    hot[l] = on[l] ? hot_of[key_next - n_groups] : 0;
                }
            }
        }
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int64_t b = b0 + (int64_t)l * kSortThreads;
            my_keys[pass][l] = b < B ? key[l] : -1;
            if (b < B) {
                keys[b] = key[l];
                decisions[b] = packed[l] | ((unsigned long long)min(steps[l], kPackedSteps) << 60);
                if (stage.root) {
                    const int root = key[l] < n_groups ? state[l] : 0;
                    stage.root[b] = (uint32_t)root | ((uint32_t)steps[l] << kStageRootBits);
                    if (root) stage.mark0[root] = stage_stamp(seed);
                }
            }
        }
    }
    // the tile's histogram row, in the LDS the tables occupied
    __syncthreads();
    int32_t *cnt = reinterpret_cast<int32_t *>(keys_smem);
    for (int i = threadIdx.x; i < n_buckets; i += kSortThreads) cnt[i] = 0;
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < kPasses; ++pass)
#pragma unroll
        for (int l = 0; l < L; ++l)
            if (my_keys[pass][l] >= 0) atomicAdd(&cnt[my_keys[pass][l]], 1);
    __syncthreads();
    int32_t *row = hist + (int64_t)blockIdx.x * n_buckets;
    for (int i = threadIdx.x; i < n_buckets; i += kSortThreads) row[i] = cnt[i];
}

// ---------------------------------------------------------------------------------------- 2. stable counting sort by key
// hist[blk][bucket] = #lanes of block blk (kSortLanes consecutive lanes) with that key.
template <int TILE>
__global__ __launch_bounds__(kSortThreads) void k_bucket_hist(int64_t B, int n_buckets, const int32_t *__restrict__ keys,
                                                              int32_t *__restrict__ hist) {
    extern __shared__ int32_t cnt[];
    for (int i = threadIdx.x; i < n_buckets; i += kSortThreads) cnt[i] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * TILE;
#pragma unroll
    for (int r = 0; r < TILE / kSortThreads; ++r) {
        const int64_t b = base + (int64_t)r * kSortThreads + threadIdx.x;
        if (b < B) atomicAdd(&cnt[keys[b]], 1);
    }
    __syncthreads();
    int32_t *row = hist + (int64_t)blockIdx.x * n_buckets;
    for (int i = threadIdx.x; i < n_buckets; i += kSortThreads) row[i] = cnt[i];
}

// Workgroup -> work item, XCD-aware.  The dispatcher places workgroup b on XCD b % 8 (observed, not contractual: a wrong guess is
// slower, not wrong), each XCD has its own 4 MB L2, and the work items are in bucket order, i.e. in the order of the tree's state ids:
// with item = workgroup, every XCD pulls ALL of the tables the lanes gather from (policy rows, transition records, fast records: 8 x 9 MB
// of fabric traffic per rollout launch on configs[1], 8 x 8.5 MB per learner launch).  Here XCD x takes the x-th CONTIGUOUS eighth of the
// items, so its L2 only sees its eighth of the tree.  Bijective for any item count n (the grid holds n + 7 workgroups at least).
// Returns -1 for a workgroup without an item.
#ifndef RNAD_XCD_ITEMS
#define RNAD_XCD_ITEMS 1
#endif
__device__ __forceinline__ int xcd_item(int n) {
#if RNAD_XCD_ITEMS
    const int q = n >> 3, r = n & 7, xcd = (int)blockIdx.x & 7, k = (int)blockIdx.x >> 3;
    const int cnt = q + (xcd < r ? 1 : 0), start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return k < cnt ? start + k : -1;
#else
    return (int)blockIdx.x < n ? (int)blockIdx.x : -1;
#endif
}

// The row list of staging level LEVEL: both players' rows of every state whose stamp (LEVEL 0: its own in mark0; LEVEL 1: its anchor's
// in mark1) is this step's.  One launch: a workgroup counts its states, reserves a range of the list with ONE atomic on the list's
// length and writes its rows there -- the list is unordered across workgroups, which no consumer minds (an actor's forward writes
// logits[row]; the same rows get the same values in any order).  *count was cleared by the keys pass.
constexpr int kStagePer = 4;  // states per thread of a chunk: 4096 per workgroup pass, i.e. a few hundred reservations per list
// one chunk (kSortThreads * kStagePer consecutive states) of the list of level LEVEL, by one workgroup of kSortThreads threads
template <int LEVEL>
__device__ __forceinline__ void stage_rows_chunk(int64_t chunk, int64_t S, const uint32_t *__restrict__ mark, const int32_t *__restrict__ anchor1,
                                                 uint32_t stamp, int32_t *__restrict__ rows, unsigned long long *__restrict__ count) {
    __shared__ int32_t wave_n[kSortThreads / 64];
    __shared__ unsigned long long base_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // a wave takes kStagePer * 64 consecutive states, 64 at a time (coalesced loads of the stamps)
    const int64_t s0 = (chunk * (kSortThreads / 64) + wave) * (kStagePer * 64) + lane;
    uint64_t votes[kStagePer];
    int32_t mine = 0;
#pragma unroll
    for (int r = 0; r < kStagePer; ++r) {
        const int64_t s = s0 + r * 64;
        bool hit = false;
        if (s > 0 && s < S) {
            if (LEVEL == 0) {
                hit = mark[s] == stamp;
            } else {
                const int32_t a = anchor1[s];
                hit = a != 0 && mark[a] == stamp;
            }
        }
        votes[r] = __ballot(hit);
        mine += (int32_t)__popcll(votes[r]);
    }
    if (lane == 0) wave_n[wave] = mine;
    __syncthreads();
    int32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kSortThreads / 64; ++w) {
        if (w < wave) before += wave_n[w];
        total += wave_n[w];
    }
    if (threadIdx.x == 0) base_s = total ? atomicAdd(count, 2ull * (unsigned long long)total) : 0ull;
    __syncthreads();
    unsigned long long at = base_s + 2ull * (unsigned long long)before;
#pragma unroll
    for (int r = 0; r < kStagePer; ++r) {
        if ((votes[r] >> lane) & 1ull) {
            const unsigned long long mine_at = at + 2ull * (unsigned long long)__popcll(votes[r] & ((1ull << lane) - 1ull));
            rows[mine_at] = (int32_t)(s0 + r * 64);
            rows[mine_at + 1] = (int32_t)(S + s0 + r * 64);
        }
        at += 2ull * (unsigned long long)__popcll(votes[r]);
    }
    __syncthreads();  // (wave_n / base_s are reused by the caller's next chunk)
}

template <int LEVEL>
__global__ __launch_bounds__(kSortThreads) void k_stage_rows(int64_t S, const uint32_t *__restrict__ mark, const int32_t *__restrict__ anchor1,
                                                             uint64_t seed, const rnad_step_params_t *__restrict__ sp,
                                                             int32_t *__restrict__ rows, unsigned long long *__restrict__ count) {
    if (sp) seed = sp->seed;
    stage_rows_chunk<LEVEL>(blockIdx.x, S, mark, anchor1, stage_stamp(seed), rows, count);
}

// The transition every lane draws at the root of the group subtree it enters (env steps t, t + 1 with t = the steps it spent above the
// cut): both players' actions from the root's policy rows, the chance outcome -- the draws k_bucket_rollout_items makes there (same
// seed, lane, step -> same uniforms; same rows) -- and a stamp on the state it leads to.  After the sort (it reads lane_ids).
template <int A>
__global__ __launch_bounds__(kThreads) void k_stage_walk(const Trans *__restrict__ trans, int C, int64_t S, int64_t B, int T_cap,
                                                         const float *__restrict__ policy_tab, int64_t tab_stride, int vec4,
                                                         const uint32_t *__restrict__ sorted, const int32_t *__restrict__ lane_ids,
                                                         uint64_t seed, const rnad_step_params_t *__restrict__ sp, int64_t lane0,
                                                         uint32_t *__restrict__ mark1) {
    // (XCD-aware as xcd_item: XCD x walks the x-th contiguous eighth of the bucket-ordered lanes)
    const int nb = (int)gridDim.x, xq = nb >> 3, xr = nb & 7, xcd = (int)blockIdx.x & 7, xk = (int)blockIdx.x >> 3;
    const int64_t blk = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xk;
    if (xk >= xq + (xcd < xr ? 1 : 0)) return;  // (cannot happen: the map is a bijection of [0, nb))
    const int64_t j = blk * kThreads + threadIdx.x;
    if (j >= B) return;
    // in BUCKET order (thread j takes lane lane_ids[j], its root word travelled with it through the sort): the lanes of a wave sit in a few
    // roots of one group, and the rows, the transition records and the stamps they touch share cache lines (in lane order every gather of
    // a wave touched 64 lines: 46 us on configs[3])
    const uint32_t word = sorted[j];
    const int root = (int)(word & ((1u << kStageRootBits) - 1u)), t = (int)(word >> kStageRootBits);  // t: env steps above the group
    if (root == 0 || t + 1 >= T_cap) return;
    if (sp) seed = sp->seed;
    const int64_t b = lane_ids[j];
    float pol0[A], pol1[A], u[3];
    load_policy_row<A>(policy_tab, root, tab_stride, vec4 != 0, pol0);
    load_policy_row<A>(policy_tab, S + root, tab_stride, vec4 != 0, pol1);
    rnad_decision_uniforms(seed, (uint64_t)(lane0 + b), (uint32_t)t, u);
    const int a0 = pick<A>(pol0, u[0]), a1 = pick<A>(pol1, u[1]);
    int next;
    float rew;
    transition_lane<A>(trans, C, root, a0, a1, nullptr, u[2], next, rew);
    if (next != 0) mark1[next] = stage_stamp(seed);
}

// Column-wise exclusive prefix of hist over the blocks (in place) and the column totals.  A workgroup takes kScanCols buckets; its
// threads split the blocks into kSortThreads / kScanCols contiguous parts: partial sums -> prefix over the parts in LDS -> second
// sweep.  (Few rows per thread: the two sweeps are chains of dependent loads.)
constexpr int kScanCols = 16, kScanParts = kSortThreads / kScanCols;

struct Item {
    int32_t begin, count, bucket, single;  // lanes [begin, begin + count) of that bucket; single: the bucket's only item
};

// bucket_start = exclusive prefix of the totals; one work item per `chunk` lanes of a non-empty bucket.  Two phases (every workgroup of
// k_bucket_scatter runs the first for itself -- a launch of its own for it costs more than redoing a 1 000-entry prefix 256 times --,
// workgroup 0 alone the second):
// the prefixes (lanes and items) of all buckets, then the items.  While no bucket has more than kItemsInline items a thread writes
// out the buckets it scanned; otherwise (most of the lanes end up in a few buckets once the policy sharpens: thousands of items in
// one bucket) item k finds its bucket by bisection in the item prefix kept in LDS.
constexpr int kItemsInline = 16;

// start_s (LDS, [n_buckets]): receives bucket_start for the caller.  write_items: this workgroup also writes the work list.
// first_s (LDS, [n_buckets + 1]): scratch, the first item of every bucket (exclusive prefix of the item counts).
__device__ __forceinline__ void items_phase(int n_buckets, int chunk, const int32_t *__restrict__ totals, int32_t *__restrict__ start_s,
                                            int32_t *__restrict__ first_s, bool write_items, Item *__restrict__ items,
                                            int32_t *__restrict__ n_items, int32_t *__restrict__ bucket_start_out = nullptr) {
    __shared__ int32_t wave_l[16], wave_i[16];
    __shared__ int32_t total_s, most;
    if (threadIdx.x == 0) most = 0;
    __syncthreads();
    // a thread takes `per` CONSECUTIVE buckets (one on configs[1]; 12 at the 11 470 buckets of configs[3], where a pass of the workgroup
    // per 1024 buckets with its three barriers each was a third of k_bucket_scatter): local sums, ONE scan over the workgroup, local prefixes
    const int per = (n_buckets + kSortThreads - 1) / kSortThreads;
    const int b0 = min(n_buckets, (int)threadIdx.x * per), b1 = min(n_buckets, b0 + per);
    int32_t own_l = 0, own_i = 0;
    bool big = false;
    for (int i = b0; i < b1; ++i) {
        const int32_t n = totals[i], ni = (n + chunk - 1) / chunk;
        own_l += n;
        own_i += ni;
        big |= ni > kItemsInline;
    }
    if (big) most = kItemsInline + 1;  // (any writer: only "> kItemsInline" matters)
    int32_t sl = own_l, si = own_i;  // inclusive scans within the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t a = __shfl_up(sl, off, 64), b = __shfl_up(si, off, 64);
        if ((int)(threadIdx.x & 63) >= off) {
            sl += a;
            si += b;
        }
    }
    if ((threadIdx.x & 63) == 63) {
        wave_l[threadIdx.x >> 6] = sl;
        wave_i[threadIdx.x >> 6] = si;
    }
    __syncthreads();
    int32_t run_l = sl - own_l, run_i = si - own_i;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) {
        run_l += wave_l[w];
        run_i += wave_i[w];
    }
    if (threadIdx.x == kSortThreads - 1) total_s = run_i + own_i;
    for (int i = b0; i < b1; ++i) {
        const int32_t n = totals[i], ni = (n + chunk - 1) / chunk;
        start_s[i] = run_l;
        first_s[i] = run_i;
        if (write_items && bucket_start_out) bucket_start_out[i] = run_l;  // (global copy: the leaf-path learner finds a bucket's lanes by it)
        run_l += n;
        run_i += ni;
    }
    __syncthreads();
    const int32_t total = total_s;
    if (threadIdx.x == 0) {
        first_s[n_buckets] = total;
        if (write_items) *n_items = total;
    }
    __syncthreads();
    if (!write_items) return;
    if (most <= kItemsInline) {
        for (int i = threadIdx.x; i < n_buckets; i += kSortThreads) {
            const int32_t n = totals[i], first = first_s[i], ni = first_s[i + 1] - first, start = start_s[i];
            for (int32_t j = 0; j < ni; ++j) items[first + j] = Item{start + j * chunk, min(chunk, n - j * chunk), i, ni == 1 ? 1 : 0};
        }
        return;
    }
    for (int32_t k = threadIdx.x; k < total; k += kSortThreads) {
        int lo = 0, hi = n_buckets;  // the last bucket with first_s[bucket] <= k (empty buckets share their successor's prefix)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (first_s[mid] <= k) lo = mid;
            else hi = mid;
        }
        const int32_t n = totals[lo], j = k - first_s[lo], start = start_s[lo];
        items[k] = Item{start + j * chunk, min(chunk, n - j * chunk), lo, n <= chunk ? 1 : 0};
    }
}

// (The items were tried in this kernel's last workgroup, elected by a ticket: the device-scope release / acquire fences that the
// election needs -- the workgroups sit on different XCDs -- write back whatever the previous kernels left dirty in the L2s:
// 14.5 us against 5.2 + 5.6 for two launches.  They ride in k_bucket_scatter instead.)
// r05: the launch also carries the copies of the distinct observations' records (KeysExpand: nothing between the table launch and the
// rollout reads the rows of non-representatives -- the keys pass reads the upper rows through rep_of) in workgroups of its own behind the
// scan's: the scan is a launch floor with a handful of workgroups, the copies fill the rest of the chip at memory speed.
__global__ __launch_bounds__(kSortThreads) void k_bucket_scan(int n_blocks, int n_buckets, int32_t *__restrict__ hist,
                                                              int32_t *__restrict__ totals, int scan_blocks, KeysExpand ex) {
    if ((int)blockIdx.x >= scan_blocks) {
        keys_expand_share(ex, kSortThreads, (int)blockIdx.x - scan_blocks, (int)gridDim.x - scan_blocks);
        return;
    }
    __shared__ int32_t part[kScanParts][kScanCols + 1];
    const int col = threadIdx.x & (kScanCols - 1), g = threadIdx.x / kScanCols;
    const int c = blockIdx.x * kScanCols + col;
    const int per = (n_blocks + kScanParts - 1) / kScanParts, r0 = min(n_blocks, g * per), r1 = min(n_blocks, r0 + per);
    // (a thread's rows stay in registers between the two sweeps when there are at most kScanKeep of them: one read of the matrix instead
    // of two -- 11.7 MB at the 11 470 buckets of configs[3])
    constexpr int kScanKeep = 16;
    const bool keep = per <= kScanKeep;
    int32_t v[kScanKeep];
    int32_t sum = 0;
    if (c < n_buckets) {
        if (keep) {
#pragma unroll
            for (int k = 0; k < kScanKeep; ++k) v[k] = r0 + k < r1 ? hist[(int64_t)(r0 + k) * n_buckets + c] : 0;
#pragma unroll
            for (int k = 0; k < kScanKeep; ++k) sum += v[k];
        } else {
            for (int r = r0; r < r1; ++r) sum += hist[(int64_t)r * n_buckets + c];
        }
    }
    part[g][col] = sum;
    __syncthreads();
    if (g == 0) {  // one thread per column: exclusive prefix over the parts, total
        int32_t run = 0;
        for (int i = 0; i < kScanParts; ++i) {
            const int32_t x = part[i][col];
            part[i][col] = run;
            run += x;
        }
        if (c < n_buckets) totals[c] = run;
    }
    __syncthreads();
    if (c < n_buckets) {
        int32_t before = part[g][col];
        if (keep) {
#pragma unroll
            for (int k = 0; k < kScanKeep; ++k) {
                if (r0 + k < r1) hist[(int64_t)(r0 + k) * n_buckets + c] = before;
                before += v[k];
            }
        } else {
            for (int r = r0; r < r1; ++r) {
                const int32_t x = hist[(int64_t)r * n_buckets + c];
                hist[(int64_t)r * n_buckets + c] = before;
                before += x;
            }
        }
    }
}

// lane_ids[bucket_start[key] + (lanes of earlier blocks with that key) + (earlier lanes of this block with that key)] = lane.
// The 16 waves of a workgroup each own 256 consecutive lanes.  They rank their lanes against `wave_rows` counter rows in LDS (16, 8, 4, 2
// or 1: as many as fit beside the bucket prefixes -- 16 up to ~1 300 buckets, 8 up to ~3 600): row r serves the waves 16 r / R .. 16 (r + 1)
// / R - 1, which take turns in wave order; within one LDS atomic instruction the lanes that hit the same counter are served in lane
// order, and a wave's instructions are issued in order.  The rows are then turned into their starting offsets -- row r starts where rows
// 0 .. r - 1 end -- and the ranks become positions: the stable counting sort, the same permutation on every run.
// bucket_start (the exclusive prefix of the column totals) is taken by every workgroup for itself; workgroup 0 also writes the
// learner's work list (items_phase).
// Staged actors (trees that are large next to the batch; rnad_bucket_sort): staged_rows receives, ascending, both players' rows of every
// state inside a group some lane descends into (totals[group] > 0) -- what the actor still has to be evaluated on before the rollout --
// and *n_staged their number: every workgroup takes the exclusive prefix of the non-empty groups' spans for itself and writes the
// groups blockIdx.x, blockIdx.x + gridDim.x, ...  (Before r04: k_group_flags + a three-launch compaction of 2S flags.)  visited (int32
// [2S], optional) is cleared here for the rollout that follows (the two rows of the absorbing state are set: absorbed slots show them).
template <int TILE>
__global__ __launch_bounds__(kSortThreads) void k_bucket_scatter(int64_t B, int n_buckets, const int32_t *__restrict__ keys,
                                                                 const int32_t *__restrict__ hist, const int32_t *__restrict__ totals, int chunk,
                                                                 Item *__restrict__ items, int32_t *__restrict__ n_items,
                                                                 int32_t *__restrict__ lane_ids, int wave_rows, int64_t S, int n_groups,
                                                                 const int32_t *__restrict__ bucket_lo, const int32_t *__restrict__ bucket_span,
                                                                 const int32_t *__restrict__ group_by_lo, int32_t *__restrict__ staged_rows,
                                                                 int64_t *__restrict__ n_staged, int32_t *__restrict__ visited,
                                                                 const uint32_t *__restrict__ stage_root, uint32_t *__restrict__ stage_sorted,
                                                                 const uint32_t *__restrict__ stage_mark0, int32_t *__restrict__ stage_rows0,
                                                                 unsigned long long *__restrict__ stage_count0, uint64_t seed,
                                                                 const rnad_step_params_t *__restrict__ sp, int32_t *__restrict__ bucket_start_out) {
    extern __shared__ int32_t cnt[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t base = (int64_t)blockIdx.x * TILE + (int64_t)wave * (TILE / 16);
    int32_t key[TILE / kSortThreads];
    uint32_t root_word[TILE / kSortThreads];  // (second staging level: the lanes' root words go through the same permutation)
#pragma unroll
    for (int r = 0; r < TILE / kSortThreads; ++r) {
        const int64_t b = base + r * 64 + lane;
        key[r] = b < B ? keys[b] : -1;
        root_word[r] = (stage_root && b < B) ? stage_root[b] : 0u;
    }
    if (visited) {
        for (int64_t r = (int64_t)blockIdx.x * kSortThreads + threadIdx.x; r < 2 * S; r += (int64_t)gridDim.x * kSortThreads)
            visited[r] = (r == 0 || r == S) ? 1 : 0;
    }
    if (stage_rows0) {  // second staging level: the row list of the roots the keys pass stamped (k_stage_rows<0>'s work without its launch)
        if (sp) seed = sp->seed;
        const int64_t n_chunks = (S + kSortThreads * kStagePer - 1) / (kSortThreads * kStagePer);
        for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x)
            stage_rows_chunk<0>(chunk, S, stage_mark0, nullptr, stage_stamp(seed), stage_rows0, stage_count0);
    }
    if (staged_rows) {
        // exclusive prefix over the groups of (non-empty ? span : 0), in the LDS words the sort uses afterwards
        __shared__ int32_t wave_s[16];
        __shared__ int32_t carry_s;
        int32_t *pre = cnt;  // [n_groups + 1]
        if (threadIdx.x == 0) carry_s = 0;
        __syncthreads();
        for (int k0 = 0; k0 < n_groups; k0 += kSortThreads) {  // k: position in ascending order of the groups' first state
            const int k = k0 + threadIdx.x;
            const int g = k < n_groups ? group_by_lo[k] : 0;
            const int32_t n = (k < n_groups && totals[g] > 0) ? bucket_span[g] : 0;
            int32_t incl = n;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int32_t o = __shfl_up(incl, off, 64);
                if (lane >= off) incl += o;
            }
            if (lane == 63) wave_s[wave] = incl;
            __syncthreads();
            int32_t before = carry_s;
            for (int w = 0; w < wave; ++w) before += wave_s[w];
            if (k < n_groups) pre[k] = before + incl - n;
            __syncthreads();
            if (threadIdx.x == kSortThreads - 1) carry_s = before + incl;
            __syncthreads();
        }
        const int32_t total = carry_s;
        if (blockIdx.x == 0 && threadIdx.x == 0) *n_staged = 2 * (int64_t)total;
        for (int k = blockIdx.x * 16 + wave; k < n_groups; k += gridDim.x * 16) {  // a wave per group
            const int g = group_by_lo[k];
            if (totals[g] <= 0) continue;
            const int32_t lo = bucket_lo[g], n = bucket_span[g], at = pre[k];
            for (int j = lane; j < n; j += 64) {
                staged_rows[at + j] = lo + j;
                staged_rows[total + at + j] = (int32_t)S + lo + j;
            }
        }
        __syncthreads();
    }
    // (this tile's row of the prefixed histogram, requested before the prefix of the totals instead of after it: one round trip to
    // memory less on the workgroup's critical path)
    const int32_t *row = hist + (int64_t)blockIdx.x * n_buckets;
    const int32_t row_first = (int)threadIdx.x < n_buckets ? row[threadIdx.x] : 0;
    items_phase(n_buckets, chunk, totals, cnt, cnt + n_buckets, blockIdx.x == 0, items, n_items, bucket_start_out);  // cnt = bucket_start
    __syncthreads();
    if ((int)threadIdx.x < n_buckets) cnt[threadIdx.x] += row_first;
    for (int i = threadIdx.x + kSortThreads; i < n_buckets; i += kSortThreads) cnt[i] += row[i];
    __syncthreads();
    // a lane's position = the bucket's start + the tile's offset (cnt) + the lanes of the tile's earlier counter rows + its rank in its row
    const int R = wave_rows, G = 16 / R, my_row = wave / G;
    int32_t *wcnt = cnt + 2 * n_buckets + 1;  // [R][n_buckets]
    for (int i = threadIdx.x; i < R * n_buckets; i += kSortThreads) wcnt[i] = 0;
    __syncthreads();
    int32_t rank[TILE / kSortThreads];
    for (int turn = 0; turn < G; ++turn) {  // the waves of a row, in wave order
        if (wave % G == turn) {
#pragma unroll
            for (int r = 0; r < TILE / kSortThreads; ++r) rank[r] = key[r] >= 0 ? atomicAdd(&wcnt[my_row * n_buckets + key[r]], 1) : 0;
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < n_buckets; i += kSortThreads) {
        int32_t at = cnt[i];
        for (int w = 0; w < R; ++w) {
            const int32_t n = wcnt[w * n_buckets + i];
            wcnt[w * n_buckets + i] = at;
            at += n;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < TILE / kSortThreads; ++r)
        if (key[r] >= 0) {
            const int32_t at = wcnt[my_row * n_buckets + key[r]] + rank[r];
            lane_ids[at] = (int32_t)(base + r * 64 + lane);
            if (stage_sorted) stage_sorted[at] = root_word[r];
        }
}

// ---------------------------------------------------------------------------------------- 3. rollout in bucket order
// Episodes.generate (episode.py:194-212) for thread j = lane lane_ids[j]: per env step the tabular actor's policy row of
// (player to move, state) (the policy head of net.py:45-46, evaluated once per row by k_policy_rows / k_row_records instead of
// once per slot: same function, same input, same bits), the seeded draw (net.py:49; include/rnad_rng.h), the record, and on the column player's
// turn the chance draw and transition (episode.py:106-121) -- the arithmetic of k_act, with state and the row action kept in
// registers across the T_cap steps and column j of every [T_cap, B] buffer written coalesced.  Observations are not written
// (a function of (t & 1, indices): materialised on demand), `values` only if asked for.
// alive_part[block][t] = #lanes of the block with indices[t] != 0 (summed by k_bucket_alive: no atomics).
template <int A>
__global__ __launch_bounds__(kThreads) void k_bucket_rollout(const Trans *__restrict__ trans, int C, int64_t S, int64_t B, int T_cap,
                                                             const float *__restrict__ policy_tab, int64_t tab_stride,
                                                             const float *__restrict__ value_tab, int64_t value_stride,
                                                             const uint8_t *__restrict__ mask_tab, uint64_t seed,
                                                             const rnad_step_params_t *__restrict__ sp, int64_t lane0,
                                                             const int32_t *__restrict__ lane_ids,
                                                             const unsigned long long *__restrict__ decisions, int32_t *__restrict__ indices,
                                                             uint8_t *__restrict__ mbits, float *__restrict__ policy,
                                                             int32_t *__restrict__ actions, float *__restrict__ rewards,
                                                             float *__restrict__ values, int32_t *__restrict__ alive_part) {
    __shared__ int32_t cnt[kThreads / 64][kMaxSteps + 1];
    const int64_t j = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const bool active = j < B;
    if (sp) seed = sp->seed;
    const int32_t lane_local = active ? lane_ids[j] : 0;
    const uint64_t lane = (uint64_t)(lane0 + lane_local);
    const unsigned long long packed = active ? decisions[lane_local] : 0ull;  // the lane's first decisions, drawn by k_bucket_keys
    const int n_packed = (int)(packed >> 60);
    const int wave = threadIdx.x >> 6;
    int state = 1, prev = 0;
    float u[3] = {0.0f, 0.0f, 0.0f};  // the seeded uniforms of the current game transition
    for (int t = 0; t < T_cap; ++t) {
        const uint64_t live = __ballot(active && state != 0);
        if ((threadIdx.x & 63) == 0) cnt[wave][t] = (int32_t)__popcll(live);
        if (active) {
            if (!(t & 1) && t + 1 >= n_packed) rnad_decision_uniforms(seed, lane, (uint32_t)t, u);  // (some step of it is not a replay)
            const int64_t i = (int64_t)t * B + j;
            const int64_t row = (int64_t)(t & 1) * S + state;
            const bool replay = t < n_packed;
            const int bits6 = replay ? (int)(packed >> (6 * t)) & 63 : 0;
            int action = bits6 & 7;
            const uint32_t bits = mask_tab[row];
            float pol[A];
#pragma unroll
            for (int a = 0; a < A; ++a) pol[a] = policy_tab[row * tab_stride + a];
            if (!replay) action = pick<A>(pol, u[t & 1]);
            indices[i] = state;
            mbits[i] = (uint8_t)bits;
#pragma unroll
            for (int a = 0; a < A; ++a) policy[i * A + a] = pol[a];
            actions[i] = action;
            if (values) values[i] = value_tab ? value_tab[row * value_stride] : 0.0f;
            int next = state;
            float rew = 0.0f;  // row turn: torch.zeros (episode.py:101)
            if (t & 1) {
                if (replay)
                    transition_apply<A>(trans, C, state, prev, action, bits6 >> 3, next, rew);
                else
                    transition_lane<A>(trans, C, state, prev, action, nullptr, u[2], next, rew);
            } else {
                prev = action;
            }
            rewards[i] = rew;
            state = next;
        }
    }
    const uint64_t live = __ballot(active && state != 0);
    if ((threadIdx.x & 63) == 0) cnt[wave][T_cap] = (int32_t)__popcll(live);
    if (active) indices[(int64_t)T_cap * B + j] = state;
    __syncthreads();
    if ((int)threadIdx.x <= T_cap) {
        int32_t s = 0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) s += cnt[w][threadIdx.x];
        alive_part[(int64_t)blockIdx.x * (T_cap + 1) + threadIdx.x] = s;
    }
}

// The COMPACT rollout: the trajectory as 32 bytes per lane instead of 25 per slot.  A lane of bucket b sits in path_states[b][t] at
// every step t < n_shared(b) -- above its group, and at the root of a group that is one subtree -- so nothing is stored for those
// steps; below, a state is bucket_lo[b] + (0 .. rows - 1): one byte (REL = uint8_t while the cut's tables have <= 255 rows, else
// uint16_t), `state - bucket_lo + 1`, 0 = absorbed.  states [T_cap + 1, B] of REL (rows t < n_shared of a column are never written nor
// read), the lane's actions packed 3 bits per step (acts_out) and the one non-zero reward of the episode (rewards *= (indices == 0),
// episode.py:120-121: only the transition into state 0 pays; reward_out).  Everything else of a slot is a function of (t & 1, state)
// and the actor's table (k_bucket_indices / k_bucket_expand write the dense buffers on demand).
//
// One workgroup per work item (the learner's list: <= chunk lanes of one bucket).  Every lane of a bucket that is ONE subtree took the
// same path to its root (ids are DFS pre-order: a state has exactly one parent entry), so the steps above the cut are played once per
// workgroup from the first lane's decision word -- uniform values, scalar loads -- instead of a random 8-byte gather (128 B of fabric
// traffic) and a replay per lane; below the cut every lane draws for itself.  One TRANSITION (the row player's step and the column
// player's step in the same state) per iteration: both policy rows of the state are requested together and the transition's uniforms
// are computed while they travel, so a transition costs two dependent memory latencies (policy rows, transition record) instead of
// three.  Same draws (keyed by lane and step), same episodes as k_bucket_rollout.  An absorbed lane stops: no draws, no table reads.
// Workgroups beyond the item count leave zero alive counts.
// (Staging the group's transition records and policy rows in LDS for the drawn steps was measured: 42.0 us against 36.3 without --
// 12.7 KB of copies per 256 lanes cost more than the gathers they replace.)
// base + a 32-bit BYTE offset: the form the compiler turns into `global_load ... v_off, s[base:base+1]` (no 64-bit VALU address
// arithmetic); every table of the pipeline is smaller than 4 GB
template <typename T>
__device__ __forceinline__ const T *at_bytes(const void *base, uint32_t byte_offset) {
    return reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_offset);
}
template <typename T>
__device__ __forceinline__ T *at_bytes(void *base, uint32_t byte_offset) {
    return reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byte_offset);
}

template <typename REL>
__device__ __forceinline__ REL rel_of(int state, int lo) {
    return (REL)(state == 0 ? 0 : state - lo + 1);
}

// L lanes per thread (workgroups of kThreads / L threads, lanes l * NT + thread of the item's pass; build switch).  The kernel spends 59 %
// of a resident wave's time on s_waitcnt with the SIMDs' wave slots full (41 VGPRs), so more independent chains per thread looked like
// the lever -- measured on configs[1] / configs[3]: L = 1 27.0 / 61.8 us, L = 2 28.2 / 62.4, L = 4 36.8 / 80.3.  It is not the number of
// lanes in flight that bounds it.
#ifndef RNAD_ROLLOUT_LANES
#define RNAD_ROLLOUT_LANES 1
#endif
constexpr int kRolloutLanes = RNAD_ROLLOUT_LANES;

// What a thread of k_bucket_play_learn hands from its rollout to its update in registers (single-pass items: lane item.begin + thread):
// the packed actions, the reward, the relative state of the last four stored transitions (16 bits each, the latest in the low bits) and
// the state the episode ended in -- the values the update would otherwise read back from the trajectory it has just written.
struct Hand {
    unsigned long long acts, relq;
    float reward;
    int final_rel;
};
constexpr int kHandPairs = 4;

// Distinct trajectories of a work item (k_bucket_play_learn).  A state has exactly one parent entry (ids are DFS pre-order), so the state a
// lane was last alive in, together with the outcome it drew there -- (row action, column action, chance) into the absorbing state, or
// "still in the tree after T_cap steps" -- fixes its whole trajectory: states, actions, reward.  Lanes of an item with the same key
// therefore add the SAME addends to the same rows, and the update of the item is  sum over its distinct keys of  count x addends  -- in
// 64-bit integers: the bits of the sum over the lanes.  The rollout counts the lanes per key in LDS and lists, for every key it sees
// first, one lane that has it; the learner runs on the list.
struct Distinct {  // (passed by value: a pointer to it would keep the struct in scratch memory)
    int32_t *count = nullptr;   // LDS [rows of the group][A * A * C + 1]: lanes of the item per key (zero before the rollout); NULL: off
    int32_t *n = nullptr;       // LDS: keys listed so far
    int32_t *key = nullptr;     // LDS [capacity]: the listed keys ...
    int32_t *column = nullptr;  // ... and the trajectory column (bucket order) of the lane that listed each
    int codes = 0;              // A * A * C + 1
    int keys = 0;               // counters; count[keys] stays 1: the weight of a lane listed on its own (a key outside the table: not expected)
};

// r06, leaf paths under a sharp policy: LeafCount -- a bucket that holds at least `crowded_lanes` lanes (and at most kLeafHist columns) is
// counted HERE, by the work items that play its lanes (a histogram over the bucket's columns in LDS, its non-zero bins added to col_count
// with one global atomic each: lanes that pile up on few trajectories cost a handful of atomics per item), instead of by every learner work
// item of the bucket scanning all of the bucket's lanes (k_bucket_learn_c<WEIGHTED>: items x lanes of the bucket -- what made the leaf-path
// learner lose to the per-lane one once a policy had sharpened).  Both kernels take the decision from the sort's
// bucket totals, so they agree.
constexpr int kLeafHist = 1024;
struct LeafCount {
    int32_t *col_count = nullptr;          // [n_cols] lanes per column, of crowded buckets only; the learner clears what it reads
    const int32_t *bucket_col0 = nullptr;  // [n_buckets + 1] first column of a bucket
    const int32_t *lane_total = nullptr;   // [n_buckets] the sort's totals
    int32_t crowded_lanes = 0;
};
__device__ __forceinline__ bool leaf_crowded(const LeafCount &lc, int bucket) {
    return lc.col_count != nullptr && lc.crowded_lanes > 0 && lc.lane_total[bucket] >= lc.crowded_lanes &&
           lc.bucket_col0[bucket + 1] - lc.bucket_col0[bucket] <= kLeafHist;
}

template <int A, typename REL, int L, bool DISTINCT = false, bool HIST = false>
__device__ __forceinline__ void rollout_items_body(const Trans *__restrict__ trans, int C, int64_t S, int64_t B, int T_cap,
                                                                   const float *__restrict__ policy_tab, int64_t tab_stride, int vec4,
                                                                   uint64_t seed, const rnad_step_params_t *__restrict__ sp, int64_t lane0,
                                                                   const int32_t *__restrict__ lane_ids,
                                                                   const unsigned long long *__restrict__ decisions,
                                                                   const Item *__restrict__ items, const int32_t *__restrict__ n_items,
                                                                   const int32_t *__restrict__ bucket_path, const int32_t *__restrict__ bucket_lo,
                                                                   const int32_t *__restrict__ path_states, int path_stride, int n_groups,
                                                                   REL *__restrict__ states, int32_t *__restrict__ alive_part,
                                                                   unsigned long long *__restrict__ acts_out,
                                                                   float *__restrict__ reward_out, int32_t *__restrict__ visited,
                                                                   double *__restrict__ norm_rep = nullptr, Hand *hand = nullptr,
                                                                   const Distinct distinct = Distinct{},
                                                                   const int32_t *__restrict__ col_of = nullptr,
                                                                   int32_t *__restrict__ leaf_col = nullptr, const LeafCount lc = LeafCount{}) {
    constexpr int NT = kThreads / L, NW = NT / 64;
    __shared__ int32_t leaf_hist[HIST ? kLeafHist : 1];
    static_assert(NT >= 64 && NT > kCompactSteps, "a wave at least, and a thread per alive counter");
    __shared__ int32_t cnt[NW][kMaxSteps + 1];
    // a row per WORKGROUP (the sum over the rows does not care which item it held) -- or, norm_rep given (k_bucket_play_learn: no launch
    // between the rollout and the finish to add the rows up in), atomics into one of kReplicas rows of counts, which k_bucket_finish sums
    int32_t *my_alive = alive_part + (norm_rep ? (int64_t)(blockIdx.x & (kReplicas - 1)) : (int64_t)blockIdx.x) * (T_cap + 1);
    const int my_item = xcd_item(*n_items);
    if (my_item < 0) {
        if (!norm_rep && (int)threadIdx.x <= T_cap) my_alive[threadIdx.x] = 0;
        return;
    }
    const Item item = items[my_item];
    if (sp) seed = sp->seed;
    const int wave = threadIdx.x >> 6;
    const int path_word = bucket_path[item.bucket];
    const bool shared = item.bucket < n_groups && (path_word & kSharedRoot) != 0;
    const int n_shared = (path_word & (kSharedRoot - 1)) + (shared ? 2 : 0);  // steps whose state is the bucket's, not the lane's
    const int lo = bucket_lo[item.bucket];
    const int32_t *my_path = path_states + (int64_t)item.bucket * path_stride;
    const uint32_t B32 = (uint32_t)B;
    if ((threadIdx.x & 63) == 0)
        for (int t = 0; t <= T_cap; ++t) cnt[wave][t] = 0;
    bool count_here = false;
    int32_t col0 = 0, n_bins = 0;
    if constexpr (HIST) {
        count_here = leaf_crowded(lc, item.bucket);
        if (count_here) {
            col0 = lc.bucket_col0[item.bucket];
            n_bins = lc.bucket_col0[item.bucket + 1] - col0;
            for (int i = threadIdx.x; i < n_bins; i += NT) leaf_hist[i] = 0;
            __syncthreads();
        }
    }
    for (int base = 0; base < item.count; base += kThreads) {  // (one pass: an item holds <= chunk = kThreads lanes unless RNAD_BUCKET_CHUNK says otherwise)
        bool active[L];
        uint32_t j[L];
        int32_t lane_local[L];
        uint64_t lane[L];
        unsigned long long packed[L];  // the first decisions, drawn by k_bucket_keys: of the bucket (shared) or of this lane
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int k = base + l * NT + (int)threadIdx.x;
            active[l] = k < item.count;
            j[l] = (uint32_t)(item.begin + k);
            lane_local[l] = active[l] ? lane_ids[j[l]] : 0;
            lane[l] = (uint64_t)(lane0 + lane_local[l]);
        }
        if (shared) {
            const unsigned long long first = decisions[lane_ids[item.begin]];
            const unsigned long long uni = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(first >> 32)) << 32) |
                                           (unsigned int)__builtin_amdgcn_readfirstlane((int)first);
#pragma unroll
            for (int l = 0; l < L; ++l) packed[l] = uni;
        } else {
#pragma unroll
            for (int l = 0; l < L; ++l) packed[l] = active[l] ? decisions[lane_local[l]] : 0ull;
        }
        int state[L], n_packed[L], t_from = 0;
        unsigned long long relq = 0ull;  // (hand: lane 0 of the thread)
        int last_key[L];                 // (distinct: the key of the lane's trajectory, -1 while it is in the tree)
        unsigned long long acts[L];
        float reward_final[L];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            n_packed[l] = (int)(packed[l] >> 60);
            last_key[l] = -1;
            state[l] = active[l] ? 1 : 0;
            acts[l] = 0ull;
            reward_final[l] = 0.0f;
        }
        if (shared) {  // the transitions above the cut, once for the workgroup: the states are the bucket's path, the actions the first lane's
            const int n_pre = min(min(n_packed[0], T_cap) & ~1, n_shared - 2);
            int32_t live_now = 0;
#pragma unroll
            for (int l = 0; l < L; ++l) live_now += (int32_t)__popcll(__ballot(active[l]));
            for (int t = 0; t < n_pre; t += 2) {
                if ((threadIdx.x & 63) == 0) {
                    cnt[wave][t] += live_now;
                    cnt[wave][t + 1] += live_now;
                }
                const int bits0 = (int)(packed[0] >> (6 * t)) & 63, bits1 = (int)(packed[0] >> (6 * (t + 1))) & 63;
                if (visited && threadIdx.x == 0 && base == 0) {
                    const int at = my_path[t];
                    visited[at] = visited[S + at] = 1;
                }
#pragma unroll
                for (int l = 0; l < L; ++l)
                    acts[l] |= (unsigned long long)(bits0 & 7) << (3 * t) | (unsigned long long)(bits1 & 7) << (3 * (t + 1));
            }
            t_from = n_pre;
#pragma unroll
            for (int l = 0; l < L; ++l) state[l] = active[l] ? my_path[n_pre] : 0;  // (n_pre <= n_path: an upper state of the path, or the group's root)
        }
        for (int t = t_from; t < T_cap; t += 2) {
            const bool two = t + 1 < T_cap;  // (an odd T_cap ends with a row step alone)
            int32_t live = 0;
#pragma unroll
            for (int l = 0; l < L; ++l) live += (int32_t)__popcll(__ballot(state[l] != 0));
            if ((threadIdx.x & 63) == 0) {
                cnt[wave][t] += live;
                if (two) cnt[wave][t + 1] += live;  // the row player's step leaves the state as it is
            }
            bool go[L], replay0[L], replay1[L];
            float pol0[L][A], pol1[L][A], u[L][3];
            // every lane's stores and table requests first, then the draws: the L chains overlap their two dependent latencies
#pragma unroll
            for (int l = 0; l < L; ++l) {
                go[l] = active[l];
                if (go[l] && t >= n_shared) {  // (n_shared is even: both steps of the transition, or neither)
                    const REL r = rel_of<REL>(state[l], lo);
                    if (hand && l == 0) relq = (relq << 16) | (unsigned long long)r;
                    *at_bytes<REL>(states, ((uint32_t)t * B32 + j[l]) * (uint32_t)sizeof(REL)) = r;
                    if (two) *at_bytes<REL>(states, ((uint32_t)(t + 1) * B32 + j[l]) * (uint32_t)sizeof(REL)) = r;
                }
                go[l] = go[l] && state[l] != 0;
                replay0[l] = t < n_packed[l];
                replay1[l] = t + 1 < n_packed[l];
                if (go[l]) {
                    const int64_t row0 = state[l], row1 = S + state[l];
                    if (visited) {
                        visited[row0] = 1;  // (every writer stores the same value)
                        if (two) visited[row1] = 1;
                    }
                    if (!replay0[l]) load_policy_row<A>(policy_tab, row0, tab_stride, vec4 != 0, pol0[l]);
                    if (two && !replay1[l]) load_policy_row<A>(policy_tab, row1, tab_stride, vec4 != 0, pol1[l]);
                }
            }
#pragma unroll
            for (int l = 0; l < L; ++l)
                if (go[l] && (!replay0[l] || (two && !replay1[l]))) rnad_decision_uniforms(seed, lane[l], (uint32_t)t, u[l]);  // computed while the rows travel
            int a0[L], a1[L], bits1[L];
#pragma unroll
            for (int l = 0; l < L; ++l) {
                if (!go[l]) continue;
                const int bits0 = replay0[l] ? (int)(packed[l] >> (6 * t)) & 63 : 0;
                bits1[l] = replay1[l] ? (int)(packed[l] >> (6 * (t + 1))) & 63 : 0;
                a0[l] = replay0[l] ? (bits0 & 7) : pick<A>(pol0[l], u[l][0]);
                acts[l] |= (unsigned long long)a0[l] << (3 * t);
                if (two) {
                    a1[l] = replay1[l] ? (bits1[l] & 7) : pick<A>(pol1[l], u[l][1]);
                    acts[l] |= (unsigned long long)a1[l] << (3 * (t + 1));
                }
            }
            if (two) {
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    if (!go[l]) continue;
                    int next, outcome;
                    float rew;
                    if (replay1[l]) {
                        outcome = bits1[l] >> 3;
                        transition_apply<A>(trans, C, state[l], a0[l], a1[l], outcome, next, rew);
                    } else {
                        transition_lane<A>(trans, C, state[l], a0[l], a1[l], nullptr, u[l][2], next, rew, &outcome);
                    }
                    if (next == 0) {
                        reward_final[l] = rew;
                        if (DISTINCT) last_key[l] = (state[l] - lo) * distinct.codes + (a0[l] * A + a1[l]) * C + outcome;
                        if (leaf_col)
                            // leaf paths (rnad_leaf_paths_t): the transition a lane leaves the tree by fixes its whole trajectory (a state has one
                            // parent entry) -- the column of that trajectory, per lane in bucket order; the learner's workgroups count the lanes
                            // of their columns in LDS (a global counter per column took a device-scope atomic per lane: 15 us per 2^20 lanes)
                        {
                            const int32_t col = col_of[(((uint32_t)state[l] * A + a0[l]) * A + a1[l]) * C + outcome];
                            leaf_col[j[l]] = col;
                            if constexpr (HIST)
                                if (count_here) atomicAdd(&leaf_hist[col - col0], 1);
                        }

                    }
                    state[l] = next;
                }
            }
        }
        int32_t live = 0;
#pragma unroll
        for (int l = 0; l < L; ++l) {
            if (active[l]) {
                acts_out[j[l]] = acts[l];
                reward_out[j[l]] = reward_final[l];
                if (T_cap >= n_shared) *at_bytes<REL>(states, ((uint32_t)T_cap * B32 + j[l]) * (uint32_t)sizeof(REL)) = rel_of<REL>(state[l], lo);
            }
            live += (int32_t)__popcll(__ballot(state[l] != 0));
        }
        if ((threadIdx.x & 63) == 0) cnt[wave][T_cap] += live;
        if (DISTINCT) {
#pragma unroll
            for (int l = 0; l < L; ++l) {
                if (!active[l]) continue;
                const int key = state[l] != 0 ? (state[l] - lo) * distinct.codes + distinct.codes - 1 : last_key[l];
                const bool in_table = key >= 0 && key < distinct.keys;
                if (!in_table || atomicAdd(distinct.count + key, 1) == 0) {  // the first lane with this trajectory lists it
                    const int at = atomicAdd(distinct.n, 1);
                    distinct.key[at] = in_table ? key : distinct.keys;
                    distinct.column[at] = (int32_t)j[l];
                }
            }
        }
        if (hand) *hand = Hand{acts[0], relq, reward_final[0], active[0] ? (int)rel_of<REL>(state[0], lo) : 0};
    }
    __syncthreads();
    if constexpr (HIST) {
        if (count_here) {
            for (int i = threadIdx.x; i < n_bins; i += NT) {
                const int32_t n = leaf_hist[i];
                if (n != 0) atomicAdd(lc.col_count + col0 + i, n);
            }
        }
    }
    if ((int)threadIdx.x <= T_cap) {
        int32_t sum = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += cnt[w][threadIdx.x];
        if (!norm_rep) my_alive[threadIdx.x] = sum;
        else if (sum != 0) atomicAdd(my_alive + threadIdx.x, sum);
        if (norm_rep) cnt[0][threadIdx.x] = sum;  // (every wave's count of this step has been read: by this very thread)
    }
    if (norm_rep) {  // N_P of the workgroup's lanes: the live slots of parity P (alive[T_cap]: after the last step, not a slot)
        __syncthreads();
        if (threadIdx.x < 2) {
            int32_t n = 0;
            for (int t = threadIdx.x; t < T_cap; t += 2) n += cnt[0][t];
            if (n != 0) atomicAdd(norm_rep + 2 * (blockIdx.x & (kReplicas - 1)) + threadIdx.x, (double)n);
        }
    }
}

template <int A, typename REL, int L>
__global__ __launch_bounds__(kThreads / L) void k_bucket_rollout_items(const Trans *__restrict__ trans, int C, int64_t S, int64_t B, int T_cap,
                                                                   const float *__restrict__ policy_tab, int64_t tab_stride, int vec4,
                                                                   uint64_t seed, const rnad_step_params_t *__restrict__ sp, int64_t lane0,
                                                                   const int32_t *__restrict__ lane_ids,
                                                                   const unsigned long long *__restrict__ decisions,
                                                                   const Item *__restrict__ items, const int32_t *__restrict__ n_items,
                                                                   const int32_t *__restrict__ bucket_path, const int32_t *__restrict__ bucket_lo,
                                                                   const int32_t *__restrict__ path_states, int path_stride, int n_groups,
                                                                   REL *__restrict__ states, int32_t *__restrict__ alive_part,
                                                                   unsigned long long *__restrict__ acts_out,
                                                                   float *__restrict__ reward_out, int32_t *__restrict__ visited) {
    rollout_items_body<A, REL, L>(trans, C, S, B, T_cap, policy_tab, tab_stride, vec4, seed, sp, lane0, lane_ids, decisions, items, n_items,
                                  bucket_path, bucket_lo, path_states, path_stride, n_groups, states, alive_part, acts_out, reward_out, visited);
}

// states (relative, bucket-ordered) <-> indices int32 [T1, B]: one workgroup per work item.
//   k_bucket_indices: what Episodes.indices / rnad_bucket_expand read of a compact trajectory;
//   k_bucket_pack: a recorded (or dense) bucket-ordered trajectory into the compact layout.
template <typename REL>
__global__ __launch_bounds__(kThreads) void k_bucket_indices(int T1, int64_t B, const Item *__restrict__ items, const int32_t *__restrict__ n_items,
                                                             const int32_t *__restrict__ bucket_path, const int32_t *__restrict__ bucket_lo,
                                                             const int32_t *__restrict__ path_states, int path_stride, int n_groups,
                                                             const REL *__restrict__ states, int32_t *__restrict__ indices) {
    if ((int)blockIdx.x >= *n_items) return;
    const Item item = items[blockIdx.x];
    const int path_word = bucket_path[item.bucket];
    const int n_shared = (path_word & (kSharedRoot - 1)) + ((item.bucket < n_groups && (path_word & kSharedRoot)) ? 2 : 0);
    const int lo = bucket_lo[item.bucket];
    const int32_t *my_path = path_states + (int64_t)item.bucket * path_stride;
    for (int k = threadIdx.x; k < item.count; k += kThreads) {
        const int64_t j = (int64_t)item.begin + k;
        for (int t = 0; t < T1; ++t) {
            int state;
            if (t < n_shared) {
                state = my_path[t];
            } else {
                const int r = (int)states[(int64_t)t * B + j];
                state = r == 0 ? 0 : lo + r - 1;
            }
            indices[(int64_t)t * B + j] = state;
        }
    }
}

template <typename REL>
__global__ __launch_bounds__(kThreads) void k_bucket_pack(int T1, int64_t B, int rows, const Item *__restrict__ items, const int32_t *__restrict__ n_items,
                                                          const int32_t *__restrict__ bucket_path, const int32_t *__restrict__ bucket_lo,
                                                          const int32_t *__restrict__ path_states, int path_stride, int n_groups,
                                                          const int32_t *__restrict__ indices, REL *__restrict__ states,
                                                          int32_t *__restrict__ bad, int max_count) {
    if ((int)blockIdx.x >= *n_items) return;
    const Item item = items[blockIdx.x];
    if (item.count > max_count && threadIdx.x == 0) *bad = 1;  // (leaf paths: the weighted learner counts an item's columns in ONE pass of kThreads)
    const int path_word = bucket_path[item.bucket];
    const int n_shared = (path_word & (kSharedRoot - 1)) + ((item.bucket < n_groups && (path_word & kSharedRoot)) ? 2 : 0);
    const int lo = bucket_lo[item.bucket];
    const int32_t *my_path = path_states + (int64_t)item.bucket * path_stride;
    for (int k = threadIdx.x; k < item.count; k += kThreads) {
        const int64_t j = (int64_t)item.begin + k;
        for (int t = 0; t < T1; ++t) {
            const int state = indices[(int64_t)t * B + j];
            if (t < n_shared) {
                if (state != my_path[t]) *bad = 1;  // (this column is not a lane of this bucket)
            } else {
                if (state != 0 && (state < lo || state - lo >= rows)) *bad = 1;
                states[(int64_t)t * B + j] = rel_of<REL>(state, lo);
            }
        }
    }
}

// alive[t] = sum over the blocks of alive_part[block][t]; norm[P] += alive[t] for the steps of parity P: the loss normalisers N_P
// of learn/vtrace.py:373,388 (f64 sums of integers: exact in any order; zeroed by k_bucket_keys).  One workgroup per column t.
__device__ __forceinline__ void alive_column(int n_blocks, int T1, int t, const int32_t *__restrict__ alive_part, int32_t *__restrict__ alive,
                                             double *__restrict__ norm) {
    int32_t s = 0;
    for (int r = threadIdx.x; r < n_blocks; r += kThreads) s += alive_part[(int64_t)r * T1 + t];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __shared__ int32_t part[kThreads / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t x = 0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) x += part[w];
        alive[t] = x;
        if (norm && t < T1 - 1 && x != 0) atomicAdd(norm + (t & 1), (double)x);  // alive[T_cap]: after the last step, not a slot
    }
}

__global__ __launch_bounds__(kThreads) void k_bucket_alive(int n_blocks, int T1, const int32_t *__restrict__ alive_part,
                                                           int32_t *__restrict__ alive, double *__restrict__ norm) {
    alive_column(n_blocks, T1, blockIdx.x, alive_part, alive, norm);
}

// The counts k_bucket_play_learn left in kReplicas rows: alive[t], norm[P] (integers: the same bits in any order); clears the rows.
__device__ __forceinline__ void alive_from_replicas(int T1, int32_t *__restrict__ alive_rep, double *__restrict__ norm_rep,
                                                    int32_t *__restrict__ alive, double *__restrict__ norm) {
    if ((int)threadIdx.x < T1) {
        int32_t x = 0;
        for (int c = 0; c < kReplicas; ++c) x += alive_rep[c * T1 + threadIdx.x];
        alive[threadIdx.x] = x;
    }
    if (threadIdx.x >= 64 && threadIdx.x < 66) {
        double x = 0.0;
        for (int c = 0; c < kReplicas; ++c) x += norm_rep[2 * c + (threadIdx.x - 64)];
        norm[threadIdx.x - 64] = x;
    }
}
__global__ __launch_bounds__(kThreads) void k_bucket_alive_rep(int T1, int32_t *__restrict__ alive_rep, double *__restrict__ norm_rep,
                                                               int32_t *__restrict__ alive, double *__restrict__ norm) {
    alive_from_replicas(T1, alive_rep, norm_rep, alive, norm);
    __syncthreads();
    for (int i = threadIdx.x; i < kReplicas * T1; i += kThreads) alive_rep[i] = 0;
    if (threadIdx.x < 2 * kReplicas) norm_rep[threadIdx.x] = 0.0;
}

// ---------------------------------------------------------------------------------------- 4. learner
// round-to-nearest-even of a double that holds |d| < 2^51, as an integer: adding 1.5 * 2^52 leaves the rounded integer in the low
// mantissa bits (3 instructions instead of the ~20 of a generic f64 -> i64 conversion; same result as __double2ll_rn).
__device__ __forceinline__ long long round_to_ll(double d) {
    constexpr double kMagic = 6755399441055744.0;  // 1.5 * 2^52
    return __double_as_longlong(d + kMagic) - __double_as_longlong(kMagic);
}

struct FixedPoint {
    double scale_l, scale_v;  // 2^f: units per 1.0 of an addend of dL/dlogit / dL/dv
    double inv_l, inv_v;      // 2^-f: k_bucket_finish multiplies (exact: powers of two -- the bits of the division it replaces, without its ~30 fp64 instructions)
    float limit_l, limit_v;   // |addend| must stay below this (2^(62 - kLaneBits) units)
    float scale_l32, scale_v32;  // the same scales as floats (exact: powers of two)
    int check_l;                 // 2 * clip does not fit below limit_l (neurd_clip >= 2^29, e.g. "no clipping"): range-check every dL/dlogit addend
};

// One slot of the on-policy update from its row's fast record f (k_row_records): vtrace_step for the mover P ("ours") and for the
// other player ("opp"), nerd_row and the fixed-point addends, with the row-only operands read instead of recomputed.  Operation for
// operation what learn_math.hpp does per slot (products with the one-hot action and with valid == 1 drop out exactly).
template <int A, bool LOSSES, int P>
__device__ __forceinline__ void fast_slot(const float *__restrict__ f, const float *__restrict__ lg, int act, float rew, const VtHp &vh,
                                          const rnad_learn_params_t &hp, const FixedPoint &fx, Carry (&cy)[2], long long (&out)[A + 1],
                                          double (&part)[4], bool &ovf) {
    const float v = f[0], vv = f[1], e0 = f[2];
    const uint32_t bits = __float_as_uint(f[3]);
    float cs = f[4 + 2 * A], inv_mu = f[4 + 3 * A];
#pragma unroll
    for (int a = 1; a < A; ++a) {
        cs = act == a ? f[4 + 2 * A + a] : cs;
        inv_mu = act == a ? f[4 + 3 * A + a] : inv_mu;
    }
    Carry &me = cy[P], &op = cy[1 - P];  // P is a template parameter: the carries stay in registers
    const float r_me = P ? -rew : rew, r_op = P ? rew : -rew;  // player 1 is paid -r (rnad.py:368)
    // ours (vtrace.py:262-312)
    const float ru = r_me + vh.gamma * me.ru + e0;
    const float dr = r_me + vh.gamma * me.r;
    const float w = cs * me.is;
    // torch.clamp(max=) hands a NaN on (and the NaN would poison the tables through the value gradient): fminf + the flag do the same
    ovf |= w != w;
    const float vt = vv + fminf(w, vh.rho) * (ru + vh.gamma * me.nv - vv) + vh.lambda_ * fminf(w, vh.c) * vh.gamma * (me.nvt - me.nv);
    const float tail = dr + vh.gamma * me.is * me.nvt - vv;
    const float bonus = inv_mu * tail;
    float q[A], base = 0.0f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        q[a] = vv + f[4 + A + a] + (act == a ? bonus : 0.0f);
        base += f[4 + a] * q[a];
    }
    me.r = 0.0f; me.ru = 0.0f; me.nv = vv; me.nvt = vt; me.is = 1.0f;
    // opp (vtrace.py:313-319): _player_others = -1
    const float ne0 = -e0;
    const float ru_x = r_op + vh.gamma * op.ru + ne0;
    const float dr_x = r_op + vh.gamma * op.r;
    op.r = ne0 + cs * dr_x; op.ru = ru_x; op.nv = vh.gamma * op.nv; op.nvt = vh.gamma * op.nvt; op.is = cs * op.is;
    // get_loss_nerd row (vtrace.py:410-429) and its gradient
    float wv[A], wsum = 0.0f, nerd = 0.0f, mean = 0.0f;
    if (LOSSES) {
#pragma unroll
        for (int a = 0; a < A; ++a) mean += lg[a] * (float)((bits >> a) & 1);
        mean = mean / (float)A;
    }
#pragma unroll
    for (int a = 0; a < A; ++a) {
        float adv = q[a] - base;
        ovf |= adv != adv;  // (a NaN advantage reaches the gradient in the reference; here it sets the flag that poisons the tables)
#ifdef RNAD_NO_OPT_MED3
        adv = __builtin_amdgcn_fmed3f(adv, -hp.clip, hp.clip);  // torch.clamp(adv, -clip, clip) of a non-NaN
        const float fo = ((bits >> (8 + a)) & 1 ? fminf(adv, 0.0f) : 0.0f) + ((bits >> (16 + a)) & 1 ? fmaxf(adv, 0.0f) : 0.0f);
#else
        // torch.clamp(adv, -clip, clip) of a non-NaN, then apply_force_with_threshold (vtrace.py:362-366, :417):
        //   [gate_lo] min(adv_c, 0) + [gate_hi] max(adv_c, 0)  ==  median(adv, gate_lo ? -clip : 0, gate_hi ? clip : 0)
        // (one of the two terms is always a zero, so the sum is the other one exactly)
        const float lo = (bits >> (8 + a)) & 1 ? -hp.clip : 0.0f, hi = (bits >> (16 + a)) & 1 ? hp.clip : 0.0f;
        const float fo = __builtin_amdgcn_fmed3f(adv, lo, hi);
#endif
        wv[a] = fo;  // legal * f: the gates of an illegal action are closed
        wsum += wv[a];
        if (LOSSES) nerd += (float)((bits >> a) & 1) * ((lg[a] - mean) * fo);
    }
    const float share = div_by<A>(wsum);
    const float d = v - vt;
    if (LOSSES) {
        part[P] += (double)(d * d);
        part[2 + P] += -(double)nerd;
    }
    const float gv = 2.0f * d;
    ovf |= !(fabsf(gv) < fx.limit_v);
    // the scales are powers of two: the product is exact in fp32 (no f64 multiply), the rounding happens in round_to_ll
    out[A] = round_to_ll((double)(gv * fx.scale_v32));
#pragma unroll
    for (int a = 0; a < A; ++a) {
        // |g| <= 2 clip < limit_l by construction here (adv is clamped and not NaN, or the flag is set): no range check -- unless the
        // clip itself is beyond the fixed-point range (fx.check_l)
        const float g = (bits >> a) & 1 ? wv[a] - share : 0.0f;
#ifdef RNAD_NO_OPT_CHECK
        if (fx.check_l) ovf |= !(fabsf(g) < fx.limit_l);
#else
        if (LOSSES && fx.check_l) ovf |= !(fabsf(g) < fx.limit_l);  // (such updates are routed to the LOSSES instantiation)
#endif
        out[a] = round_to_ll((double)(-g * fx.scale_l32));
    }
}

// The pass of k_learn_fused<TAB> (learn.hip; learn/rnad.py:365-425) for the lanes of ONE work item -- they all went through the
// same states down to the bucket state -- with the per-slot gradients added up per (player, state) row instead of being written:
//   rows at steps t < n_path (above the bucket): one row per step for the whole workgroup -> kPathSlots LDS copies of it, lane l adds
//   into copy l & 15;
//   rows below: LDS table indexed by (state - bucket state), both players, kTabStride<A> words apart (odd: the rows of 16 lanes fall
//   into distinct banks -- with 4-word rows 61 % of the LDS cycles of this kernel were bank conflicts, profiles/r04_pmc_lds.txt).
// Addends are the UN-normalised gradients  G_l[a] = -(w - legal * sum(w) / A)  and  G_v = 2 (v - v_target)  in 64-bit fixed
// point; k_bucket_finish applies w_n / N_P and w_v / N_P (the reference scales every slot by them: vtrace.py:374,389 and
// rnad.py:424; summing first changes the rounding of the last bit only).  losses_raw[4] += sum d^2 (P = 0, 1), sum -nerd (P = 0, 1).
//
// Two kernels share the layout and the epilogue (learn_epilogue):
//   k_bucket_learn<A>            the DENSE trajectory (indices, actions, rewards, acting policy per slot; off-policy batches too)
//   k_bucket_learn_c<A, REL, L>  the COMPACT trajectory of k_bucket_rollout_items, on-policy (below)
template <int A>
constexpr int kTabStride = (A + 1) | 1;

// What both learner kernels do once the time loops are over: loss sums, overflow flag, path rows -> a replica of the upper-row table,
// shared root rows -> the table, the table -> acc (plain stores for a bucket's only item, atomics otherwise).
template <int A>
__device__ __forceinline__ void learn_epilogue(unsigned long long *__restrict__ tab, const int32_t *__restrict__ path_state, const Item &item,
                                               int s_b, int n_path, int n_shared, int kPathWords, int sub_rows, int64_t S, int n_groups,
                                               int up_stride, const int32_t *__restrict__ bucket_of, const double (&part)[4], bool ovf,
                                               double (*loss_part)[4], unsigned long long *__restrict__ acc,
                                               unsigned long long *__restrict__ rep, double *__restrict__ losses_raw,
                                               int32_t *__restrict__ overflow) {
    constexpr int PS = (A + 1) | 1, TS = kTabStride<A>;
    // losses (logging): four fp64 partial sums per block
    if (losses_raw) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            double x = part[u];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if ((threadIdx.x & 63) == 0) loss_part[threadIdx.x >> 6][u] = x;
        }
    }
    if (ovf) *overflow = 1;
    __syncthreads();
    if (losses_raw && threadIdx.x < 4) {
        double x = 0.0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) x += loss_part[w][threadIdx.x];
        if (x != 0.0) atomicAdd(losses_raw + threadIdx.x, x);
    }
    // path rows -> one of the replicas of the upper-row table (spreads the same-address atomics of the root rows)
    for (int e = threadIdx.x; e < n_path * (A + 1); e += kThreads) {
        const int t = e / (A + 1), a = e % (A + 1);
        unsigned long long x = 0ull;
#pragma unroll
        for (int c = 0; c < kPathSlots; ++c) x += tab[(t * kPathSlots + c) * PS + a];
        if (x != 0ull) {
            const int64_t slot = bucket_of[path_state[t]] - n_groups;  // path states are upper states
            // (row-major: the kReplicas copies of a row are contiguous, which is how k_bucket_finish reads them -- a wave per row)
            atomicAdd(rep + ((((int64_t)(t & 1) * up_stride + slot) * kReplicas) + (blockIdx.x & (kReplicas - 1))) * (A + 1) + a, x);
        }
    }
    // rows of the bucket's own subtree: this workgroup owns them unless the bucket was split over several items
    if (item.bucket >= n_groups) return;  // terminal bucket: every step was a path step
    for (int e = threadIdx.x; e < (n_shared - n_path) * (A + 1); e += kThreads) {  // the shared root's two rows join the table
        const int t = n_path + e / (A + 1), a = e % (A + 1);
        unsigned long long x = 0ull;
#pragma unroll
        for (int c = 0; c < kPathSlots; ++c) x += tab[(t * kPathSlots + c) * PS + a];
        tab[kPathWords + ((t & 1) * sub_rows + (path_state[t] - s_b)) * TS + a] += x;
    }
    __syncthreads();
    const int end = (int)(S - s_b < sub_rows ? S - s_b : sub_rows);
#pragma unroll
    for (int P = 0; P < 2; ++P) {
        unsigned long long *dst0 = acc + ((int64_t)P * S + s_b) * (A + 1);
        for (int r = threadIdx.x; r < end * (A + 1); r += kThreads) {
            const int row = r / (A + 1), a = r % (A + 1);  // (constant divisor)
            const unsigned long long x = tab[kPathWords + (P * sub_rows + row) * TS + a];
            if (x != 0ull && !(RNAD_ABLATE & 32)) {  // (32: timing experiment without the flush of the table)
                if (item.single) dst0[r] = x;
                else atomicAdd(dst0 + r, x);
            }
        }
    }
}

template <int A>
__global__ __launch_bounds__(kThreads) void k_bucket_learn(int T, int64_t B, int64_t S, int sub_rows, int path_words, int n_groups, int up_stride,
                                                           const Item *__restrict__ items, const int32_t *__restrict__ n_items,
                                                           const int32_t *__restrict__ bucket_of, const int32_t *__restrict__ bucket_lo,
                                                           const int32_t *__restrict__ bucket_path, const int32_t *__restrict__ indices,
                                                           const int32_t *__restrict__ actions, const float *__restrict__ rewards,
                                                           const float *__restrict__ mu_, const float *__restrict__ rec_,
                                                           rnad_learn_params_t hp, FixedPoint fx, unsigned long long *__restrict__ acc,
                                                           unsigned long long *__restrict__ rep, double *__restrict__ losses_raw,
                                                           int32_t *__restrict__ overflow) {
    // [max_path path rows][kPathSlots copies][(A + 1) | 1]  |  [sub_rows rows of player 0 | sub_rows rows of player 1][kTabStride<A>]
    extern __shared__ unsigned long long tab[];
    constexpr int PS = (A + 1) | 1, TS = kTabStride<A>;
    const int kPathWords = path_words;  // u64 words of the path region: max_path rows of the cut x kPathSlots copies
    __shared__ int32_t path_state[kMaxPath];
    __shared__ double loss_part[kThreads / 64][4];
    if ((int)blockIdx.x >= *n_items) return;
    const Item item = items[blockIdx.x];
    const int s_b = bucket_lo[item.bucket];        // first state id of the group (terminal buckets: the upper state; no rows below)
    const int n_path = bucket_path[item.bucket] & (kSharedRoot - 1);  // env steps above the group: upper states shared by every lane of the bucket
    // ... and, when the group is a single subtree, the two steps at its root: all lanes of the item add into the same two rows
    // (64-way same-address LDS atomics otherwise), so those take copies in the path region too and are folded into the table at the end
    const int n_shared = n_path + ((bucket_path[item.bucket] & kSharedRoot) ? 2 : 0);
    constexpr int RS = kRowStride<A>;
    const int n_tab = item.bucket < n_groups ? kPathWords + 2 * sub_rows * TS : n_path * kPathSlots * PS;
    for (int i = threadIdx.x; i < n_tab; i += kThreads) tab[i] = 0ull;
    __syncthreads();
    const VtHp vh{-hp.eta, hp.lambda_, hp.c, hp.rho, hp.gamma};
    double part[4] = {0.0, 0.0, 0.0, 0.0};
    bool ovf = false;
    for (int base = 0; base < item.count; base += kThreads) {
        const bool active = base + (int)threadIdx.x < item.count;
        const int64_t j = (int64_t)item.begin + base + threadIdx.x;
        Carry cy[2];
        // Software pipeline over the time loop: the states of step t - 2 and the slot's inputs of step t - 1 (action, acting policy,
        // reward, the row record -- whose address needs that step's state) are requested before the arithmetic of step t, so the
        // two dependent memory latencies of a step (state -> record) overlap the V-trace / NeuRD arithmetic of its successors.
        constexpr int kFetch = kRowLearn<A>;  // floats of a record this kernel reads
        struct Slot {
            int act;
            float rew;
            float mu[A];
            float rec[kFetch];
        };
        auto fetch = [&](int t, int state, Slot &o) {
            if (state == 0) return;
            const int64_t i = (int64_t)t * B + j;
            const int64_t row = (int64_t)(t & 1) * S + state;
            const float4 *rp = reinterpret_cast<const float4 *>(rec_ + row * RS);
#pragma unroll
            for (int u = 0; u < kFetch / 4; ++u) {
                const float4 r4 = rp[u];
                o.rec[4 * u] = r4.x; o.rec[4 * u + 1] = r4.y; o.rec[4 * u + 2] = r4.z; o.rec[4 * u + 3] = r4.w;
            }
            o.act = actions[i];
#pragma unroll
            for (int a = 0; a < A; ++a) o.mu[a] = mu_[i * A + a];
            o.rew = (t & 1) ? rewards[i] : 0.0f;  // row turns carry torch.zeros (episode.py:101)
        };
        int s_next = active ? indices[(int64_t)(T - 1) * B + j] : 0;
        int s_next2 = (active && T >= 2) ? indices[(int64_t)(T - 2) * B + j] : 0;
        Slot nxt;
        fetch(T - 1, s_next, nxt);
        for (int t = T - 1; t >= 0; --t) {
            const int state = s_next;
            const Slot cur = nxt;
            s_next = s_next2;
            if (t >= 1) fetch(t - 1, s_next, nxt);
            s_next2 = (active && t >= 2) ? indices[(int64_t)(t - 2) * B + j] : 0;
            const bool valid = state != 0;  // rnad.py:369
            const int P = t & 1;            // turns[t, :] (episode.py:96-98)
            long long q[A + 1];
#pragma unroll
            for (int a = 0; a <= A; ++a) q[a] = 0;
            if (valid) {
                const int act = cur.act;
                const float *rec = cur.rec;  // this row's record (k_row_records)
                const uint32_t bits = __float_as_uint(rec[3 * A + 2]);
                float mu[A], lg[A], pip[A], lpol[A], legal[A], oh[A];
#pragma unroll
                for (int a = 0; a < A; ++a) {
                    mu[a] = cur.mu[a];
                    lg[a] = rec[a];
                    pip[a] = rec[A + 2 + a];    // process_policy(pi) of the learner (rnad.py:374)
                    lpol[a] = rec[2 * A + 2 + a];  // log_policy_reg (rnad.py:382)
                    legal[a] = (float)((bits >> a) & 1);
                    oh[a] = act == a ? 1.0f : 0.0f;
                }
                const float rew = cur.rew;
                const float vtn = rec[A + 1];
                float vt[2], qv[2][A];
                vtrace_step<A>(cy[0], vh, true, P == 0, 1.0f, vtn, rew, mu, pip, lpol, oh, vt[0], qv[0]);   // player 0 (rnad.py:384-406)
                vtrace_step<A>(cy[1], vh, true, P == 1, 1.0f, vtn, -rew, mu, pip, lpol, oh, vt[1], qv[1]);  // player 1: rewards = -r (:368)
                const float d = rec[A] - (P ? vt[1] : vt[0]);
                float qp[A], g[A];
#pragma unroll
                for (int a = 0; a < A; ++a) qp[a] = P ? qv[1][a] : qv[0][a];
                const float nerd = nerd_row<A>(lg, pip, qp, legal, hp.clip, hp.threshold, g);
                part[P] += (double)(d * d);
                part[2 + P] += -(double)nerd;
                const float gv = 2.0f * d;
                ovf |= !(fabsf(gv) < fx.limit_v);
                q[A] = round_to_ll((double)gv * fx.scale_v);
#pragma unroll
                for (int a = 0; a < A; ++a) {
                    ovf |= !(fabsf(g[a]) < fx.limit_l);
                    q[a] = round_to_ll((double)(-g[a]) * fx.scale_l);
                }
            } else {
                cy[0] = Carry{};  // reset_carry (vtrace.py:320)
                cy[1] = Carry{};
            }
            if (t < n_shared) {  // a step above the bucket state (or at a shared root): one row for the whole workgroup, kPathSlots copies of it in LDS
                if (threadIdx.x == 0 && base == 0) path_state[t] = state;
                if (valid) {
                    unsigned long long *dst = tab + (t * kPathSlots + (threadIdx.x & (kPathSlots - 1))) * PS;
#pragma unroll
                    for (int a = 0; a <= A; ++a) atomicAdd(dst + a, (unsigned long long)q[a]);
                }
            } else if (valid) {
                unsigned long long *dst = tab + kPathWords + (P * sub_rows + (state - s_b)) * TS;
#pragma unroll
                for (int a = 0; a <= A; ++a) atomicAdd(dst + a, (unsigned long long)q[a]);
            }
        }
    }
    learn_epilogue<A>(tab, path_state, item, s_b, n_path, n_shared, kPathWords, sub_rows, S, n_groups, up_stride, bucket_of, part, ovf,
                      loss_part, acc, rep, losses_raw, overflow);
}

// The on-policy learner of the default step: the compact trajectory of k_bucket_rollout_items -- relative states below the cut, the
// lane's packed actions and its one reward -- played with the pi columns of the records as the actor (mu == pi, the very floats the
// rollout sampled from).  rec_ points at the FAST records (k_row_records): whatever of vtrace_step / nerd_row has row-only operands
// arrives precomputed, and a slot is left with the carries, the value target, q, the advantage and one division -- the same operations
// on the same operands as the dense kernel, hence the same sums, bit for bit (test_compact_trajectory_is_the_dense_one).
// LOSSES: loss_v / loss_nerd sums for a logging step, which need the logits of the dense record (logit_).
//
// The time loop runs backwards in two phases (r04; 70 -> see DESIGN.md section 5.1):
//   1. t >= n_shared: every lane is in a state of its own -- one byte of `states`, a 64-byte record gathered through the vector
//      memory path, software-pipelined (the state of step t - 2 and the record of step t - 1 in flight during the arithmetic of step t),
//      ds_add_u64 into the row of the LDS table.  4 of the 12 steps on configs[1].
//   2. t < n_shared: the whole workgroup is in path_states[bucket][t].  Nothing is read per lane; the record arrives through the SCALAR
//      cache into SGPRs (one s_load_dwordx16 per step and wave instead of 4 x 64 lanes of vector loads: the L1's tag rate, 64 % busy in
//      the r03 kernel, is out of the picture), the gate bits are scalar, and only the carries, the action selects and the addends are
//      vector work.  Every active lane is valid here (it reached the group).
// All offsets are 32-bit off scalar bases (tables < 4 GB: the 64-bit address arithmetic was ~10 % of the r03 kernel's VALU work).
#ifndef RNAD_PHASE2_VMEM
#define RNAD_PHASE2_VMEM 0
#endif
#ifdef RNAD_LEARN_WAVES
#define RNAD_LEARN_ATTR __attribute__((amdgpu_waves_per_eu(RNAD_LEARN_WAVES, RNAD_LEARN_WAVES)))
#else
#define RNAD_LEARN_ATTR
#endif
template <int A, typename REL, bool LOSSES, bool DISTINCT = false, bool WEIGHTED = false>
__device__ __forceinline__ void learn_c_body(int T, int64_t B, int64_t S, int sub_rows, int path_words, int n_groups, int up_stride,
                                                             const Item *__restrict__ items, const int32_t *__restrict__ n_items,
                                                             const int32_t *__restrict__ bucket_of, const int32_t *__restrict__ bucket_lo,
                                                             const int32_t *__restrict__ bucket_path, const int32_t *__restrict__ path_states,
                                                             int path_stride, const REL *__restrict__ states, const float *__restrict__ rec_,
                                                             const unsigned long long *__restrict__ acts_,
                                                             const float *__restrict__ reward_, const float *__restrict__ logit_,
                                                             rnad_learn_params_t hp, FixedPoint fx, unsigned long long *__restrict__ acc,
                                                             unsigned long long *__restrict__ rep, double *__restrict__ losses_raw,
                                                             int32_t *__restrict__ overflow, const int32_t *__restrict__ alive_part,
                                                             int alive_blocks, int T1, int32_t *__restrict__ alive,
                                                             double *__restrict__ norm_out, const Hand *hand = nullptr,
                                                             const Distinct distinct = Distinct{}, const int32_t *__restrict__ leaf_col = nullptr,
                                                             const int32_t *__restrict__ lane_start = nullptr,
                                                             const int32_t *__restrict__ lane_total = nullptr, const LeafCount lc = LeafCount{}) {
    extern __shared__ unsigned long long tab[];
    constexpr int PS = (A + 1) | 1, TS = kTabStride<A>, FS = kFastStride<A>, RS = kRowStride<A>;
    const int kPathWords = path_words;
    __shared__ int32_t path_state[kMaxPath];
    __shared__ double loss_part[kThreads / 64][4];
    // The rollout left its per-workgroup alive counts un-summed (rnad_rollout_bucketed_compact with alive == NULL): workgroup t < T1
    // adds up column t first -- k_bucket_alive's work without its launch; the normalisers are read by k_bucket_finish, after this kernel.
    if (alive_part && (int)blockIdx.x < T1) alive_column(alive_blocks, T1, blockIdx.x, alive_part, alive, norm_out);
    const int my_item = xcd_item(*n_items);
    if (my_item < 0) return;
    const Item item = items[my_item];
    const int s_b = bucket_lo[item.bucket];
    const int path_word = bucket_path[item.bucket];
    const int n_path = path_word & (kSharedRoot - 1);
    const int n_shared = n_path + ((item.bucket < n_groups && (path_word & kSharedRoot)) ? 2 : 0);
    const int32_t *my_path = path_states + (uint32_t)item.bucket * (uint32_t)path_stride;
    const int n_tab = item.bucket < n_groups ? kPathWords + 2 * sub_rows * TS : n_path * kPathSlots * PS;
    for (int i = threadIdx.x; i < n_tab; i += kThreads) tab[i] = 0ull;
    if ((int)threadIdx.x < n_shared) path_state[threadIdx.x] = my_path[threadIdx.x];
    __syncthreads();
    const VtHp vh{-hp.eta, hp.lambda_, hp.c, hp.rho, hp.gamma};
    const uint32_t B32 = (uint32_t)B, S32 = (uint32_t)S;
    const int t_low = min(T, n_shared);  // phase 2 covers [0, t_low), phase 1 [t_low, T)
    double part[4] = {0.0, 0.0, 0.0, 0.0};
    bool ovf = false;
    // distinct: one pass per kThreads listed trajectories, each weighted with the number of lanes that took it
    // WEIGHTED (leaf paths, rnad_leaf_paths_t): column j is a trajectory of the TREE and its weight the number of lanes of the batch that
    // played it; a column nobody played is skipped.  64-bit integer sums: weight x addend is what the per-lane learner adds up over those
    // lanes, bit for bit.  The weights of this item's columns: the lanes of the item's BUCKET are lane_total[bucket] consecutive entries of
    // leaf_col (bucket order) from lane_start[bucket] on -- every workgroup of the bucket reads them all and counts, in LDS, those that
    // fall into its own columns (<= kThreads of them).
    constexpr bool by_trajectory = DISTINCT || WEIGHTED;
    __shared__ int32_t w_lds[WEIGHTED ? kThreads : 1];
    if (WEIGHTED) {
        if (leaf_crowded(lc, item.bucket)) {  // (the rollout's work items counted this bucket's lanes per column: LeafCount)
            int32_t w = 0;
            if ((int)threadIdx.x < item.count) {
                w = lc.col_count[item.begin + threadIdx.x];
                if (w != 0) lc.col_count[item.begin + threadIdx.x] = 0;  // zero again for the next step
            }
            w_lds[threadIdx.x] = w;
        } else {
            w_lds[threadIdx.x] = 0;
            __syncthreads();
            const int32_t l0 = lane_start[item.bucket], ln = lane_total[item.bucket];
            for (int i = threadIdx.x; i < ln; i += kThreads) {
                const uint32_t c = (uint32_t)(leaf_col[l0 + i] - item.begin);
                if (c < (uint32_t)item.count) atomicAdd(&w_lds[c], 1);
            }
        }
        __syncthreads();
    }
    const int n_work = DISTINCT ? *distinct.n : item.count;
    for (int base = 0; base < n_work; base += kThreads) {
        bool active = base + (int)threadIdx.x < n_work;
        const uint32_t j = DISTINCT ? (uint32_t)(active ? distinct.column[base + threadIdx.x] : item.begin) : (uint32_t)(item.begin + base) + threadIdx.x;
        long long weight = DISTINCT ? (long long)(active ? distinct.count[distinct.key[base + threadIdx.x]] : 0) : 1ll;
        if (WEIGHTED) {
            const int32_t w = active ? w_lds[threadIdx.x] : 0;  // (one pass: a leaf-path item holds <= kThreads columns)
            weight = w;
            active = w != 0;
            if (__ballot(active) == 0ull) continue;  // (a wave whose columns are all empty: nothing to add, no barrier inside the loop)
        }
        Carry cy[2];
        // (k_bucket_play_learn, an item played in one pass: this thread's own values arrive in registers)
        const bool handed = hand != nullptr && item.count <= kThreads;
        const unsigned long long acts = handed ? hand->acts : active ? acts_[j] : 0ull;
        const float reward_final = handed ? hand->reward : active ? reward_[j] : 0.0f;
        const int last_pair = (T - 1) >> 1;
        // the state the last step of the window led to (rewards *= (indices == 0), episode.py:120-121: the step into state 0 pays)
        auto state_at = [&](int t) {
            if (handed) {
                if (t == T) return hand->final_rel;
                const int k = last_pair - (t >> 1);
                if (k < kHandPairs) return (int)((hand->relq >> (16 * k)) & 0xffffull);
            }
            return (int)*at_bytes<REL>(states, ((uint32_t)t * B32 + j) * (uint32_t)sizeof(REL));
        };
        bool after_zero = active && (T < n_shared ? false : state_at(T) == 0);
        // ------------------------------------------------------------------ phase 1: per-lane states below the cut
        if (t_low < T && !(RNAD_ABLATE & 8)) {
            struct Slot {
                float rec[FS];
                float lg[A];  // LOSSES: the row's logits
            };
            auto fetch = [&](int t, int rel, Slot &o) {
                if (rel == 0) return;
                const uint32_t row = (uint32_t)(t & 1) * S32 + (uint32_t)(s_b + rel - 1);
                const float4 *rp = at_bytes<float4>(rec_, row * (uint32_t)(FS * sizeof(float)));
#pragma unroll
                for (int u = 0; u < FS / 4; ++u) {
                    const float4 r4 = rp[u];
                    o.rec[4 * u] = r4.x; o.rec[4 * u + 1] = r4.y; o.rec[4 * u + 2] = r4.z; o.rec[4 * u + 3] = r4.w;
                }
                if (LOSSES) {
                    const float *lp = at_bytes<float>(logit_, row * (uint32_t)(RS * sizeof(float)));
#pragma unroll
                    for (int a = 0; a < A; ++a) o.lg[a] = lp[a];
                }
            };
            int r_next = active ? state_at(T - 1) : 0;
            int r_next2 = (active && T - 2 >= t_low) ? state_at(T - 2) : 0;
            Slot nxt;
            fetch(T - 1, r_next, nxt);
            for (int t = T - 1; t >= t_low; --t) {
                const int rel = r_next;
                const Slot cur = nxt;
                r_next = r_next2;
                if (t - 1 >= t_low) fetch(t - 1, r_next, nxt);
                r_next2 = (active && t - 2 >= t_low) ? state_at(t - 2) : 0;
                const bool valid = rel != 0;  // rnad.py:369
                long long q[A + 1];
#pragma unroll
                for (int a = 0; a <= A; ++a) q[a] = 0;
                if (valid) {
                    const int act = (int)(acts >> (3 * t)) & 7;
                    if (t & 1)  // uniform: the mover is a template parameter of the slot arithmetic
                        fast_slot<A, LOSSES, 1>(cur.rec, cur.lg, act, after_zero ? reward_final : 0.0f, vh, hp, fx, cy, q, part, ovf);
                    else
                        fast_slot<A, LOSSES, 0>(cur.rec, cur.lg, act, 0.0f, vh, hp, fx, cy, q, part, ovf);  // row turns: torch.zeros (episode.py:101)
#if RNAD_ABLATE & 1
                    ovf |= (q[0] ^ q[1] ^ q[A]) == 0x123456789ll;
#else
                    unsigned long long *dst = tab + kPathWords + ((t & 1) * sub_rows + (rel - 1)) * TS;
#pragma unroll
                    for (int a = 0; a <= A; ++a) atomicAdd(dst + a, (unsigned long long)(by_trajectory ? q[a] * weight : q[a]));
#endif
                } else {
                    cy[0] = Carry{};  // reset_carry (vtrace.py:320)
                    cy[1] = Carry{};
                }
                after_zero = !valid;
            }
        }
        // ------------------------------------------------------------------ phase 2: the bucket's own states, records through the scalar cache
        // (software-pipelined like phase 1: the path state of step t - 2 and the record of step t - 1 are requested before the
        // arithmetic of step t; a scalar load that is waited for on the spot costs its whole latency twice per step)
        if (active && t_low > 0 && !(RNAD_ABLATE & 16)) {
            auto row_of = [&](int t, int state) { return (uint32_t)(t & 1) * S32 + (uint32_t)state; };
            auto load_rec = [&](uint32_t row, float (&f)[FS], float (&lg)[A]) {
#if RNAD_PHASE2_VMEM  // experiment: the same record through the vector path (every lane the same address)
                asm volatile("" : "+v"(row));
                const float4 *r4p = at_bytes<float4>(rec_, row * (uint32_t)(FS * sizeof(float)));
#pragma unroll
                for (int u = 0; u < FS / 4; ++u) {
                    const float4 r4 = r4p[u];
                    f[4 * u] = r4.x; f[4 * u + 1] = r4.y; f[4 * u + 2] = r4.z; f[4 * u + 3] = r4.w;
                }
#else
                const float *rp = at_bytes<float>(rec_, row * (uint32_t)(FS * sizeof(float)));
#pragma unroll
                for (int u = 0; u < FS; ++u) f[u] = rp[u];
#endif
                if (LOSSES) {
                    const float *lp = at_bytes<float>(logit_, row * (uint32_t)(RS * sizeof(float)));
#pragma unroll
                    for (int a = 0; a < A; ++a) lg[a] = lp[a];
                }
            };
            auto slot = [&](int t, const float (&f)[FS], const float (&lg)[A]) {
                const int act = (int)(acts >> (3 * t)) & 7;
                long long q[A + 1];
                if (t & 1)
                    fast_slot<A, LOSSES, 1>(f, lg, act, after_zero ? reward_final : 0.0f, vh, hp, fx, cy, q, part, ovf);
                else
                    fast_slot<A, LOSSES, 0>(f, lg, act, 0.0f, vh, hp, fx, cy, q, part, ovf);
                after_zero = false;
#if RNAD_ABLATE & 1
                ovf |= (q[0] ^ q[1] ^ q[A]) == 0x123456789ll;
#else
                unsigned long long *dst = tab + (t * kPathSlots + (threadIdx.x & (kPathSlots - 1))) * PS;
#pragma unroll
                for (int a = 0; a <= A; ++a) atomicAdd(dst + a, (unsigned long long)(by_trajectory ? q[a] * weight : q[a]));
#endif
            };
            // two record buffers that swap roles (no copies): the time loop is unrolled by two, and t_low is even whenever it is a path
            // length (n_shared); an odd window (T < n_shared, T odd) plays its first step on its own
            float fa[FS], la[A], fb[FS], lb[A];
            int t = t_low - 1;
            load_rec(row_of(t, __builtin_amdgcn_readfirstlane(my_path[t])), fa, la);
            if (!(t & 1)) {  // (t even: a lone row step first)
                slot(t, fa, la);
                --t;
                if (t >= 0) load_rec(row_of(t, __builtin_amdgcn_readfirstlane(my_path[t])), fa, la);
            }
            for (; t >= 1; t -= 2) {  // t odd: the column step in fa, then the row step of the same state in fb
                load_rec(row_of(t - 1, __builtin_amdgcn_readfirstlane(my_path[t - 1])), fb, lb);
                slot(t, fa, la);
                if (t >= 2) load_rec(row_of(t - 2, __builtin_amdgcn_readfirstlane(my_path[t - 2])), fa, la);
                slot(t - 1, fb, lb);
            }
        }
    }
    learn_epilogue<A>(tab, path_state, item, s_b, n_path, n_shared, kPathWords, sub_rows, S, n_groups, up_stride, bucket_of, part, ovf,
                      loss_part, acc, rep, losses_raw, overflow);
}

template <int A, typename REL, bool LOSSES, bool WEIGHTED = false>
__global__ __launch_bounds__(kThreads) RNAD_LEARN_ATTR void k_bucket_learn_c(int T, int64_t B, int64_t S, int sub_rows, int path_words, int n_groups, int up_stride,
                                                             const Item *__restrict__ items, const int32_t *__restrict__ n_items,
                                                             const int32_t *__restrict__ bucket_of, const int32_t *__restrict__ bucket_lo,
                                                             const int32_t *__restrict__ bucket_path, const int32_t *__restrict__ path_states,
                                                             int path_stride, const REL *__restrict__ states, const float *__restrict__ rec_,
                                                             const unsigned long long *__restrict__ acts_,
                                                             const float *__restrict__ reward_, const float *__restrict__ logit_,
                                                             rnad_learn_params_t hp, FixedPoint fx, unsigned long long *__restrict__ acc,
                                                             unsigned long long *__restrict__ rep, double *__restrict__ losses_raw,
                                                             int32_t *__restrict__ overflow, const int32_t *__restrict__ alive_part,
                                                             int alive_blocks, int T1, int32_t *__restrict__ alive,
                                                             double *__restrict__ norm_out, const int32_t *__restrict__ leaf_col,
                                                             const int32_t *__restrict__ lane_start, const int32_t *__restrict__ lane_total,
                                                             LeafCount lc) {
    learn_c_body<A, REL, LOSSES, false, WEIGHTED>(T, B, S, sub_rows, path_words, n_groups, up_stride, items, n_items, bucket_of, bucket_lo, bucket_path,
                                                  path_states, path_stride, states, rec_, acts_, reward_, logit_, hp, fx, acc, rep, losses_raw, overflow,
                                                  alive_part, alive_blocks, T1, alive, norm_out, nullptr, Distinct{}, leaf_col, lane_start, lane_total, lc);
}

// The rollout half of the leaf-path step (rnad_leaf_paths_t): k_bucket_rollout_items with the alive counts and normalisers left in the
// replica rows (as k_bucket_play_learn leaves them: k_bucket_finish adds them up) and, per lane, the column of the transition it leaves the
// tree by (leaf_col, bucket order: the scratch words that held the sort keys).
template <int A, typename REL>
__global__ __launch_bounds__(kThreads) void k_bucket_play_count(const Trans *__restrict__ trans, int C, int64_t S, int64_t B, int T_cap,
                                                                const float *__restrict__ policy_tab, int64_t tab_stride, int vec4, uint64_t seed,
                                                                const rnad_step_params_t *__restrict__ sp, int64_t lane0,
                                                                const int32_t *__restrict__ lane_ids,
                                                                const unsigned long long *__restrict__ decisions, const Item *__restrict__ items,
                                                                const int32_t *__restrict__ n_items, const int32_t *__restrict__ bucket_path,
                                                                const int32_t *__restrict__ bucket_lo, const int32_t *__restrict__ path_states,
                                                                int path_stride, int n_groups, REL *__restrict__ states,
                                                                int32_t *__restrict__ alive_rep, double *__restrict__ norm_rep,
                                                                unsigned long long *__restrict__ acts_out, float *__restrict__ reward_out,
                                                                const int32_t *__restrict__ col_of, int32_t *__restrict__ leaf_col, LeafCount lc) {
    rollout_items_body<A, REL, 1, false, true>(trans, C, S, B, T_cap, policy_tab, tab_stride, vec4, seed, sp, lane0, lane_ids, decisions, items, n_items,
                                  bucket_path, bucket_lo, path_states, path_stride, n_groups, states, alive_rep, acts_out, reward_out, nullptr,
                                  norm_rep, nullptr, Distinct{}, col_of, leaf_col, lc);
}

// Rollout and learner of a work item in ONE launch (r04): the workgroup that played the item's lanes runs their update right away --
// thread k reads back what it played itself (packed actions, reward, states of lane item.begin + k: its own stores, program order, no
// fence; RNAD_HANDOVER=1 keeps them in registers instead -- struct Hand --, which costs the sixth wave per SIMD and more than it saves),
// so the trajectory is the one k_bucket_rollout_items leaves and the sums are the ones k_bucket_learn_c adds up, bit for bit.  What it buys is
// a launch floor and the overlap of the rollout's gather latency with the learner's arithmetic across the workgroups of a CU
// (tools/micro/overlap_probe.py: the two kernels side by side on two streams take 80 us where back to back they take 95).
// The alive counts and the normalisers go, with atomics, into one of kReplicas rows each (alive_rep, norm_rep: behind the learner's
// accumulators, zero between updates); k_bucket_finish -- or k_bucket_alive_rep, when the finish is the caller's -- adds the rows up,
// hands alive[] / norm[] out and clears them.
#ifndef RNAD_HANDOVER
#define RNAD_HANDOVER 0
#endif
// (build switch: waves per SIMD the compiler must leave room for.  On its own it takes 77 VGPRs (DISTINCT: 79) -- six waves; with
// RNAD_HANDOVER it takes 85, runs at five and loses 7 %: 73.8 -> 79.1 us)
#ifndef RNAD_PLAY_LEARN_WAVES
#define RNAD_PLAY_LEARN_WAVES 0
#endif
#if RNAD_PLAY_LEARN_WAVES > 0
#define RNAD_PLAY_LEARN_ATTR __attribute__((amdgpu_waves_per_eu(RNAD_PLAY_LEARN_WAVES, RNAD_PLAY_LEARN_WAVES)))
#else
#define RNAD_PLAY_LEARN_ATTR
#endif
template <int A, typename REL, bool DISTINCT>
__global__ __launch_bounds__(kThreads) RNAD_PLAY_LEARN_ATTR void k_bucket_play_learn(
    const Trans *__restrict__ trans, int C, int64_t S, int64_t B, int T_cap, const float *__restrict__ policy_tab, int64_t tab_stride, int vec4,
    uint64_t seed, const rnad_step_params_t *__restrict__ sp, int64_t lane0, const int32_t *__restrict__ lane_ids,
    const unsigned long long *__restrict__ decisions, const Item *__restrict__ items, const int32_t *__restrict__ n_items,
    const int32_t *__restrict__ bucket_path, const int32_t *__restrict__ bucket_lo, const int32_t *__restrict__ path_states, int path_stride,
    int n_groups, REL *states, int32_t *__restrict__ alive_rep, double *__restrict__ norm_rep, unsigned long long *acts, float *reward,
    int sub_rows, int path_words, int up_stride, const int32_t *__restrict__ bucket_of, const float *__restrict__ rec_, rnad_learn_params_t hp,
    FixedPoint fx, unsigned long long *__restrict__ acc, unsigned long long *__restrict__ rep, int32_t *__restrict__ overflow,
    int distinct_words, int distinct_keys, int distinct_capacity) {
    // distinct_keys > 0: the learner runs once per distinct trajectory of the item (struct Distinct); its arrays sit behind the
    // learner's table in the dynamic LDS
    extern __shared__ unsigned long long lds_all[];
    Distinct dd;
    dd.count = reinterpret_cast<int32_t *>(lds_all + distinct_words);  // [distinct_keys] | a counter that stays 1 | the list length
    dd.n = dd.count + distinct_keys + 1;
    dd.key = dd.n + 1;
    dd.column = dd.key + distinct_capacity;
    dd.codes = A * A * C + 1;
    dd.keys = distinct_keys;
    if (DISTINCT) {
        for (int i = threadIdx.x; i < distinct_keys + 2; i += kThreads) dd.count[i] = i == distinct_keys ? 1 : 0;
        __syncthreads();
    }
    Hand hand{0ull, 0ull, 0.0f, 0};
    rollout_items_body<A, REL, 1, DISTINCT>(trans, C, S, B, T_cap, policy_tab, tab_stride, vec4, seed, sp, lane0, lane_ids, decisions, items, n_items,
                                  bucket_path, bucket_lo, path_states, path_stride, n_groups, states, alive_rep, acts, reward, nullptr, norm_rep,
                                  RNAD_HANDOVER ? &hand : nullptr, dd);
    learn_c_body<A, REL, false, DISTINCT>(T_cap, B, S, sub_rows, path_words, n_groups, up_stride, items, n_items, bucket_of, bucket_lo, bucket_path,
                                path_states, path_stride, states, rec_, acts, reward, nullptr, hp, fx, acc, rep, nullptr, overflow, nullptr, 0,
                                T_cap + 1, nullptr, nullptr, RNAD_HANDOVER ? &hand : nullptr, dd);
}

// acc -> fp32 tables, normalised: dlogit_tab[P * S + s] = w_n * (G_l / N_P), dv_tab likewise with w_v (learn/vtrace.py:374,389;
// rnad.py:424).  Clears what it read, so that the accumulators are zero again for the next update.  An addend beyond the
// fixed-point range poisons the tables with NaN instead of passing silently.
//   workgroups [0, row_blocks): one thread per (player, state) row below the cut;
//   the others: the rows above the buckets, one WAVE per row -- its kReplicas copies are read by the 64 lanes (lane c: replica c)
//   and summed on the DPP network; lane 63 converts.
// The workgroup that takes the last ticket clears the loss sums and the overflow flag: every workgroup has read them by then.
template <int A>
__global__ __launch_bounds__(kThreads) void k_bucket_finish(int64_t S, int row_blocks, const int32_t *__restrict__ row_list,
                                                            const int64_t *__restrict__ n_rows, int n_upper, int n_groups,
                                                            const int32_t *__restrict__ upper_list, const int32_t *__restrict__ bucket_of,
                                                            unsigned long long *__restrict__ acc, unsigned long long *__restrict__ rep,
                                                            const double *__restrict__ norm, float w_v, float w_n, FixedPoint fx,
                                                            int32_t *__restrict__ overflow, double *__restrict__ losses_raw,
                                                            double *__restrict__ losses, float *__restrict__ dlogit_tab,
                                                            float *__restrict__ dv_tab, int upper_blocks, int n_multi,
                                                            const int32_t *__restrict__ multi_start,
                                                            const int32_t *__restrict__ multi_order,
                                                            const int32_t *__restrict__ multi_first, int rows_below_cut,
                                                            int32_t *__restrict__ alive_rep,
                                                            double *__restrict__ norm_rep, int T1, int32_t *__restrict__ alive_out,
                                                            double *__restrict__ norm_out) {
    static_assert(kReplicas == 64, "one replica per lane");
    // norm_rep given (after k_bucket_play_learn): the normalisers are still spread over kReplicas rows -- lane c reads row c, the wave adds
    // them up (integers below 2^53: exact in any order, the bits of k_bucket_alive's sums); workgroup 0 also hands alive[] / norm[] out
    float nf0, nf1;
    if (norm_rep) {
        double n0 = norm_rep[2 * (threadIdx.x & 63)], n1 = norm_rep[2 * (threadIdx.x & 63) + 1];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            n0 += __shfl_xor(n0, off, 64);
            n1 += __shfl_xor(n1, off, 64);
        }
        // (`norm` given as well: the caller's normalisers -- those of a global batch -- divide; the rows are still handed out and cleared)
        nf0 = norm ? norm_of(norm) : norm_of(&n0);
        nf1 = norm ? norm_of(norm + 1) : norm_of(&n1);
        if (blockIdx.x == 0) alive_from_replicas(T1, alive_rep, norm_rep, alive_out, norm_out);
    } else {
        nf0 = norm_of(norm);
        nf1 = norm_of(norm + 1);
    }
    const bool bad = *overflow != 0;
    const float nan = __uint_as_float(0x7fc00000u);
    if (blockIdx.x == 0 && threadIdx.x == 0 && losses) {
        losses[0] = losses_raw[0] / (double)nf0 + losses_raw[1] / (double)nf1;
        losses[1] = losses_raw[2] / (double)nf0 + losses_raw[3] / (double)nf1;
    }
#ifndef RNAD_FINISH_ABLATE
#define RNAD_FINISH_ABLATE 0  // (timing experiments, tools/build_variant.sh: 1 = no row threads, 2 = no groups, 4 = no upper rows, 8 = no tickets)
#endif
    if ((int)blockIdx.x < row_blocks) {
        if (RNAD_FINISH_ABLATE & 1) return;
        // (few, fat workgroups: every one of them takes a ticket below)
        const int64_t limit = row_list ? *n_rows : 2 * S;
        const int64_t stride = (int64_t)row_blocks * kThreads;
        // kFinishRows rows per thread and pass, all their loads in flight together (a row at a time is a chain of memory round trips)
        for (int64_t i0 = (int64_t)blockIdx.x * kThreads + threadIdx.x; i0 < limit; i0 += stride * kFinishRows) {
            int64_t r[kFinishRows];
            bool on[kFinishRows];
            long long x[kFinishRows][A + 1];
#pragma unroll
            for (int k = 0; k < kFinishRows; ++k) {
                const int64_t i = i0 + k * stride;
                r[k] = i < limit ? (row_list ? (int64_t)row_list[i] : i) : 0;  // row list: the rows the batch visited (all others hold zero sums)
                on[k] = i < limit;
            }
#pragma unroll
            for (int k = 0; k < kFinishRows; ++k) {
                const int64_t s = r[k] >= S ? r[k] - S : r[k];
                if (!rows_below_cut) on[k] = on[k] && !(n_upper > 0 && bucket_of[s] >= n_groups);  // (rows above the cut: the wave-per-row workgroups)
            }
#pragma unroll
            for (int k = 0; k < kFinishRows; ++k)
#pragma unroll
                for (int a = 0; a <= A; ++a) x[k][a] = on[k] ? (long long)acc[r[k] * (A + 1) + a] : 0ll;
#pragma unroll
            for (int k = 0; k < kFinishRows; ++k) {
                if (!on[k]) continue;
#pragma unroll
                for (int a = 0; a <= A; ++a)
                    if (x[k][a] != 0) acc[r[k] * (A + 1) + a] = 0ull;
                const float nf = r[k] >= S ? nf1 : nf0;
#pragma unroll
                for (int a = 0; a < A; ++a) dlogit_tab[r[k] * A + a] = bad ? nan : w_n * ((float)((double)x[k][a] * fx.inv_l) / nf);
                dv_tab[r[k]] = bad ? nan : w_v * ((float)((double)x[k][A] * fx.inv_v) / nf);
            }
        }
    } else if ((int)blockIdx.x >= row_blocks + upper_blocks) {
        // Rows with the same observation (csrc/rows_dedup.hip), groups of more than one row: a wave per group converts its rows like the
        // threads above and adds them up into the table row of the group's representative -- lane l the rows l, l + 64, ... in that order,
        // then one butterfly: operation for operation k_rows_segment_sum on the tables of a finish over all rows, without the tables.
        const int g = ((int)blockIdx.x - row_blocks - upper_blocks) * (kThreads / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (RNAD_FINISH_ABLATE & 2) return;
        if (g < n_multi) {
            float sum[A + 1];
#pragma unroll
            for (int a = 0; a <= A; ++a) sum[a] = 0.0f;
            // kFinishRows rows per lane in flight, added in ascending order
            auto add_rows = [&](const int64_t (&r)[kFinishRows]) {
                long long x[kFinishRows][A + 1];
#pragma unroll
                for (int k = 0; k < kFinishRows; ++k)
#pragma unroll
                    for (int a = 0; a <= A; ++a) x[k][a] = r[k] >= 0 ? (long long)acc[r[k] * (A + 1) + a] : 0ll;
#pragma unroll
                for (int k = 0; k < kFinishRows; ++k) {
                    if (r[k] < 0) continue;
#pragma unroll
                    for (int a = 0; a <= A; ++a)
                        if (x[k][a] != 0) acc[r[k] * (A + 1) + a] = 0ull;
                    const float nf = r[k] >= S ? nf1 : nf0;
#pragma unroll
                    for (int a = 0; a < A; ++a) sum[a] += bad ? nan : w_n * ((float)((double)x[k][a] * fx.inv_l) / nf);
                    sum[A] += bad ? nan : w_v * ((float)((double)x[k][A] * fx.inv_v) / nf);
                }
            };
            int64_t rep_row;
            if (multi_first) {
                // r06: the group's first 64 * kFinishRows rows from a table padded per group (lane-major as the loop below takes them, -1
                // beyond the group): their accumulators are requested one round trip after the launch starts instead of two (group ->
                // multi_start -> multi_order -> acc was 10 of this kernel's 16 us on configs[1]'s 1 682 groups)
                int64_t r[kFinishRows];
#pragma unroll
                for (int k = 0; k < kFinishRows; ++k) r[k] = (int64_t)multi_first[((int64_t)g * kFinishRows + k) * 64 + lane];
                const int lo = multi_start[g], hi = multi_start[g + 1];  // (in flight beside them: only groups beyond the table need them)
                rep_row = __shfl(r[0], 0, 64);
                add_rows(r);
                for (int i0 = lo + 64 * kFinishRows + lane; i0 < hi; i0 += 64 * kFinishRows) {
#pragma unroll
                    for (int k = 0; k < kFinishRows; ++k) r[k] = i0 + 64 * k < hi ? (int64_t)multi_order[i0 + 64 * k] : -1;
                    add_rows(r);
                }
            } else {
                const int lo = multi_start[g], hi = multi_start[g + 1];
                for (int i0 = lo + lane; i0 < hi; i0 += 64 * kFinishRows) {
                    int64_t r[kFinishRows];
#pragma unroll
                    for (int k = 0; k < kFinishRows; ++k) r[k] = i0 + 64 * k < hi ? (int64_t)multi_order[i0 + 64 * k] : -1;
                    add_rows(r);
                }
                rep_row = multi_order[lo];
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
                for (int a = 0; a <= A; ++a) sum[a] += __shfl_xor(sum[a], off, 64);
            }
            if (lane == 0) {
                const int64_t r = rep_row;
#pragma unroll
                for (int a = 0; a < A; ++a) dlogit_tab[r * A + a] = sum[a];
                dv_tab[r] = sum[A];
            }
        }
    } else {
        const int u = ((int)blockIdx.x - row_blocks) * (kThreads / 64) + (threadIdx.x >> 6), c = threadIdx.x & 63;
        if (RNAD_FINISH_ABLATE & 4) return;
        if (u < 2 * n_upper) {
            const int P = u / n_upper, pos = u % n_upper;
            unsigned long long *src = rep + (((int64_t)P * n_upper + pos) * kReplicas + c) * (A + 1);
            const int64_t r = (int64_t)P * S + upper_list[pos];
            long long x[A + 1];
#pragma unroll
            for (int a = 0; a <= A; ++a) {
                const unsigned long long v = src[a];
                if (v != 0ull) src[a] = 0ull;
                x[a] = wave_total_in_lane63((long long)v);
            }
            if (c == 63) {
                const float nf = P ? nf1 : nf0;
#pragma unroll
                for (int a = 0; a < A; ++a) dlogit_tab[r * A + a] = bad ? nan : w_n * ((float)((double)x[a] * fx.inv_l) / nf);
                dv_tab[r] = bad ? nan : w_v * ((float)((double)x[A] * fx.inv_v) / nf);
            }
        }
    }
    __shared__ int last_s;
    if (threadIdx.x == 0) last_s = 0;
    __syncthreads();  // this workgroup's reads of the flag and the loss sums are complete
    if (RNAD_FINISH_ABLATE & 8) return;
    if (threadIdx.x == 0) {
        // (no __threadfence: a device-scope release writes the XCD's L2 back -- 22 us here -- and nothing this workgroup wrote is read
        // by the clearing one)
        // the LAST workgroup clears the flag and the loss sums.  Tickets in two levels -- kTicketGroups counters, then one: atomics on ONE
        // address retire at ~10 ns each on this part, i.e. 18 us for the 1 820 workgroups of a configs[3] launch (5 of the 8 us on
        // configs[1]) when every workgroup took the same ticket
        int32_t *ticket = overflow + 1, *sub = overflow + 2;
        const int per = ((int)gridDim.x + kTicketGroups - 1) / kTicketGroups, g = (int)blockIdx.x / per;
        const int members = min(per, (int)gridDim.x - g * per), groups = ((int)gridDim.x + per - 1) / per;
        if (atomicAdd(sub + g, 1) == members - 1) {
            sub[g] = 0;
            if (atomicAdd(ticket, 1) == groups - 1) {
                *ticket = 0;
                *overflow = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) losses_raw[u] = 0.0;
                last_s = 1;
            }
        }
    }
    if (norm_rep) {  // the rows of counts are zero again for the next update (every workgroup has read them: it took its ticket afterwards)
        __syncthreads();
        if (last_s) {
            for (int i = threadIdx.x; i < kReplicas * T1; i += kThreads) alive_rep[i] = 0;
            if (threadIdx.x < 2 * kReplicas) norm_rep[threadIdx.x] = 0.0;
        }
    }
}

FixedPoint fixed_point_for(const rnad_learn_params_t &hp) {
    // |G_l| <= 2 * clip by construction (vtrace.py:417: adv is clipped, w = legal * f, |f| <= |adv|); G_v = 2 (v - v_target) has no
    // a-priori bound: 2^11 is far beyond any payoff scale the path produces (checked per addend, NaN on overflow).
    const int budget = 62 - kLaneBits;  // bits an addend may use
    int e_l = 1;
    while (std::ldexp(1.0, e_l) <= 2.0 * std::fabs((double)hp.clip) && e_l < 30) ++e_l;
    const int e_v = 11;
    FixedPoint fx;
    fx.scale_l = std::ldexp(1.0, budget - e_l);
    fx.scale_v = std::ldexp(1.0, budget - e_v);
    fx.inv_l = std::ldexp(1.0, -(budget - e_l));
    fx.inv_v = std::ldexp(1.0, -(budget - e_v));
    fx.limit_l = (float)std::ldexp(1.0, e_l);
    fx.limit_v = (float)std::ldexp(1.0, e_v);
    fx.scale_l32 = (float)fx.scale_l;
    fx.scale_v32 = (float)fx.scale_v;
    fx.check_l = !(2.0 * std::fabs((double)hp.clip) < std::ldexp(1.0, e_l)) ? 1 : 0;  // (also true for a NaN / infinite clip)
    return fx;
}

}  // namespace

// ---------------------------------------------------------------------------------------- entry points
extern "C" int rnad_bucket_plan(const rnad_tree_t *tree, int64_t B, int64_t *out) {
    RNAD_REQUIRE(tree && out, "rnad_bucket_plan: null argument");
    Plan p;
    if (!make_plan(tree, B, p)) {
        set_error("rnad_bucket_plan: this tree / batch cannot be bucketed (ids not DFS pre-order, no level whose subtree table fits "
                  "the LDS, or more than 2^%d lanes)", kLaneBits);
        return 3;
    }
    const int64_t A1 = tree->A + 1;
    const int nb = p.cut->n_buckets, nu = p.cut->n_upper;
    out[0] = p.cut->rows;
    out[1] = nb;
    out[2] = nu;
    out[3] = p.cut->n_groups;
    out[4] = p.max_items;
    // scratch of the rollout (bytes): decisions [B] u64 | keys [B] | hist [sort_blocks][n_buckets] | totals [n_buckets] | bucket_start [n_buckets] |
    // alive_part [blocks][T_cap + 1 <= kMaxSteps + 1] | policy [2S][(A + 3) & ~3] (16-byte aligned rows)
    out[5] = 8 * B + 4 * (B + (int64_t)p.sort_blocks * nb + 2 * (int64_t)nb + std::max<int64_t>(p.max_items, (int64_t)blocks_for(B)) * (kMaxSteps + 1) +
                  2 * tree->S * ((tree->A + 3) & ~3)) + 256;
    // accumulators of the learner (bytes, must be zero before the first update): acc [2S][A+1] u64 | rep [64][2][n_upper][A+1] u64 |
    // losses_raw [4] f64 | overflow [1] i32
    // | norm_rep [64][2] f64 | alive_rep [64][kCompactSteps + 1] i32 (the counts of rnad_rollout_learn_bucketed_compact)
    out[6] = 8 * (2 * tree->S * A1 + (int64_t)kReplicas * 2 * std::max(nu, 1) * A1 + 4) + 16 + 4 * kTicketGroups +  // sums | loss sums | flag, ticket | group tickets
             8 * 2 * kReplicas + 4 * kReplicas * (kCompactSteps + 1);
    out[7] = p.lds;
    out[8] = p.rel_bytes;
    out[9] = p.tile;
    out[10] = p.chunk;
    note_sized(tree, B, p);
    return 0;
}

extern "C" int rnad_bucket_map(const rnad_tree_t *tree, int64_t B, int32_t *bucket_of, int32_t *n_groups) {
    RNAD_REQUIRE(tree && bucket_of && n_groups, "rnad_bucket_map: null argument");
    Plan p;
    RNAD_REQUIRE(make_plan(tree, B, p), "rnad_bucket_map: this tree / batch cannot be bucketed (see rnad_bucket_plan)");
    std::copy(p.cut->host_bucket_of.begin(), p.cut->host_bucket_of.end(), bucket_of);
    *n_groups = p.cut->n_groups;
    return 0;
}

extern "C" int rnad_bucket_shared_steps(const rnad_tree_t *tree, int64_t B, int32_t *n_shared) {
    RNAD_REQUIRE(tree && n_shared, "rnad_bucket_shared_steps: null argument");
    Plan p;
    RNAD_REQUIRE(make_plan(tree, B, p), "rnad_bucket_shared_steps: this tree / batch cannot be bucketed (see rnad_bucket_plan)");
    std::vector<int32_t> path((size_t)p.cut->n_buckets);
    DeviceGuard guard(tree->device);
    RNAD_HIP_OK(hipMemcpy(path.data(), p.cut->bucket_path, path.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int b = 0; b < p.cut->n_buckets; ++b)
        n_shared[b] = (path[(size_t)b] & (kSharedRoot - 1)) + ((b < p.cut->n_groups && (path[(size_t)b] & kSharedRoot)) ? 2 : 0);
    return 0;
}

namespace {
// The compact rollout runs one workgroup per work item (k_bucket_rollout_items), the dense one a workgroup per kThreads columns: either
// leaves one row of alive counts per workgroup.
int64_t alive_rows(const rnad_tree_t *, int64_t B, const Plan &p, bool compact) { return compact ? p.max_items : (int64_t)blocks_for(B); }

// REL = the type of a relative state under plan p
#define RNAD_DISPATCH_REL(p_, ...)        \
    if ((p_).rel_bytes == 1) {            \
        using REL = uint8_t;              \
        __VA_ARGS__;                      \
    } else {                              \
        using REL = uint16_t;             \
        __VA_ARGS__;                      \
    }
int64_t alive_rows_max(int64_t B, const Plan &p) { return std::max<int64_t>(p.max_items, (int64_t)blocks_for(B)); }

struct Scratch {
    unsigned long long *decisions;  // [B]
    int32_t *keys, *hist, *totals, *bucket_start, *alive_part;
    float *policy;  // [2S][A]: the actor's policy per row when the caller hands logits
};
Scratch carve_scratch(void *ws, int64_t B, const Plan &p) {
    Scratch s;
    s.decisions = (unsigned long long *)ws;
    s.keys = (int32_t *)(s.decisions + B);
    s.hist = s.keys + B;
    s.totals = s.hist + (int64_t)p.sort_blocks * p.cut->n_buckets;
    s.bucket_start = s.totals + p.cut->n_buckets;
    s.alive_part = s.bucket_start + p.cut->n_buckets;
    s.policy = (float *)(((uintptr_t)(s.alive_part + alive_rows_max(B, p) * (kMaxSteps + 1)) + 15) & ~(uintptr_t)15);  // rows of (A + 3) & ~3 floats
    return s;
}
}  // namespace

namespace {
__global__ void k_step_params_set(rnad_step_params_t *dst, uint64_t seed, float alpha, float one_minus_alpha) {
    dst->seed = seed;
    dst->alpha = alpha;
    dst->one_minus_alpha = one_minus_alpha;
}

struct QueueEntries {
    rnad_step_params_t e[RNAD_STEP_QUEUE];
};
__global__ void k_step_queue_set(rnad_step_queue_t *q, int n, QueueEntries entries) {
    const int k = threadIdx.x;
    if (k < n) q->ahead[k] = entries.e[k];
    if (k == 0) {
        q->live = entries.e[0];
        q->cursor = 0;
        q->n = n;
    }
}
}  // namespace

extern "C" int rnad_step_queue_set(rnad_step_queue_t *device_queue, int n, const rnad_step_params_t *entries, void *stream) {
    RNAD_REQUIRE(device_queue && entries && n >= 1 && n <= RNAD_STEP_QUEUE, "rnad_step_queue_set: 1..%d entries", RNAD_STEP_QUEUE);
    QueueEntries q{};
    for (int k = 0; k < n; ++k) q.e[k] = entries[k];
    hipLaunchKernelGGL(k_step_queue_set, dim3(1), dim3(64), 0, (hipStream_t)stream, device_queue, n, q);
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

// The values travel as kernel arguments (copied at launch), so the host may call this again before the GPU has consumed the
// previous step's values -- unlike an asynchronous copy from a host buffer that is about to be overwritten.
extern "C" int rnad_step_params_set(rnad_step_params_t *device_params, uint64_t seed, float alpha, float one_minus_alpha, void *stream) {
    RNAD_REQUIRE(device_params, "rnad_step_params_set: null argument");
    hipLaunchKernelGGL(k_step_params_set, dim3(1), dim3(1), 0, (hipStream_t)stream, device_params, seed, alpha, one_minus_alpha);
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int64_t rnad_bucket_record_stride(int A) { return (4 * (int64_t)A + 3 + 3) & ~(int64_t)3; }
extern "C" int64_t rnad_bucket_fast_record_stride(int A) { return 4 + 4 * (int64_t)A; }
extern "C" int64_t rnad_bucket_policy_row_stride(int A) { return ((int64_t)A + 3) & ~(int64_t)3; }

extern "C" int rnad_bucket_records(const rnad_tree_t *tree, const float *logit_tab, const float *v_tab, const float *v_target_tab,
                                   const float *logit_reg_tab, const float *logit_reg_tab_, const rnad_learn_params_t *hp,
                                   const rnad_step_params_t *device_params, float *records, float *fast_records, float *policy_rows,
                                   const int32_t *rows, const int64_t *n_rows, void *stream) {
    RNAD_REQUIRE(tree && logit_tab && v_tab && v_target_tab && logit_reg_tab && logit_reg_tab_ && hp && records,
                 "rnad_bucket_records: null argument");
    RNAD_REQUIRE(((uintptr_t)records & 15) == 0 && ((uintptr_t)fast_records & 15) == 0 && ((uintptr_t)policy_rows & 15) == 0,
                 "rnad_bucket_records: records must be 16-byte aligned");
    RNAD_REQUIRE(hp->n_disc >= 1, "rnad_bucket_records: n_disc must be positive");
    RNAD_REQUIRE(!rows == !n_rows, "rnad_bucket_records: rows and n_rows go together");
    RNAD_DISPATCH_A(tree->A, hipLaunchKernelGGL((k_row_records<kA>), dim3(blocks_for(2 * tree->S)), dim3(kThreads), 0, (hipStream_t)stream,
                                                2 * tree->S, logit_tab, v_tab, v_target_tab, logit_reg_tab, logit_reg_tab_,
                                                (const uint8_t *)tree->mask_tab, *hp, device_params, records, fast_records, policy_rows, rows, n_rows));
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

namespace {
// visited[r] = 0, except the two rows of the absorbing state: absorbed slots of the dense buffers show them (k_bucket_expand)
__global__ __launch_bounds__(kThreads) void k_clear_visited(int64_t rows, int64_t S, int32_t *__restrict__ visited) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t r = ((int64_t)blockIdx.x * 4 + u) * kThreads + threadIdx.x;
        if (r < rows) visited[r] = (r == 0 || r == S) ? 1 : 0;
    }
}

struct RolloutBuffers {  // the dense trajectory (rnad_traj_t) or the compact one
    int T_cap;
    int64_t B;
    void *indices;  // dense: int32 [T_cap + 1, B]; compact: the relative states, REL [T_cap + 1, B]
    uint8_t *mask_bits;
    float *policy;
    int32_t *actions;
    float *rewards, *values;
    int32_t *alive;
    unsigned long long *acts;  // compact: 3 bits per step
    float *final_reward;       // compact
    int32_t *visited;          // compact, optional: [2S] flags of the rows the batch went through
};

// phases: 1 = keys + sort (needs the actor's rows of the upper states), 2 = the rollout in bucket order (needs the rows of every
// non-empty group), 3 = both.  A caller that splits them (rnad_bucket_sort / rnad_bucket_play) evaluates its actor in stages:
// the upper rows, then -- once the sort has shown which groups the batch descends into (group_flags) -- the rows of those groups.
// play_rows / n_play_rows (phase 2 of a split call with a logits table): the rows that were evaluated for it.
struct CountReps {  // where k_bucket_play_learn leaves the alive counts and the normalisers (behind the tickets of the accumulators)
    double *norm_rep = nullptr;
    int32_t *alive_rep = nullptr;
    int T1 = 0;
    int32_t *alive_out = nullptr;
    double *norm_out = nullptr;
};
struct FusedLearn {  // the learner of the batch in the rollout's launch (k_bucket_play_learn): what k_bucket_learn_c needs beyond the rollout's arguments
    const float *fast;
    const rnad_learn_params_t *hp;
    void *accumulators;
    bool distinct;  // the learner once per distinct trajectory of a work item (struct Distinct)
    const rnad_leaf_paths_t *leaf = nullptr;  // the learner on the tree's leaf paths, weighted with the lanes the rollout counted (two launches)
};
// k_bucket_play_learn on the distinct trajectories of a work item (struct Distinct): larger items -- more lanes share a trajectory --
// while the counters (a key per state of the group and outcome) and the list fit the LDS beside the learner's table.
constexpr int kFusedChunkDefault = 512;      // (measured on configs[1]: 256 / 512 / 768 / 2048 lanes per item -> 0.181 / 0.171 / 0.171 / 0.177 ms per step)
constexpr int kFusedLdsBudget = 26 * 1024;  // bytes of LDS per workgroup, the learner's table included: six workgroups to a CU
int fused_distinct_keys(const rnad_tree_t *tree, const Plan &p, bool asked) {
    // opt-in (the caller's flag; RNAD_FUSED_DISTINCT=0 / 1 overrides it): the same sums bit for bit (tests/test_hip_bucket.py); what it buys
    // depends on how many lanes share a trajectory -- configs[1]: 4 % slower under the uniform policies of fresh nets, 6.5 % faster 5 000
    // updates later, 14 % after 30 000
    if (const char *e = getenv("RNAD_FUSED_DISTINCT")) asked = atoi(e) != 0;
    if (!asked) return 0;
    return p.cut->rows * (tree->A * tree->A * tree->C + 1);
}
int fused_chunk(const rnad_tree_t *tree, const Plan &p, bool asked) {
    int chunk = kFusedChunkDefault;
    if (const char *e = getenv("RNAD_FUSED_CHUNK")) chunk = std::max(64, atoi(e));
    chunk = std::max(chunk, p.chunk);
    const int keys = fused_distinct_keys(tree, p, asked);
    if (keys == 0 || (size_t)p.lds + ((size_t)keys + 2 + 2 * (size_t)chunk) * 4 > (size_t)kFusedLdsBudget) return p.chunk;  // (per lane, as played)
    return chunk;
}
CountReps count_reps(const rnad_tree_t *tree, const Plan &p, void *accumulators, int T1, int32_t *alive_out, double *norm_out) {
    const int64_t A1 = tree->A + 1;
    unsigned long long *rep = (unsigned long long *)accumulators + 2 * tree->S * A1;
    double *losses_raw = (double *)(rep + (int64_t)kReplicas * 2 * std::max(p.cut->n_upper, 1) * A1);
    int32_t *overflow = (int32_t *)(losses_raw + 4);
    CountReps c;
    c.norm_rep = (double *)(overflow + 2 + kTicketGroups);  // (8-byte aligned: kTicketGroups is even)
    c.alive_rep = (int32_t *)(c.norm_rep + 2 * kReplicas);
    c.T1 = T1;
    c.alive_out = alive_out;
    c.norm_out = norm_out;
    return c;
}

int finish_impl(const rnad_tree_t *tree, const Plan &p, const double *norm, const rnad_learn_params_t *hp, void *accumulators, double *losses,
                float *dlogit_tab, float *dv_tab, const int32_t *rows, const int64_t *n_rows, const rnad_row_groups_t *groups,
                hipStream_t stream, const CountReps *counts = nullptr);

int rollout_bucketed_impl(const rnad_tree_t *tree, const RolloutBuffers &tr, bool compact, const float *table, int64_t table_stride,
                          int table_is_policy, const float *value_table, int64_t value_stride, uint64_t seed, int64_t lane0,
                          const rnad_step_params_t *device_params, void *scratch, int32_t *lane_ids, int32_t *items, int32_t *n_items,
                          double *norm, hipStream_t stream, int phases = 3, int32_t *group_flags = nullptr,
                          const int32_t *play_rows = nullptr, const int64_t *n_play_rows = nullptr, int32_t *staged_rows = nullptr,
                          int64_t *n_staged = nullptr, bool visited_is_clear = false, void *stage_buf = nullptr, int32_t *stage_rows0 = nullptr,
                          const KeysExpand *expand = nullptr, const FusedLearn *fused = nullptr) {
    Plan p;
    RNAD_REQUIRE(make_plan(tree, tr.B, p), "rnad_rollout_bucketed: this tree / batch cannot be bucketed (see rnad_bucket_plan)");
    RNAD_REQUIRE_SIZED(tree, tr.B, p);
    const int64_t B = tr.B, S = tree->S;
    const Scratch s = carve_scratch(scratch, B, p);
    const StageOut stage = carve_stage(stage_buf, B, S);
    const int n_steps = std::min(p.cut->max_path, tr.T_cap), nb = p.cut->n_buckets;
    ProfScope prof(PROF_ACT, stream);
    const bool sort_phase = (phases & 1) != 0, play_phase = (phases & 2) != 0;
    // `visited` is cleared by a kernel (memset nodes of captured graphs are not to be trusted, see learn_bucketed_impl): by the sort's last
    // kernel when this call (or the rnad_bucket_sort before it, visited_is_clear) runs one, by a launch of its own otherwise
    if (tr.visited && play_phase && !sort_phase && !visited_is_clear)
        hipLaunchKernelGGL(k_clear_visited, dim3(blocks_for(2 * S, kThreads * 4)), dim3(kThreads), 0, stream, 2 * S, S, tr.visited);
    const float *policy_tab = table;
    int64_t policy_stride = table_stride;
    if (!table_is_policy) {  // logits given: the policy head once per (player, state) row -- of the rows this phase can need
        const uint8_t *mt = (const uint8_t *)tree->mask_tab;
        if (phases == 3) {
            RNAD_DISPATCH_A(tree->A, hipLaunchKernelGGL((k_policy_rows<kA>), dim3(blocks_for(2 * S)), dim3(kThreads), 0, stream, 2 * S, table,
                                                        table_stride, mt, s.policy, nullptr, nullptr, nullptr, 0, S));
        } else if (sort_phase) {
            const int nu = p.cut->n_upper;
            RNAD_DISPATCH_A(tree->A, hipLaunchKernelGGL((k_policy_rows<kA>), dim3(blocks_for(2 * ((int64_t)nu + 1))), dim3(kThreads), 0, stream,
                                                        2 * S, table, table_stride, mt, s.policy, nullptr, nullptr,
                                                        (const int32_t *)p.cut->upper_list, nu, S));
        } else {
            RNAD_REQUIRE(play_rows && n_play_rows, "rnad_bucket_play: a logits table needs the list of rows that were evaluated for it");
            RNAD_DISPATCH_A(tree->A, hipLaunchKernelGGL((k_policy_rows<kA>), dim3(blocks_for(2 * S)), dim3(kThreads), 0, stream, 2 * S, table,
                                                        table_stride, mt, s.policy, play_rows, n_play_rows, nullptr, 0, S));
        }
        policy_tab = s.policy;
        policy_stride = (tree->A + 3) & ~3;
    }
    const int vec4 = (policy_stride % 4 == 0 && ((uintptr_t)policy_tab & 15) == 0) ? 1 : 0;
    bool keys_with_hist = false;
    if (sort_phase) {
        ProfScope one(PROF_BUCKET_KEYS, stream);
        const size_t keys_lds = keys_lds_bytes(p.cut->n_upper, nb, tree->A, tree->C);
        const bool walk_global = getenv("RNAD_KEYS_GLOBAL") && atoi(getenv("RNAD_KEYS_GLOBAL")) != 0;  // (tests: the fallback on any tree)
        const bool no_full_lds = getenv("RNAD_KEYS_LDS") && atoi(getenv("RNAD_KEYS_LDS")) == 0;  // (tests: the hybrid walk on small trees)
        const bool use_lds = p.cut->upper_walk && keys_lds <= kKeysLds && !walk_global && !no_full_lds;
        if (expand && !use_lds) {  // (only k_bucket_keys_lds carries the copies: a launch of their own in front of the other walks)
            float *tabs[4];
            int32_t widths[4];
            for (int k = 0; k < expand->n; ++k) {
                tabs[k] = reinterpret_cast<float *>(expand->tab[k]);
                widths[k] = 4 * expand->quads[k];
            }
            if (int rc = rnad_rows_expand(expand->rows, expand->rep_of, expand->n, tabs, widths, stream)) return rc;
            expand = nullptr;
        }
        if (use_lds) {  // the upper states' tables fit the LDS: walk there, one sort tile per workgroup
            keys_with_hist = true;
            if (keys_lds > 48 * 1024)
                RNAD_DISPATCH_TILE(p, RNAD_DISPATCH_A(tree->A, RNAD_HIP_OK(hipFuncSetAttribute((const void *)k_bucket_keys_lds<kA, kPlayLds, kTile>,
                                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)keys_lds))));
            RNAD_DISPATCH_TILE(p, RNAD_DISPATCH_A(tree->A, hipLaunchKernelGGL((k_bucket_keys_lds<kA, kPlayLds, kTile>), dim3(p.sort_blocks), dim3(kSortThreads), keys_lds, stream,
                                                        (const UpperWalk *)p.cut->upper_walk, (const int32_t *)p.cut->upper_list, p.cut->n_upper, nb,
                                                        tree->C, S, B, n_steps, policy_tab, policy_stride, (int)p.cut->host_bucket_of[1],
                                                        p.cut->n_groups, seed, device_params, lane0, s.keys, s.decisions, s.hist, norm, stage,
                                                        expand ? *expand : KeysExpand{})));
            // (the copies themselves: workgroups of the scan launch below)
        } else if (p.cut->upper_walk && p.cut->n_hot > 0 && p.cut->n_hot < p.cut->n_upper && !walk_global &&
                   !(getenv("RNAD_KEYS_HYBRID") && atoi(getenv("RNAD_KEYS_HYBRID")) == 0)) {
            // the top levels of the upper states in LDS, the rest from the global tables
            const size_t hyb_lds = std::max(keys_hybrid_lds_bytes(p.cut->n_hot, p.cut->n_upper, tree->A, tree->C), (size_t)nb * sizeof(int32_t));
            keys_with_hist = true;
            if (hyb_lds > 48 * 1024)
                RNAD_DISPATCH_TILE(p, RNAD_DISPATCH_A(tree->A, RNAD_HIP_OK(hipFuncSetAttribute((const void *)k_bucket_keys_hybrid<kA, kPlayLds, kTile>,
                                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)hyb_lds))));
            RNAD_DISPATCH_TILE(p, RNAD_DISPATCH_A(tree->A, hipLaunchKernelGGL((k_bucket_keys_hybrid<kA, kPlayLds, kTile>), dim3(p.sort_blocks), dim3(kSortThreads), hyb_lds, stream,
                                                        (const UpperWalk *)p.cut->upper_walk, (const int32_t *)p.cut->upper_list,
                                                        (const int32_t *)p.cut->hot_list, (const int32_t *)p.cut->hot_of, p.cut->n_hot, p.cut->n_upper,
                                                        tree->trans, (const int32_t *)p.cut->bucket_of, tree->C, S, B, n_steps, policy_tab, policy_stride,
                                                        vec4, (int)p.cut->host_bucket_of[1], p.cut->n_groups, seed, device_params, lane0, s.keys,
                                                        s.decisions, norm, stage, nb, s.hist)));
        } else {
            RNAD_DISPATCH_A(tree->A, hipLaunchKernelGGL((k_bucket_keys<kA, kPlay>), dim3(blocks_for(B, kThreads * kPlay)), dim3(kThreads), 0, stream, tree->trans, tree->C,
                                                        S, B, n_steps, policy_tab, policy_stride, vec4, (const int32_t *)p.cut->bucket_of,
                                                        p.cut->n_groups, seed, device_params, lane0, s.keys, s.decisions, norm, stage));
        }
    }
    const size_t lds = (size_t)nb * sizeof(int32_t);
    if (sort_phase) {
        ProfScope sort_passes(PROF_BUCKET_SORT, stream);
        if (!keys_with_hist)
            RNAD_DISPATCH_TILE(p, hipLaunchKernelGGL(k_bucket_hist<kTile>, dim3(p.sort_blocks), dim3(kSortThreads), lds, stream, B, nb, (const int32_t *)s.keys, s.hist));
        {
            const int scan_blocks = (nb + kScanCols - 1) / kScanCols;
            int copy_blocks = 0;
            if (expand) {  // a thread per (row, 16-byte chunk) and pass; two passes per thread keep the launch at a few hundred workgroups
                copy_blocks = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (expand->rows * expand->max_quads + 2 * kSortThreads - 1) / (2 * kSortThreads)));
                if (const char *e = getenv("RNAD_EXPAND_BLOCKS")) copy_blocks = std::max(1, atoi(e));  // tuning knob
            }
            hipLaunchKernelGGL(k_bucket_scan, dim3(scan_blocks + copy_blocks), dim3(kSortThreads), 0, stream, p.sort_blocks, nb, s.hist, s.totals,
                               scan_blocks, expand ? *expand : KeysExpand{});
            expand = nullptr;  // (done)
        }
        // bucket_start | first item of every bucket | as many counter rows as fit the LDS (16 waves share them in turns)
        int wave_rows = 16;
        while (wave_rows > 1 && ((2 + wave_rows) * (size_t)nb + 1) * sizeof(int32_t) > kKeysLds) wave_rows >>= 1;
        if (const char *force = getenv("RNAD_SCATTER_ROWS")) wave_rows = std::max(1, std::min(wave_rows, atoi(force)));  // tuning knob
        const size_t scatter_lds = ((2 + wave_rows) * (size_t)nb + 1) * sizeof(int32_t);
        RNAD_REQUIRE(scatter_lds <= 160 * 1024, "rnad_bucket_sort: %d buckets do not fit the sort's LDS", nb);
        if (scatter_lds > 48 * 1024)
            RNAD_DISPATCH_TILE(p, RNAD_HIP_OK(hipFuncSetAttribute((const void *)k_bucket_scatter<kTile>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)scatter_lds)));
        RNAD_DISPATCH_TILE(p, hipLaunchKernelGGL(k_bucket_scatter<kTile>, dim3(p.sort_blocks), dim3(kSortThreads), scatter_lds, stream, B, nb, (const int32_t *)s.keys,
                           (const int32_t *)s.hist, (const int32_t *)s.totals, fused ? fused_chunk(tree, p, fused->distinct) : p.chunk, (Item *)items, n_items, lane_ids, wave_rows, S,
                           p.cut->n_groups, (const int32_t *)p.cut->bucket_lo, (const int32_t *)p.cut->bucket_span,
                           (const int32_t *)p.cut->group_by_lo, staged_rows, n_staged, tr.visited, (const uint32_t *)stage.root, stage.sorted,
                           (const uint32_t *)stage.mark0, stage.root ? stage_rows0 : nullptr, stage.counts, seed, device_params, s.bucket_start));
        if (group_flags)
            hipLaunchKernelGGL(k_group_flags, dim3(blocks_for(S)), dim3(kThreads), 0, stream, S, (const int32_t *)p.cut->bucket_of,
                               p.cut->n_groups, (const int32_t *)s.totals, group_flags);
    }
    RNAD_HIP_OK(hipGetLastError());
    if (!play_phase) return 0;
    const unsigned grid = blocks_for(B);
    int alive_n = (int)grid;  // rows of alive_part the rollout kernel leaves
    {
        ProfScope one(fused ? PROF_BUCKET_LEARN : PROF_BUCKET_ROLLOUT, stream);  // (k_bucket_play_learn is booked as the learner: the larger share)
#define RNAD_BUCKET_ROLLOUT()                                                                                                            \
    hipLaunchKernelGGL((k_bucket_rollout<kA>), dim3(grid), dim3(kThreads), 0, stream, tree->trans, tree->C, S, B, tr.T_cap, policy_tab,     \
                       policy_stride, value_table, value_stride, (const uint8_t *)tree->mask_tab, seed, device_params, lane0,               \
                       (const int32_t *)lane_ids, (const unsigned long long *)s.decisions, (int32_t *)tr.indices, tr.mask_bits, tr.policy, tr.actions, \
                       tr.rewards, tr.values, s.alive_part)
        if (compact) {
            RNAD_REQUIRE(items && n_items, "rnad_rollout_bucketed_compact: the work list of the sort is needed to play");
            alive_n = (int)p.max_items;
            if (fused) {
                const int64_t A1 = tree->A + 1;
                unsigned long long *acc = (unsigned long long *)fused->accumulators;
                unsigned long long *rep = acc + 2 * S * A1;
                const int nu = p.cut->n_upper;
                double *losses_raw = (double *)(rep + (int64_t)kReplicas * 2 * std::max(nu, 1) * A1);
                int32_t *overflow = (int32_t *)(losses_raw + 4);
                const FixedPoint fx = fixed_point_for(*fused->hp);
                const CountReps cr = count_reps(tree, p, fused->accumulators, tr.T_cap + 1, nullptr, nullptr);
                RNAD_REQUIRE(!fx.check_l, "rnad_rollout_learn_bucketed_compact: a NeuRD clip of 2^29 or more takes the two-launch path");
                RNAD_REQUIRE(!tr.visited, "rnad_rollout_learn_bucketed_compact: no visited flags (lazy rows evaluate their records after the rollout)");
                const int f_chunk = fused_chunk(tree, p, fused->distinct);
                const int d_keys = f_chunk > p.chunk || getenv("RNAD_FUSED_CHUNK") ? fused_distinct_keys(tree, p, fused->distinct) : 0;
                const int d_keys_fit = (size_t)p.lds + ((size_t)d_keys + 2 + 2 * (size_t)f_chunk) * 4 <= (size_t)kFusedLdsBudget ? d_keys : 0;
                const size_t f_lds = (size_t)p.lds + (d_keys_fit ? ((size_t)d_keys_fit + 2 + 2 * (size_t)f_chunk) * 4 : 0);
#define RNAD_PLAY_LEARN()                                                                                                              \
    do {                                                                                                                               \
        auto kern = d_keys_fit > 0 ? k_bucket_play_learn<kA, REL, true> : k_bucket_play_learn<kA, REL, false>;                         \
        if (f_lds > 48 * 1024) RNAD_HIP_OK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f_lds)); \
        hipLaunchKernelGGL(kern, dim3((unsigned)p.max_items), dim3(kThreads), f_lds, stream, tree->trans, tree->C, S, B, tr.T_cap,     \
                           policy_tab, policy_stride, vec4, seed, device_params, lane0, (const int32_t *)lane_ids,                     \
                           (const unsigned long long *)s.decisions, (const Item *)items, (const int32_t *)n_items,                     \
                           (const int32_t *)p.cut->bucket_path, (const int32_t *)p.cut->bucket_lo, (const int32_t *)p.cut->path_states, \
                           std::max(p.cut->max_path, 1), p.cut->n_groups, (REL *)tr.indices, cr.alive_rep, cr.norm_rep, tr.acts,       \
                           tr.final_reward, p.cut->rows, p.path_words, std::max(nu, 1), (const int32_t *)p.cut->bucket_of, fused->fast, *fused->hp, fx, \
                           acc, rep, overflow, p.lds / 8, d_keys_fit, f_chunk);                                                        \
    } while (0)
                if (fused->leaf) {
                    // rollout (lanes counted per terminal transition), then the learner on the leaf paths of the tree: n_cols columns whatever
                    // the batch -- and none of the per-lane learner's work for lanes that share a trajectory
                    const rnad_leaf_paths_t &lf = *fused->leaf;
                    RNAD_REQUIRE(lf.n_cols >= 1 && lf.states && lf.acts && lf.final_reward && lf.items && lf.n_items && lf.col_of && lf.max_items >= 1,
                                 "rnad_rollout_learn_bucketed_compact: incomplete rnad_leaf_paths_t");
                    RNAD_REQUIRE(lf.rows == p.cut->rows && lf.T_cap == tr.T_cap && tr.T_cap == 2 * tree->max_depth,
                                 "rnad_rollout_learn_bucketed_compact: the leaf paths were built for another cut / window (rows %d vs %d, T_cap %d vs %d, "
                                 "2 * depth %d)", lf.rows, p.cut->rows, lf.T_cap, tr.T_cap, 2 * tree->max_depth);
#define RNAD_PLAY_COUNT()                                                                                                              \
    do {                                                                                                                               \
        hipLaunchKernelGGL((k_bucket_play_count<kA, REL>), dim3((unsigned)p.max_items), dim3(kThreads), 0, stream, tree->trans, tree->C, S, B,   \
                           tr.T_cap, policy_tab, policy_stride, vec4, seed, device_params, lane0, (const int32_t *)lane_ids,          \
                           (const unsigned long long *)s.decisions, (const Item *)items, (const int32_t *)n_items,                     \
                           (const int32_t *)p.cut->bucket_path, (const int32_t *)p.cut->bucket_lo, (const int32_t *)p.cut->path_states, \
                           std::max(p.cut->max_path, 1), p.cut->n_groups, (REL *)tr.indices, cr.alive_rep, cr.norm_rep, tr.acts,       \
                           tr.final_reward, lf.col_of, s.keys, lcount);                                                                \
        auto kern = k_bucket_learn_c<kA, REL, false, true>;                                                                           \
        if (p.lds > 48 * 1024) RNAD_HIP_OK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, p.lds)); \
        hipLaunchKernelGGL(kern, dim3((unsigned)lf.max_items + 7), dim3(kThreads), (size_t)p.lds, stream, tr.T_cap, lf.n_cols, S, p.cut->rows,  \
                           p.path_words, p.cut->n_groups, std::max(nu, 1), (const Item *)lf.items, lf.n_items,                        \
                           (const int32_t *)p.cut->bucket_of, (const int32_t *)p.cut->bucket_lo, (const int32_t *)p.cut->bucket_path,  \
                           (const int32_t *)p.cut->path_states, std::max(p.cut->max_path, 1), (const REL *)lf.states, fused->fast,     \
                           (const unsigned long long *)lf.acts, lf.final_reward, (const float *)nullptr, *fused->hp, fx, acc, rep,     \
                           (double *)nullptr, overflow, (const int32_t *)nullptr, 0, tr.T_cap + 1, (int32_t *)nullptr, (double *)nullptr, \
                           (const int32_t *)s.keys, (const int32_t *)s.bucket_start, (const int32_t *)s.totals, lcount);               \
    } while (0)
                    LeafCount lcount;
                    if (lf.col_count && lf.bucket_col0 && lf.crowded_lanes > 0)
                        lcount = LeafCount{lf.col_count, lf.bucket_col0, (const int32_t *)s.totals, lf.crowded_lanes};
                    RNAD_DISPATCH_REL(p, RNAD_DISPATCH_A(tree->A, RNAD_PLAY_COUNT()));
#undef RNAD_PLAY_COUNT
                } else
                RNAD_DISPATCH_REL(p, RNAD_DISPATCH_A(tree->A, RNAD_PLAY_LEARN()));
#undef RNAD_PLAY_LEARN
            } else
            RNAD_DISPATCH_REL(p, RNAD_DISPATCH_A(tree->A, hipLaunchKernelGGL(
                                     (k_bucket_rollout_items<kA, REL, kRolloutLanes>), dim3((unsigned)p.max_items), dim3(kThreads / kRolloutLanes), 0, stream, tree->trans, tree->C,
                                     S, B, tr.T_cap, policy_tab, policy_stride, vec4, seed, device_params, lane0, (const int32_t *)lane_ids,
                                     (const unsigned long long *)s.decisions, (const Item *)items, (const int32_t *)n_items,
                                     (const int32_t *)p.cut->bucket_path, (const int32_t *)p.cut->bucket_lo, (const int32_t *)p.cut->path_states,
                                     std::max(p.cut->max_path, 1), p.cut->n_groups, (REL *)tr.indices, s.alive_part, tr.acts, tr.final_reward,
                                     tr.visited)));
        } else {
            RNAD_DISPATCH_A(tree->A, RNAD_BUCKET_ROLLOUT());
        }
#undef RNAD_BUCKET_ROLLOUT
    }
    if (tr.alive && !fused)  // (NULL: the caller lets rnad_learn_bucketed_compact add the counts up, or calls rnad_bucket_alive)
        hipLaunchKernelGGL(k_bucket_alive, dim3(tr.T_cap + 1), dim3(kThreads), 0, stream, alive_n, tr.T_cap + 1, (const int32_t *)s.alive_part,
                           tr.alive, norm);
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}
}  // namespace

extern "C" int rnad_bucket_sort(const rnad_tree_t *tree, int T_cap, int64_t B, const float *table, int64_t table_stride, int table_is_policy,
                                uint64_t seed, int64_t lane0, const rnad_step_params_t *device_params, void *scratch, int32_t *lane_ids,
                                int32_t *items, int32_t *n_items, double *norm, int32_t *group_flags, int32_t *staged_rows, int64_t *n_staged,
                                int32_t *visited, void *stage, int32_t *stage_rows0, void *stream) {
    RNAD_REQUIRE(tree && table && scratch && lane_ids && items && n_items, "rnad_bucket_sort: null argument");
    RNAD_REQUIRE(!stage_rows0 || stage, "rnad_bucket_sort: stage_rows0 goes with stage");
    RNAD_REQUIRE(!staged_rows == !n_staged, "rnad_bucket_sort: staged_rows and n_staged go together");
    RNAD_REQUIRE(T_cap >= 1 && T_cap <= kCompactSteps && B >= 1, "rnad_bucket_sort: 1 <= T_cap <= %d, got %d", kCompactSteps, T_cap);
    RNAD_REQUIRE(table_stride >= tree->A, "rnad_bucket_sort: bad table stride");
    const RolloutBuffers out{T_cap, B, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, visited};
    return rollout_bucketed_impl(tree, out, true, table, table_stride, table_is_policy, nullptr, 1, seed, lane0, device_params, scratch,
                                 lane_ids, items, n_items, norm, (hipStream_t)stream, 1, group_flags, nullptr, nullptr, staged_rows, n_staged,
                                 false, stage, stage_rows0);
}

extern "C" int64_t rnad_bucket_stage_bytes(const rnad_tree_t *tree, int64_t B) {
    if (!tree || B < 1) return -1;
    if (tree->S > ((int64_t)1 << kStageRootBits)) return -1;  // (a root word holds the state in 26 bits)
    return (int64_t)(2 * sizeof(unsigned long long) + 2 * (size_t)B * sizeof(uint32_t) + 2 * (size_t)tree->S * sizeof(uint32_t));
}

extern "C" int rnad_bucket_stage_rows(const rnad_tree_t *tree, int64_t B, int level, uint64_t seed, const rnad_step_params_t *device_params,
                                      void *stage, int32_t *rows, void *stream) {
    RNAD_REQUIRE(tree && stage && rows && (level == 0 || level == 1), "rnad_bucket_stage_rows: null argument / level must be 0 or 1");
    Plan p;
    RNAD_REQUIRE(make_plan(tree, B, p), "rnad_bucket_stage_rows: this tree / batch cannot be bucketed (see rnad_bucket_plan)");
    RNAD_REQUIRE_SIZED(tree, B, p);
    const StageOut st = carve_stage(stage, B, tree->S);
    if (level == 0)
        hipLaunchKernelGGL(k_stage_rows<0>, dim3(blocks_for(tree->S, kSortThreads * kStagePer)), dim3(kSortThreads), 0, (hipStream_t)stream, tree->S, (const uint32_t *)st.mark0,
                           (const int32_t *)p.cut->anchor1, seed, device_params, rows, st.counts);
    else
        hipLaunchKernelGGL(k_stage_rows<1>, dim3(blocks_for(tree->S, kSortThreads * kStagePer)), dim3(kSortThreads), 0, (hipStream_t)stream, tree->S, (const uint32_t *)st.mark1,
                           (const int32_t *)p.cut->anchor1, seed, device_params, rows, st.counts + 1);
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int rnad_bucket_stage_walk(const rnad_tree_t *tree, int T_cap, int64_t B, const float *policy_rows, int64_t stride, uint64_t seed,
                                      int64_t lane0, const rnad_step_params_t *device_params, const void *scratch, const int32_t *lane_ids,
                                      void *stage, void *stream) {
    RNAD_REQUIRE(tree && policy_rows && scratch && stage && lane_ids, "rnad_bucket_stage_walk: null argument");
    RNAD_REQUIRE(stride >= tree->A && T_cap >= 1, "rnad_bucket_stage_walk: bad table stride / T_cap");
    Plan p;
    RNAD_REQUIRE(make_plan(tree, B, p), "rnad_bucket_stage_walk: this tree / batch cannot be bucketed (see rnad_bucket_plan)");
    RNAD_REQUIRE_SIZED(tree, B, p);
    const StageOut st = carve_stage(stage, B, tree->S);
    const int vec4 = (stride % 4 == 0 && ((uintptr_t)policy_rows & 15) == 0) ? 1 : 0;
    RNAD_DISPATCH_A(tree->A, hipLaunchKernelGGL((k_stage_walk<kA>), dim3(blocks_for(B)), dim3(kThreads), 0, (hipStream_t)stream, tree->trans, tree->C,
                                                tree->S, B, T_cap, policy_rows, stride, vec4, (const uint32_t *)st.sorted, lane_ids, seed,
                                                device_params, lane0, st.mark1));
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int rnad_bucket_play(const rnad_tree_t *tree, int T_cap, int64_t B, const float *table, int64_t table_stride, int table_is_policy,
                                const int32_t *rows, const int64_t *n_rows, uint64_t seed, int64_t lane0,
                                const rnad_step_params_t *device_params, void *scratch, const int32_t *lane_ids, const int32_t *items,
                                const int32_t *n_items, double *norm, void *states, int32_t *alive, uint64_t *acts, float *final_reward,
                                int32_t *visited, int visited_is_clear, void *stream) {
    void *indices = states;
    RNAD_REQUIRE(tree && table && scratch && lane_ids && items && n_items && indices && acts && final_reward, "rnad_bucket_play: null argument");
    RNAD_REQUIRE(T_cap >= 1 && T_cap <= kCompactSteps && B >= 1, "rnad_bucket_play: 1 <= T_cap <= %d, got %d", kCompactSteps, T_cap);
    RNAD_REQUIRE(table_stride >= tree->A && (!rows == !n_rows), "rnad_bucket_play: bad table stride / row list");
    const RolloutBuffers out{T_cap, B, indices, nullptr, nullptr, nullptr, nullptr, nullptr, alive, (unsigned long long *)acts, final_reward,
                             visited};
    return rollout_bucketed_impl(tree, out, true, table, table_stride, table_is_policy, nullptr, 1, seed, lane0, device_params, scratch,
                                 const_cast<int32_t *>(lane_ids), const_cast<int32_t *>(items), const_cast<int32_t *>(n_items), norm,
                                 (hipStream_t)stream, 2, nullptr, rows, n_rows, nullptr, nullptr, visited_is_clear != 0);
}

extern "C" int rnad_bucket_alive(const rnad_tree_t *tree, int T_cap, int64_t B, const void *scratch, int32_t *alive, double *norm, void *stream) {
    RNAD_REQUIRE(tree && scratch && alive, "rnad_bucket_alive: null argument");
    RNAD_REQUIRE(T_cap >= 1 && T_cap <= kMaxSteps && B >= 1, "rnad_bucket_alive: bad shape");
    Plan p;
    RNAD_REQUIRE(make_plan(tree, B, p), "rnad_bucket_alive: this tree / batch cannot be bucketed (see rnad_bucket_plan)");
    RNAD_REQUIRE_SIZED(tree, B, p);
    const Scratch s = carve_scratch(const_cast<void *>(scratch), B, p);
    hipLaunchKernelGGL(k_bucket_alive, dim3(T_cap + 1), dim3(kThreads), 0, (hipStream_t)stream, (int)alive_rows(tree, B, p, true), T_cap + 1,
                       (const int32_t *)s.alive_part, alive, norm);
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int rnad_rollout_bucketed(const rnad_tree_t *tree, const rnad_traj_t *tr, const float *table, int64_t table_stride,
                                     int table_is_policy, const float *value_table, int64_t value_stride, uint64_t seed, int64_t lane0,
                                     const rnad_step_params_t *device_params, void *scratch, int32_t *lane_ids, int32_t *items,
                                     int32_t *n_items, double *norm, void *stream) {
    RNAD_REQUIRE(tree && tr && table && scratch && lane_ids && items && n_items, "rnad_rollout_bucketed: null argument");
    RNAD_REQUIRE(tr->indices && tr->mask_bits && tr->policy && tr->actions && tr->rewards && tr->alive,
                 "rnad_rollout_bucketed: trajectory has a null buffer");
    RNAD_REQUIRE(tr->T_cap >= 1 && tr->T_cap <= kMaxSteps && tr->B >= 1, "rnad_rollout_bucketed: bad trajectory shape T_cap=%d B=%lld",
                 tr->T_cap, (long long)tr->B);
    RNAD_REQUIRE(table_stride >= tree->A && (!value_table || value_stride >= 1), "rnad_rollout_bucketed: bad table stride");
    const RolloutBuffers out{(int)tr->T_cap, tr->B, tr->indices, tr->mask_bits, tr->policy, tr->actions, tr->rewards, tr->values, tr->alive,
                             nullptr, nullptr, nullptr};
    return rollout_bucketed_impl(tree, out, false, table, table_stride, table_is_policy, value_table, value_stride, seed, lane0, device_params,
                                 scratch, lane_ids, items, n_items, norm, (hipStream_t)stream);
}

extern "C" int rnad_rollout_bucketed_compact(const rnad_tree_t *tree, int T_cap, int64_t B, const float *table, int64_t table_stride,
                                             int table_is_policy, uint64_t seed, int64_t lane0, const rnad_step_params_t *device_params,
                                             void *scratch, int32_t *lane_ids, int32_t *items, int32_t *n_items, double *norm,
                                             void *states, int32_t *alive, uint64_t *acts, float *final_reward, int32_t *visited,
                                             void *stream) {
    void *indices = states;
    RNAD_REQUIRE(tree && table && scratch && lane_ids && items && n_items && indices && acts && final_reward,
                 "rnad_rollout_bucketed_compact: null argument");
    RNAD_REQUIRE(alive || norm, "rnad_rollout_bucketed_compact: deferred alive counts (alive == NULL) need `norm` (it is cleared here)");
    RNAD_REQUIRE(T_cap >= 1 && T_cap <= kCompactSteps && B >= 1, "rnad_rollout_bucketed_compact: 1 <= T_cap <= %d (3 bits per step), got %d",
                 kCompactSteps, T_cap);
    RNAD_REQUIRE(table_stride >= tree->A, "rnad_rollout_bucketed_compact: bad table stride");
    static_assert(RNAD_MAX_ACTIONS <= 8, "3 bits per action");
    const RolloutBuffers out{T_cap, B, indices, nullptr, nullptr, nullptr, nullptr, nullptr, alive, (unsigned long long *)acts, final_reward,
                             visited};
    return rollout_bucketed_impl(tree, out, true, table, table_stride, table_is_policy, nullptr, 1, seed, lane0, device_params, scratch,
                                 lane_ids, items, n_items, norm, (hipStream_t)stream);
}

// rnad_rollout_bucketed_compact with the copies of rnad_rows_expand carried by its keys pass (distinct observations: the actor's table and
// the learner's records were evaluated on one representative row per observation; `table` must be one of the tables being expanded, or
// already complete).
extern "C" int rnad_rollout_bucketed_compact_expand(const rnad_tree_t *tree, int T_cap, int64_t B, const float *table, int64_t table_stride,
                                                    int table_is_policy, uint64_t seed, int64_t lane0,
                                                    const rnad_step_params_t *device_params, void *scratch, int32_t *lane_ids, int32_t *items,
                                                    int32_t *n_items, double *norm, void *states, int32_t *alive, uint64_t *acts,
                                                    float *final_reward, int32_t *visited, const int32_t *rep_of, int n_tables,
                                                    float *const *tables, const int32_t *floats_per_row, void *stream) {
    void *indices = states;
    RNAD_REQUIRE(tree && table && scratch && lane_ids && items && n_items && indices && acts && final_reward && rep_of && tables && floats_per_row,
                 "rnad_rollout_bucketed_compact_expand: null argument");
    RNAD_REQUIRE(alive || norm, "rnad_rollout_bucketed_compact_expand: deferred alive counts (alive == NULL) need `norm` (it is cleared here)");
    RNAD_REQUIRE(T_cap >= 1 && T_cap <= kCompactSteps && B >= 1, "rnad_rollout_bucketed_compact_expand: 1 <= T_cap <= %d, got %d", kCompactSteps, T_cap);
    RNAD_REQUIRE(table_stride >= tree->A && table_is_policy, "rnad_rollout_bucketed_compact_expand: the actor must be a table of policy rows");
    RNAD_REQUIRE(n_tables >= 1 && n_tables <= 4, "rnad_rollout_bucketed_compact_expand: 1..4 tables");
    KeysExpand ex;
    ex.rep_of = rep_of;
    ex.rows = 2 * tree->S;
    ex.n = n_tables;
    for (int k = 0; k < n_tables; ++k) {
        RNAD_REQUIRE(tables[k] && floats_per_row[k] > 0 && floats_per_row[k] % 4 == 0 && ((uintptr_t)tables[k] & 15) == 0,
                     "rnad_rollout_bucketed_compact_expand: table %d must be 16-byte aligned with a row of a multiple of 4 floats", k);
        ex.tab[k] = reinterpret_cast<float4 *>(tables[k]);
        ex.quads[k] = floats_per_row[k] / 4;
        ex.max_quads = std::max(ex.max_quads, ex.quads[k]);
    }
    const RolloutBuffers out{T_cap, B, indices, nullptr, nullptr, nullptr, nullptr, nullptr, alive, (unsigned long long *)acts, final_reward,
                             visited};
    return rollout_bucketed_impl(tree, out, true, table, table_stride, table_is_policy, nullptr, 1, seed, lane0, device_params, scratch,
                                 lane_ids, items, n_items, norm, (hipStream_t)stream, 3, nullptr, nullptr, nullptr, nullptr, nullptr, false, nullptr,
                                 nullptr, &ex);
}

// rnad_rollout_bucketed_compact(_expand) and rnad_learn_bucketed_compact of the batch it plays, T = T_cap, in one call: keys, sort, then
// ONE launch in which every work item's workgroup plays its lanes and adds up their update (k_bucket_play_learn), then the alive
// counts and -- RNAD_PLAY_LEARN_FINISH -- rnad_bucket_finish with the batch's own normalisers (a data-parallel caller leaves the flag out,
// all-reduces `norm` and calls rnad_bucket_finish itself; or it hands in norm_global -- the normalisers of the GLOBAL batch, known without
// a collective on a tree whose episodes all have the same length -- and the finish divides by those).  RNAD_PLAY_LEARN_DISTINCT: the learner half once per distinct trajectory of a
// work item (struct Distinct).  The trajectory, the counts and the gradient tables are those of the two calls, bit for bit.
extern "C" int rnad_rollout_learn_bucketed_compact(const rnad_tree_t *tree, int T_cap, int64_t B, const float *table, int64_t table_stride,
                                                   uint64_t seed, int64_t lane0, const rnad_step_params_t *device_params, void *scratch,
                                                   int32_t *lane_ids, int32_t *items, int32_t *n_items, double *norm, void *states,
                                                   int32_t *alive, uint64_t *acts, float *final_reward, const int32_t *rep_of, int n_tables,
                                                   float *const *tables, const int32_t *floats_per_row, const float *fast_records,
                                                   const rnad_learn_params_t *hp, void *accumulators, int flags, const double *norm_global,
                                                   float *dlogit_tab, float *dv_tab, const int32_t *rows, const int64_t *n_rows,
                                                   const rnad_row_groups_t *groups, const rnad_leaf_paths_t *leaf, void *stream) {
    RNAD_REQUIRE(tree && table && scratch && lane_ids && items && n_items && norm && states && alive && acts && final_reward && fast_records &&
                     hp && accumulators,
                 "rnad_rollout_learn_bucketed_compact: null argument");
    RNAD_REQUIRE(!leaf || !(flags & RNAD_PLAY_LEARN_DISTINCT), "rnad_rollout_learn_bucketed_compact: leaf paths or distinct trajectories, not both");
    RNAD_REQUIRE(T_cap >= 1 && T_cap <= kCompactSteps && B >= 1, "rnad_rollout_learn_bucketed_compact: 1 <= T_cap <= %d, got %d", kCompactSteps, T_cap);
    RNAD_REQUIRE(table_stride >= tree->A, "rnad_rollout_learn_bucketed_compact: bad table stride");
    RNAD_REQUIRE(((uintptr_t)fast_records & 15) == 0, "rnad_rollout_learn_bucketed_compact: fast_records must be 16-byte aligned");
    const bool finish = (flags & RNAD_PLAY_LEARN_FINISH) != 0;
    RNAD_REQUIRE(!finish || (dlogit_tab && dv_tab), "rnad_rollout_learn_bucketed_compact: finish needs the gradient tables");
    RNAD_REQUIRE(!rows == !n_rows, "rnad_rollout_learn_bucketed_compact: rows and n_rows go together");
    RNAD_REQUIRE(n_tables >= 0 && n_tables <= 4 && (n_tables == 0 || (rep_of && tables && floats_per_row)),
                 "rnad_rollout_learn_bucketed_compact: 0..4 tables to expand, with rep_of");
    KeysExpand ex;
    if (n_tables > 0) {
        ex.rep_of = rep_of;
        ex.rows = 2 * tree->S;
        ex.n = n_tables;
        for (int k = 0; k < n_tables; ++k) {
            RNAD_REQUIRE(tables[k] && floats_per_row[k] > 0 && floats_per_row[k] % 4 == 0 && ((uintptr_t)tables[k] & 15) == 0,
                         "rnad_rollout_learn_bucketed_compact: table %d must be 16-byte aligned with a row of a multiple of 4 floats", k);
            ex.tab[k] = reinterpret_cast<float4 *>(tables[k]);
            ex.quads[k] = floats_per_row[k] / 4;
            ex.max_quads = std::max(ex.max_quads, ex.quads[k]);
        }
    }
    const FusedLearn fused{fast_records, hp, accumulators, (flags & RNAD_PLAY_LEARN_DISTINCT) != 0, leaf};
    const RolloutBuffers out{T_cap, B, states, nullptr, nullptr, nullptr, nullptr, nullptr, alive, (unsigned long long *)acts, final_reward, nullptr};
    if (int rc = rollout_bucketed_impl(tree, out, true, table, table_stride, 1, nullptr, 1, seed, lane0, device_params, scratch, lane_ids, items,
                                       n_items, norm, (hipStream_t)stream, 3, nullptr, nullptr, nullptr, nullptr, nullptr, false, nullptr, nullptr,
                                       n_tables > 0 ? &ex : nullptr, &fused))
        return rc;
    Plan p;
    RNAD_REQUIRE(make_plan(tree, B, p), "rnad_rollout_learn_bucketed_compact: this tree / batch cannot be bucketed (see rnad_bucket_plan)");
    const CountReps cr = count_reps(tree, p, accumulators, T_cap + 1, alive, norm);
    if (!finish) {  // the counts on their own: the caller all-reduces `norm` before its rnad_bucket_finish
        hipLaunchKernelGGL(k_bucket_alive_rep, dim3(1), dim3(kThreads), 0, (hipStream_t)stream, cr.T1, cr.alive_rep, cr.norm_rep, alive, norm);
        RNAD_HIP_OK(hipGetLastError());
        return 0;
    }
    RNAD_REQUIRE(norm_global != norm, "rnad_rollout_learn_bucketed_compact: norm_global must not be the `norm` output");
    return finish_impl(tree, p, norm_global, hp, accumulators, nullptr, dlogit_tab, dv_tab, rows, n_rows, groups, (hipStream_t)stream, &cr);
}

// The dense buffers of a compact trajectory: slot (t, j) from indices[t, j] (and indices[t + 1, j] for the reward) alone.
namespace {
template <int A>
__global__ __launch_bounds__(kThreads) void k_bucket_expand(const Trans *__restrict__ trans, int C, int T, int64_t B, int64_t S,
                                                            const int32_t *__restrict__ indices,
                                                            const unsigned long long *__restrict__ acts,
                                                            const float *__restrict__ final_reward, const float *__restrict__ rec_,
                                                            const uint8_t *__restrict__ mask_tab, uint8_t *__restrict__ mbits,
                                                            float *__restrict__ policy, int32_t *__restrict__ actions,
                                                            float *__restrict__ rewards) {
    const int64_t j = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int t = blockIdx.y;
    if (j >= B) return;
    const int64_t i = (int64_t)t * B + j;
    const int state = indices[i];
    const int64_t row = (int64_t)(t & 1) * S + state;  // an absorbed slot shows the row of state 0, as the dense rollout writes it
    mbits[i] = mask_tab[row];
#pragma unroll
    for (int a = 0; a < A; ++a) policy[i * A + a] = rec_[row * kRowStride<A> + 3 * A + 3 + a];
    const int action = state != 0 ? (int)(acts[j] >> (3 * t)) & 7 : 0;
    actions[i] = action;
    float rew = 0.0f;  // row turns: torch.zeros (episode.py:101); absorbed slots: state 0 pays value 0
    if ((t & 1) && state != 0) {
        const int next = indices[i + B];
        if (next == 0) {
            rew = final_reward[j];
        } else {
            // rewards *= (indices == 0) (episode.py:120-121) keeps the sign of the payoff it zeroes: -0.0 under a negative value.  The
            // outcome taken is the one that leads to `next` (child ids are unique).
            const int prev = (int)(acts[j] >> (3 * (t - 1))) & 7;
            const Trans *e = trans + (((int64_t)state * A + prev) * A + action) * C;
            for (int c = 0; c < C; ++c) {
                const Trans et = e[c];
                if (et.next == next) {
                    rew = et.value * 0.0f;
                    break;
                }
            }
        }
    }
    rewards[i] = rew;
}
}  // namespace

extern "C" int rnad_bucket_expand(const rnad_tree_t *tree, int T, int64_t B, const int32_t *indices, const uint64_t *acts,
                                  const float *final_reward, const float *records, uint8_t *mask_bits, float *policy, int32_t *actions,
                                  float *rewards, void *stream) {
    RNAD_REQUIRE(tree && indices && acts && final_reward && records && mask_bits && policy && actions && rewards,
                 "rnad_bucket_expand: null argument");
    RNAD_REQUIRE(T >= 1 && T <= kCompactSteps && B >= 1, "rnad_bucket_expand: bad shape");
    RNAD_DISPATCH_A(tree->A, hipLaunchKernelGGL((k_bucket_expand<kA>), dim3(blocks_for(B), (unsigned)T), dim3(kThreads), 0, (hipStream_t)stream,
                                                tree->trans, tree->C, T, B, tree->S, indices, (const unsigned long long *)acts, final_reward, records,
                                                (const uint8_t *)tree->mask_tab, mask_bits, policy, actions, rewards));
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

namespace {
// upper rows out of their replicas, then sums -> normalised fp32 tables (and the two logged losses)
int finish_impl(const rnad_tree_t *tree, const Plan &p, const double *norm, const rnad_learn_params_t *hp, void *accumulators, double *losses,
                float *dlogit_tab, float *dv_tab, const int32_t *rows, const int64_t *n_rows, const rnad_row_groups_t *groups,
                hipStream_t stream, const CountReps *counts) {
    const int64_t S = tree->S, A1 = tree->A + 1;
    unsigned long long *acc = (unsigned long long *)accumulators;
    unsigned long long *rep = acc + 2 * S * A1;
    const int nu = p.cut->n_upper;
    double *losses_raw = (double *)(rep + (int64_t)kReplicas * 2 * std::max(nu, 1) * A1);
    int32_t *overflow = (int32_t *)(losses_raw + 4);
    const FixedPoint fx = fixed_point_for(*hp);
    const int n_multi = groups ? groups->n_groups : 0;
    RNAD_REQUIRE(n_multi >= 0 && (n_multi == 0 || (groups->start && groups->order && rows)),
                 "rnad_bucket_finish: row groups come with their start / order arrays and with the list of the rows outside them");
    ProfScope fin(PROF_BUCKET_FINISH, stream);
    const unsigned row_blocks = std::min(blocks_for(2 * S, kThreads * kFinishRows), 512u), upper_blocks = nu > 0 ? blocks_for(2 * (int64_t)nu, kThreads / 64) : 0;
    const unsigned group_blocks = n_multi > 0 ? blocks_for(n_multi, kThreads / 64) : 0;
    RNAD_DISPATCH_A(tree->A, hipLaunchKernelGGL((k_bucket_finish<kA>), dim3(row_blocks + upper_blocks + group_blocks), dim3(kThreads), 0, stream, S,
                                                (int)row_blocks, rows, n_rows, nu, p.cut->n_groups, (const int32_t *)p.cut->upper_list,
                                                (const int32_t *)p.cut->bucket_of, acc, rep, norm, hp->w_v, hp->w_n, fx, overflow,
                                                losses_raw, losses, dlogit_tab, dv_tab, (int)upper_blocks, n_multi,
                                                n_multi ? groups->start : (const int32_t *)nullptr,
                                                n_multi ? groups->order : (const int32_t *)nullptr,
                                                n_multi ? groups->first : (const int32_t *)nullptr, (groups && rows) ? groups->rows_below_cut : 0, counts ? counts->alive_rep : (int32_t *)nullptr,
                                                counts ? counts->norm_rep : (double *)nullptr, counts ? counts->T1 : 0,
                                                counts ? counts->alive_out : (int32_t *)nullptr, counts ? counts->norm_out : (double *)nullptr));
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

int learn_bucketed_impl(const rnad_tree_t *tree, int T, int64_t B, const void *indices, const int32_t *actions, const float *rewards,
                        const float *mu, const unsigned long long *acts, const float *final_reward, const float *records,
                        const float *fast, const int32_t *items, const int32_t *n_items, const double *norm, const rnad_learn_params_t *hp,
                        void *accumulators, double *losses, float *dlogit_tab, float *dv_tab, const int32_t *rows, const int64_t *n_rows,
                        const void *rollout_scratch, int rollout_T_cap, int32_t *alive_out, double *norm_out, const rnad_row_groups_t *groups,
                        hipStream_t stream) {
    const bool compact = acts != nullptr;  // (indices: the relative states then)
    Plan p;
    RNAD_REQUIRE(make_plan(tree, B, p), "rnad_learn_bucketed: this tree / batch cannot be bucketed (see rnad_bucket_plan)");
    const int64_t S = tree->S, A1 = tree->A + 1;
    unsigned long long *acc = (unsigned long long *)accumulators;
    unsigned long long *rep = acc + 2 * S * A1;
    const int nu = p.cut->n_upper;
    double *losses_raw = (double *)(rep + (int64_t)kReplicas * 2 * std::max(nu, 1) * A1);
    int32_t *overflow = (int32_t *)(losses_raw + 4);
    const FixedPoint fx = fixed_point_for(*hp);
    const int32_t *alive_part = nullptr;
    const int T1 = rollout_T_cap + 1;
    if (rollout_scratch) alive_part = carve_scratch(const_cast<void *>(rollout_scratch), B, p).alive_part;
    const unsigned learn_grid = (unsigned)std::max<int64_t>(p.max_items, alive_part ? T1 : 0);
    // (the loss sums and the overflow flag are zero here: k_bucket_finish of the previous update cleared them.  Not hipMemsetAsync:
    // the memset node of a captured graph was seen to write garbage after ~57 replays on ROCm 7.2,
    // tests/test_hip_graph.py::test_many_replays_stay_finite)
    ProfScope prof(PROF_LEARN, stream);
#define RNAD_BUCKET_LEARN_C(LOSSES)                                                                                                   \
    do {                                                                                                                              \
        auto kern = k_bucket_learn_c<kA, REL, LOSSES>;                                                                                \
        if (p.lds > 48 * 1024) RNAD_HIP_OK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, p.lds)); \
        hipLaunchKernelGGL(kern, dim3(learn_grid), dim3(kThreads), (size_t)p.lds, stream, T, B, S, p.cut->rows, p.path_words,         \
                           p.cut->n_groups, std::max(nu, 1), (const Item *)items, n_items, (const int32_t *)p.cut->bucket_of,         \
                           (const int32_t *)p.cut->bucket_lo, (const int32_t *)p.cut->bucket_path, (const int32_t *)p.cut->path_states, \
                           std::max(p.cut->max_path, 1), (const REL *)indices, fast, acts, final_reward, records, *hp, fx, acc, rep,  \
                           losses ? losses_raw : (double *)nullptr, overflow, alive_part, (int)alive_rows(tree, B, p, true), T1,      \
                           alive_out, norm_out, (const int32_t *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr, LeafCount{}); \
    } while (0)
    {
        ProfScope one(PROF_BUCKET_LEARN, stream);
        if (compact && (losses || fx.check_l)) {  // (the LOSSES instantiation also range-checks the dL/dlogit addends of a clip >= 2^29)
            RNAD_REQUIRE(records, "rnad_learn_bucketed_compact: a NeuRD clip of 2^29 or more needs the dense records too");
            RNAD_DISPATCH_REL(p, RNAD_DISPATCH_A(tree->A, RNAD_BUCKET_LEARN_C(true)));
        } else if (compact) {
            RNAD_DISPATCH_REL(p, RNAD_DISPATCH_A(tree->A, RNAD_BUCKET_LEARN_C(false)));
        } else {
            if (p.lds > 48 * 1024)
                RNAD_DISPATCH_A(tree->A, RNAD_HIP_OK(hipFuncSetAttribute((const void *)k_bucket_learn<kA>, hipFuncAttributeMaxDynamicSharedMemorySize, p.lds)));
            RNAD_DISPATCH_A(tree->A, hipLaunchKernelGGL((k_bucket_learn<kA>), dim3((unsigned)p.max_items), dim3(kThreads), (size_t)p.lds, stream, T, B, S,
                                                        p.cut->rows, p.path_words, p.cut->n_groups, std::max(nu, 1), (const Item *)items, n_items,
                                                        (const int32_t *)p.cut->bucket_of, (const int32_t *)p.cut->bucket_lo,
                                                        (const int32_t *)p.cut->bucket_path, (const int32_t *)indices, actions, rewards, mu, records,
                                                        *hp, fx, acc, rep, losses ? losses_raw : (double *)nullptr, overflow));
        }
    }
#undef RNAD_BUCKET_LEARN_C
    RNAD_HIP_OK(hipGetLastError());
    if (!norm) return 0;  // the caller completes the update with rnad_bucket_finish once the normalisers are known
    return finish_impl(tree, p, norm, hp, accumulators, losses, dlogit_tab, dv_tab, rows, n_rows, groups, stream);
}
}  // namespace

extern "C" int rnad_bucket_indices(const rnad_tree_t *tree, int T1, int64_t B, const void *states, const int32_t *items, const int32_t *n_items,
                                   int32_t *indices, void *stream) {
    RNAD_REQUIRE(tree && states && items && n_items && indices, "rnad_bucket_indices: null argument");
    RNAD_REQUIRE(T1 >= 1 && T1 <= kCompactSteps + 1 && B >= 1, "rnad_bucket_indices: bad shape");
    Plan p;
    RNAD_REQUIRE(make_plan(tree, B, p), "rnad_bucket_indices: this tree / batch cannot be bucketed (see rnad_bucket_plan)");
    RNAD_DISPATCH_REL(p, hipLaunchKernelGGL((k_bucket_indices<REL>), dim3((unsigned)p.max_items), dim3(kThreads), 0, (hipStream_t)stream, T1, B,
                                            (const Item *)items, n_items, (const int32_t *)p.cut->bucket_path, (const int32_t *)p.cut->bucket_lo,
                                            (const int32_t *)p.cut->path_states, std::max(p.cut->max_path, 1), p.cut->n_groups, (const REL *)states,
                                            indices));
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int rnad_bucket_pack_states(const rnad_tree_t *tree, int T1, int64_t B, const int32_t *indices, const int32_t *items,
                                       const int32_t *n_items, void *states, int32_t *mismatch, void *stream) {
    RNAD_REQUIRE(tree && states && items && n_items && indices && mismatch, "rnad_bucket_pack_states: null argument");
    RNAD_REQUIRE(T1 >= 1 && T1 <= kCompactSteps + 1 && B >= 1, "rnad_bucket_pack_states: bad shape");
    Plan p;
    RNAD_REQUIRE(make_plan(tree, B, p), "rnad_bucket_pack_states: this tree / batch cannot be bucketed (see rnad_bucket_plan)");
    RNAD_DISPATCH_REL(p, hipLaunchKernelGGL((k_bucket_pack<REL>), dim3((unsigned)p.max_items), dim3(kThreads), 0, (hipStream_t)stream, T1, B, p.cut->rows,
                                            (const Item *)items, n_items, (const int32_t *)p.cut->bucket_path, (const int32_t *)p.cut->bucket_lo,
                                            (const int32_t *)p.cut->path_states, std::max(p.cut->max_path, 1), p.cut->n_groups, indices,
                                            (REL *)states, mismatch, INT32_MAX));
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int rnad_leaf_paths_pack(const rnad_tree_t *tree, int64_t plan_B, int T1, int64_t n_cols, const int32_t *indices, const int32_t *items,
                                    const int32_t *n_items, int32_t max_items, void *states, int32_t *mismatch, int32_t *rows_out,
                                    int32_t *rel_bytes_out, void *stream) {
    RNAD_REQUIRE(tree && states && items && n_items && indices && mismatch && rows_out && rel_bytes_out, "rnad_leaf_paths_pack: null argument");
    RNAD_REQUIRE(T1 >= 1 && T1 <= kCompactSteps + 1 && n_cols >= 1 && max_items >= 1, "rnad_leaf_paths_pack: bad shape");
    Plan p;
    RNAD_REQUIRE(make_plan(tree, plan_B, p), "rnad_leaf_paths_pack: this tree / batch cannot be bucketed (see rnad_bucket_plan)");
    // (a lane's bucket and its leaf column's bucket pair up because every episode leaves the tree inside the window: T1 - 1 = 2 * depth steps
    // are enough on any tree, ragged episode lengths included -- tests/test_hip_leaf.py plays those)
    RNAD_REQUIRE(T1 - 1 == 2 * tree->max_depth, "rnad_leaf_paths_pack: leaf paths span the whole window, T1 = 2 * depth + 1");
    *rows_out = p.cut->rows;
    *rel_bytes_out = p.rel_bytes;
    RNAD_DISPATCH_REL(p, hipLaunchKernelGGL((k_bucket_pack<REL>), dim3((unsigned)max_items), dim3(kThreads), 0, (hipStream_t)stream, T1, n_cols, p.cut->rows,
                                            (const Item *)items, n_items, (const int32_t *)p.cut->bucket_path, (const int32_t *)p.cut->bucket_lo,
                                            (const int32_t *)p.cut->path_states, std::max(p.cut->max_path, 1), p.cut->n_groups, indices,
                                            (REL *)states, mismatch, kThreads));
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int rnad_bucket_finish(const rnad_tree_t *tree, int64_t B, const double *norm, const rnad_learn_params_t *hp, void *accumulators,
                                  double *losses, float *dlogit_tab, float *dv_tab, const int32_t *rows, const int64_t *n_rows,
                                  const rnad_row_groups_t *groups, void *stream) {
    RNAD_REQUIRE(tree && norm && hp && accumulators && dlogit_tab && dv_tab, "rnad_bucket_finish: null argument");
    RNAD_REQUIRE(!rows == !n_rows, "rnad_bucket_finish: rows and n_rows go together");
    Plan p;
    RNAD_REQUIRE(make_plan(tree, B, p), "rnad_bucket_finish: this tree / batch cannot be bucketed (see rnad_bucket_plan)");
    return finish_impl(tree, p, norm, hp, accumulators, losses, dlogit_tab, dv_tab, rows, n_rows, groups, (hipStream_t)stream);
}

extern "C" int rnad_learn_bucketed(const rnad_tree_t *tree, int T, int64_t B, const int32_t *indices, const int32_t *actions,
                                   const float *rewards, const float *mu, const float *records, const int32_t *items,
                                   const int32_t *n_items, const double *norm, const rnad_learn_params_t *hp, void *accumulators,
                                   double *losses, float *dlogit_tab, float *dv_tab, void *stream) {
    RNAD_REQUIRE(tree && indices && actions && rewards && mu && records && items && n_items && hp && accumulators && dlogit_tab && dv_tab,
                 "rnad_learn_bucketed: null argument");
    RNAD_REQUIRE(T >= 1 && B >= 1, "rnad_learn_bucketed: bad shape");
    return learn_bucketed_impl(tree, T, B, indices, actions, rewards, mu, nullptr, nullptr, records, nullptr, items, n_items, norm, hp,
                               accumulators, losses, dlogit_tab, dv_tab, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr,
                               (hipStream_t)stream);
}

extern "C" int rnad_learn_bucketed_compact(const rnad_tree_t *tree, int T, int64_t B, const void *states, const uint64_t *acts,
                                           const float *final_reward, const float *fast_records, const float *records,
                                           const int32_t *items, const int32_t *n_items, const double *norm, const rnad_learn_params_t *hp,
                                           void *accumulators, double *losses, float *dlogit_tab, float *dv_tab, const int32_t *rows,
                                           const int64_t *n_rows, const void *rollout_scratch, int rollout_T_cap, int32_t *alive,
                                           double *norm_out, const rnad_row_groups_t *groups, void *stream) {
    RNAD_REQUIRE(!rows == !n_rows, "rnad_learn_bucketed_compact: rows and n_rows go together");
    RNAD_REQUIRE(!rollout_scratch || (alive && rollout_T_cap >= T && rollout_T_cap <= kCompactSteps),
                 "rnad_learn_bucketed_compact: completing the rollout's alive counts needs `alive` and the rollout's T_cap");
    const void *indices = states;
    RNAD_REQUIRE(tree && indices && acts && final_reward && fast_records && items && n_items && hp && accumulators && dlogit_tab && dv_tab,
                 "rnad_learn_bucketed_compact: null argument");
    RNAD_REQUIRE(!losses || records, "rnad_learn_bucketed_compact: the losses need the dense records (logits)");
    RNAD_REQUIRE(((uintptr_t)fast_records & 15) == 0, "rnad_learn_bucketed_compact: fast_records must be 16-byte aligned");
    RNAD_REQUIRE(T >= 1 && T <= kCompactSteps && B >= 1, "rnad_learn_bucketed_compact: bad shape");
    return learn_bucketed_impl(tree, T, B, indices, nullptr, nullptr, nullptr, (const unsigned long long *)acts, final_reward, records,
                               fast_records, items, n_items, norm, hp, accumulators, losses, dlogit_tab, dv_tab, rows, n_rows,
                               rollout_scratch, rollout_T_cap, alive, norm_out, groups, (hipStream_t)stream);
}
