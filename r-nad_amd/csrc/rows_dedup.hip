// rows_dedup.hip -- distinct observations among the (player, state) rows of a tree (gfx950).
//
// Everything a net contributes to a row of the tabular update is a function of the row's OBSERVATION alone (nn/net.py:37-51: the nets
// see nothing else; learn/rnad.py:373-382: logits, values, log_policy_reg; the legal bits are part of the observation), and two rows
// with the same observation therefore get the same logits, the same records, and add their gradients dL/dout into the same function of
// the weights.  On trees whose payoffs come from a small set the rows are far from distinct: on BASELINE configs[1] (terminal values
// +-1) the 59 049 states of the deepest level have expected-value matrices in {-1, +1}^9 -- 118 098 of the 132 862 rows carry one of 512
// observations, and the whole table holds ~15 k distinct ones.  The host groups the rows by the BITS of their observation
// (rnad_hip.TreeHandle.obs_dedup: rows that differ in a single bit, -0.0 vs +0.0 included, stay apart); then
//   * the table launch (rnad_mlp_rows_records) runs on one representative row per group and k_rows_expand copies the representatives'
//     records into the rows of their group -- bit for bit what the launch on all rows writes, since a row's results do not depend on
//     which other rows a launch evaluates (tests/test_hip_dedup.py);
//   * after k_bucket_finish, k_rows_segment_sum adds the rows' dL/dlogit, dL/dv of every group with more than one row into its
//     representative (a wave per group, lanes striding over its rows in ascending order, one fixed reduction tree: the same bits on
//     every run), and the backward runs on the representatives alone: sum_rows x_row (x) dz_row = sum_groups x_group (x) sum_rows dz_row,
//     the same weight gradient in another fp32 summation order.
// Citations are baskuit/R-NaD file:line.
#include "common.hpp"

using namespace rnad;

namespace {

constexpr int kThreads = 256;

struct ExpandTables {
    int n;
    float4 *tab[4];   // [rows][quads[i]] float4
    int quads[4];
};

// row r with rep_of[r] != r: tab[i][r] = tab[i][rep_of[r]] for every table.  One thread per (row, 16-byte chunk of the widest table).
__global__ __launch_bounds__(kThreads) void k_rows_expand(int64_t rows, const int32_t *__restrict__ rep_of, ExpandTables t, int max_quads) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t r = i / max_quads;
    const int q = (int)(i % max_quads);
    if (r >= rows) return;
    const int64_t rep = rep_of[r];
    if (rep == r) return;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < t.n && q < t.quads[k]) t.tab[k][r * t.quads[k] + q] = t.tab[k][rep * t.quads[k] + q];
}

// Group g (of the groups with more than one row): rows order[start[g] .. start[g + 1]) ascending, order[start[g]] its representative.
// dlogit[rep] / dv[rep] <- the sums over the group.  A wave per group: lane l adds rows l, l + 64, ... in that order, then the 64 partial
// sums go through one butterfly (a fixed tree: reproducible).
template <int A>
__global__ __launch_bounds__(kThreads) void k_rows_segment_sum(int n_groups, const int32_t *__restrict__ start, const int32_t *__restrict__ order,
                                                               float *__restrict__ dlogit, float *__restrict__ dv) {
    const int g = (int)blockIdx.x * (kThreads / 64) + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= n_groups) return;
    const int lo = start[g], hi = start[g + 1];
    float s[A + 1];
#pragma unroll
    for (int a = 0; a <= A; ++a) s[a] = 0.0f;
    for (int i = lo + lane; i < hi; i += 64) {
        const int64_t r = order[i];
#pragma unroll
        for (int a = 0; a < A; ++a) s[a] += dlogit[r * A + a];
        s[A] += dv[r];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int a = 0; a <= A; ++a) s[a] += __shfl_xor(s[a], off, 64);
    }
    if (lane == 0) {
        const int64_t rep = order[lo];
#pragma unroll
        for (int a = 0; a < A; ++a) dlogit[rep * A + a] = s[a];
        dv[rep] = s[A];
    }
}

}  // namespace

extern "C" int rnad_rows_expand(int64_t rows, const int32_t *rep_of, int n_tables, float *const *tables, const int32_t *floats_per_row,
                                void *stream) {
    RNAD_REQUIRE(rows >= 0 && rep_of && tables && floats_per_row && n_tables >= 1 && n_tables <= 4, "rnad_rows_expand: 1..4 tables");
    if (rows == 0) return 0;
    ExpandTables t{};
    t.n = n_tables;
    int max_quads = 0;
    for (int k = 0; k < n_tables; ++k) {
        RNAD_REQUIRE(tables[k] && floats_per_row[k] > 0 && floats_per_row[k] % 4 == 0 && ((uintptr_t)tables[k] & 15) == 0,
                     "rnad_rows_expand: table %d must be 16-byte aligned with a row of a multiple of 4 floats", k);
        t.tab[k] = reinterpret_cast<float4 *>(tables[k]);
        t.quads[k] = floats_per_row[k] / 4;
        max_quads = std::max(max_quads, t.quads[k]);
    }
    const int64_t n = rows * max_quads;
    hipLaunchKernelGGL(k_rows_expand, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, (hipStream_t)stream, rows, rep_of, t,
                       max_quads);
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int rnad_rows_segment_sum(int n_groups, const int32_t *start, const int32_t *order, int A, float *dlogit_tab, float *dv_tab,
                                     void *stream) {
    RNAD_REQUIRE(n_groups >= 0 && dlogit_tab && dv_tab, "rnad_rows_segment_sum: null argument");
    if (n_groups == 0) return 0;
    RNAD_REQUIRE(start && order, "rnad_rows_segment_sum: null argument");
    const unsigned grid = (unsigned)((n_groups + kThreads / 64 - 1) / (kThreads / 64));
    RNAD_DISPATCH_A(A, hipLaunchKernelGGL((k_rows_segment_sum<kA>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, n_groups, start, order,
                                          dlogit_tab, dv_tab));
    RNAD_HIP_OK(hipGetLastError());
    return 0;
}
