// nashconv.hip -- exploitability of a joint policy on the whole tree (gfx950).
//
// Replaces: util/metric.py:93-175 (NashConvData.get_nashconv), a Python recursion with one frame and ~20 tiny tensor ops
// per state, by two level-batched sweeps over the tree's depth levels: a top-down sweep that marks the sub-tree of
// `state_index` and propagates reach probabilities, and a bottom-up sweep that computes best-response values.  One lane
// per state; a level's states are independent.  Citations are baskuit/R-NaD file:line.
#include "common.hpp"

using namespace rnad;

namespace {

constexpr int kThreads = 256;
inline unsigned blocks_for(int64_t n) { return (unsigned)((n + kThreads - 1) / kThreads); }

template <int A>
__device__ __forceinline__ const float *policy_of(int s, int root, const float *__restrict__ table, const float *__restrict__ root_policy) {
    // the recursion hands self.joint_policy to every descendant (metric.py:148-151); only the starting state uses the argument
    return s == root ? root_policy : table + (int64_t)s * 2 * A;
}

// Top-down: states of one level that are marked pass the mark and their reach probability on to their children.
//   reach(child) = reach * joint_policy_matrix_flat[idx_flat] * transition_prob                (metric.py:152)
// where joint_policy_matrix_flat = flatten(pi_col pi_row^T) indexed by the flat (t, r, c) index (:130-132), i.e. it
// reads pi_col[r] * pi_row[c] -- kept as the reference computes it.
template <int A>
__global__ __launch_bounds__(kThreads) void k_reach(const Trans *__restrict__ trans, int C, const int32_t *__restrict__ order,
                                                    int64_t n, int root, const float *__restrict__ table,
                                                    const float *__restrict__ root_policy, uint8_t *__restrict__ mark,
                                                    float *__restrict__ reach) {
    const int64_t k = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (k >= n) return;
    const int s = order[k];
    if (!mark[s]) return;
    const float *pi = policy_of<A>(s, root, table, root_policy);
    const float rs = reach[s];
    const Trans *e = trans + (int64_t)s * A * A * C;
    for (int r = 0; r < A; ++r)
        for (int c = 0; c < A; ++c) {
            const float jp = pi[A + r] * pi[c];
            for (int t = 0; t < C; ++t) {
                const Trans x = e[(r * A + c) * C + t];
                if (x.chance > 0.0f && x.next != 0) {
                    mark[x.next] = 1;
                    reach[x.next] = rs * jp * x.chance;
                }
            }
        }
}

// Bottom-up (metric.py:134-175): best-response values of one level from the finished levels below it.
template <int A>
__global__ __launch_bounds__(kThreads) void k_best_response(const Trans *__restrict__ trans, const float *__restrict__ node, int C,
                                                            const int32_t *__restrict__ order, int64_t n, int root,
                                                            const float *__restrict__ table, const float *__restrict__ root_policy,
                                                            const uint8_t *__restrict__ mark, float *__restrict__ row_best,
                                                            float *__restrict__ col_best, int32_t *__restrict__ depth) {
    constexpr int AA = A * A, NS = (AA + 2 + 3) & ~3;
    const int64_t k = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (k >= n) return;
    const int s = order[k];
    if (!mark[s]) return;
    const float *pi = policy_of<A>(s, root, table, root_policy);
    const Trans *e = trans + (int64_t)s * AA * C;
    const float *nd = node + (int64_t)s * NS;
    const uint64_t bits = (uint64_t)__float_as_uint(nd[AA]) | ((uint64_t)__float_as_uint(nd[AA + 1]) << 32);
    float rowr[A], colr[A];  // row_responses[i] = sum_j M_row[i][j] pi_col[j]; col_responses[j] = sum_i pi_row[i] M_col[i][j]
#pragma unroll
    for (int a = 0; a < A; ++a) rowr[a] = colr[a] = 0.0f;
    int maxd = 0;
#pragma unroll
    for (int r = 0; r < A; ++r) {
#pragma unroll
        for (int c = 0; c < A; ++c) {
            float mr = 0.0f, mc = 0.0f;  // torch.sum over the chance dimension (:164-165)
            for (int t = 0; t < C; ++t) {
                const Trans x = e[(r * A + c) * C + t];
                if (x.chance > 0.0f) {
                    float rb, cb;
                    if (x.next == 0) {  // terminal: row_b, col_b = v, -v (:141-144)
                        rb = x.value;
                        cb = -x.value;
                    } else {
                        rb = row_best[x.next];
                        cb = col_best[x.next];
                        const int d = depth[x.next];
                        maxd = d > maxd ? d : maxd;
                    }
                    mr += rb * x.chance;  // :159-160
                    mc += cb * x.chance;
                }
            }
            rowr[r] += mr * pi[A + c];  // matmul(row_best_case_matrix, pi_col) (:169)
            colr[c] += pi[r] * mc;      // matmul(pi_row, col_best_case_matrix) (:170)
        }
    }
    float br = -INFINITY, bc = -INFINITY;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        if ((bits >> (a * A)) & 1) br = rowr[a] > br ? rowr[a] : br;  // legal_tensor[s, 0, :, 0] (:167)
        if ((bits >> a) & 1) bc = colr[a] > bc ? colr[a] : bc;        // legal_tensor[s, 0, 0, :] (:168)
    }
    row_best[s] = br;
    col_best[s] = bc;
    depth[s] = 1 + maxd;  // :175
}

__global__ void k_seed(int root, float reach0, uint8_t *mark, float *reach) {
    mark[root] = 1;
    reach[root] = reach0;
}

}  // namespace

extern "C" int rnad_nashconv(const rnad_tree_t *tree, const float *joint_policy, const float *root_policy, int64_t state_index,
                             float reach, float *row_best, float *col_best, float *reach_out, int32_t *depth_out, void *stream_) {
    RNAD_REQUIRE(tree && joint_policy && root_policy && row_best && col_best && reach_out && depth_out, "rnad_nashconv: null argument");
    RNAD_REQUIRE(state_index >= 1 && state_index < tree->S, "rnad_nashconv: state_index %lld outside [1,%lld)", (long long)state_index,
                 (long long)tree->S);
    const int top = tree->level_of[(size_t)state_index];
    RNAD_REQUIRE(top >= 0, "rnad_nashconv: state %lld is not reachable from the root", (long long)state_index);
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(tree->device);
    RNAD_REQUIRE(guard.ok, "rnad_nashconv: cannot select device %d", tree->device);
    RNAD_REQUIRE(tree->A >= 1 && tree->A <= RNAD_MAX_ACTIONS, "rnad_nashconv: bad tree");  // nothing below may return early: `mark` is owned here
    uint8_t *mark = nullptr;
    const size_t mark_bytes = ((size_t)tree->S + 3) & ~(size_t)3;  // zero_async clears whole words
    RNAD_HIP_OK(hipMallocAsync((void **)&mark, mark_bytes, stream));
    if (int rc = zero_async(mark, mark_bytes, stream)) {
        (void)hipFreeAsync(mark, stream);
        return rc;
    }
    hipLaunchKernelGGL(k_seed, dim3(1), dim3(1), 0, stream, (int)state_index, reach, mark, reach_out);
    const int root = (int)state_index;
    RNAD_DISPATCH_A(tree->A, {
        for (int l = top; l < tree->n_levels; ++l) {
            const int64_t off = tree->level_offsets[l], n = tree->level_offsets[l + 1] - off;
            if (n > 0)
                hipLaunchKernelGGL((k_reach<kA>), dim3(blocks_for(n)), dim3(kThreads), 0, stream, tree->trans, tree->C,
                                   tree->level_order + off, n, root, joint_policy, root_policy, mark, reach_out);
        }
        for (int l = tree->n_levels - 1; l >= top; --l) {
            const int64_t off = tree->level_offsets[l], n = tree->level_offsets[l + 1] - off;
            if (n > 0)
                hipLaunchKernelGGL((k_best_response<kA>), dim3(blocks_for(n)), dim3(kThreads), 0, stream, tree->trans, tree->node,
                                   tree->C, tree->level_order + off, n, root, joint_policy, root_policy, mark, row_best, col_best,
                                   depth_out);
        }
    });
    RNAD_HIP_OK(hipGetLastError());
    RNAD_HIP_OK(hipFreeAsync(mark, stream));
    return 0;
}
