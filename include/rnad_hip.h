/*
 * rnad_hip.h -- C-ABI of librnad_hip.so: the MI355X (gfx950) implementation of the baskuit/R-NaD
 * vectorised self-play hot path.
 *
 * The reference is pure Python/PyTorch and has NO FFI/operator boundary of its own (SURVEY.md
 * section 8b); the drop-in boundary is its Python class/function API, which r-nad_amd/ mirrors.
 * This header is the native boundary underneath: each entry point names the reference code
 * (file:line in baskuit/R-NaD) whose tensor program it replaces.  INTEGRATION.md shows the
 * ctypes stub a reference maintainer would add per call site.
 *
 * Conventions
 *   - plain C, no torch types: caller-owned DEVICE pointers (tensor.data_ptr()), explicit sizes;
 *     `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls only enqueue
 *     work and never synchronise unless stated.
 *   - every function returns 0 on success, non-zero on failure; rnad_last_error() then returns a
 *     message (thread-local).  Nothing here falls back to a CPU path.
 *   - B = episodes ("lanes") in the batch, T = env steps, A = max_actions, C = max_transitions,
 *     S = states in the tree.  Trajectory tensors are [T, B, ...] row-major like the reference's
 *     torch.stack'ed lists (environment/episode.py:218-225).
 *   - dtypes differ from the reference only where noted: state indices and actions are int32
 *     (reference int64), `turns` is implicit (t & 1, see episode.py:96-98), legal masks travel as
 *     one byte of bits per (t, b) next to the fp32 observation.
 *   - fp32 arithmetic follows the reference's operation order and is compiled with
 *     -ffp-contract=off; integer results are bit-exact, fp32 results within 1e-5 (exp/log use the
 *     device libm).
 *   - one host thread per GPU; handles are not thread-safe.  The library keeps NO mutable device state of its own: the tree
 *     handle's tables are read-only after rnad_tree_create / the first rnad_bucket_plan of a cut, and everything a call writes is a
 *     caller-owned buffer -- the `scratch` / `accumulators` workspaces of the bucketed pipeline, the ticket word of
 *     rnad_optimizer_step.  Two trainers on one device (the `for eta in ...` loop of main.py:55-81 run side by side, an evaluation
 *     beside a training run) are safe on different streams as long as each brings its OWN workspaces; calls that share a workspace
 *     must be ordered by a stream (r-nad_amd/rnad_hip gives every RNaD object its own: rnad_hip.workspace_owner).
 */
#ifndef RNAD_HIP_H
#define RNAD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden */

#define RNAD_MAX_ACTIONS 8
#define RNAD_MAX_TRANSITIONS 8

const char *rnad_last_error(void);
int rnad_version(void);

/* ------------------------------------------------------------------------------------------------
 * Tree tables.  Replaces the seven per-state tensors of environment/tree.py:115-146 as the thing
 * the kernels read.  Inputs are HOST pointers in the reference layout
 *   index int64 [S,C,A,A], value/chance f32 [S,C,A,A], expected_value/legal f32 [S,1,A,A]
 * and are re-packed for the GPU (DESIGN.md "HBM layout"):
 *   node  [S][NS]        f32: expected_value[A*A], legal bitmask (bit i*A+j), pad to 16 B
 *   trans [S][A][A][C]   {int32 next, f32 chance, f32 value}   (12*C contiguous bytes per joint action)
 * ---------------------------------------------------------------------------------------------- */
typedef struct rnad_tree rnad_tree_t;

int rnad_tree_create(rnad_tree_t **out, int64_t S, int C, int A, const int64_t *index, const float *value,
                     const float *chance, const float *expected_value, const float *legal, int device);
void rnad_tree_destroy(rnad_tree_t *tree);
/* size queries: which = 0:S 1:C 2:A 3:max game depth (longest root->terminal chain of transitions)
 * 4:node stride in floats 5:bytes of device tables 6:1 if every episode has the same length (the tree is left only from
 * its deepest level: nothing for the live-row lists below to skip) */
int64_t rnad_tree_info(const rnad_tree_t *tree, int which);

/* ------------------------------------------------------------------------------------------------
 * K1  observe  --  environment/episode.py:62-68 (States.observations) + :208 (masks).
 *   obs[b,0,i,j] = player ?  -ev[s,j,i] : ev[s,i,j]      obs[b,1,i,j] = player ? legal[s,j,i] : legal[s,i,j]
 * `player` is uniform over the batch (episode.py:96-98).  obs is fp32 [B,2,A,A] (obs_half != 0:
 * fp16).  mask_bits (u8 [B], bit i = row i legal for the mover) and mask (f32 [B,A]) are optional.
 * ---------------------------------------------------------------------------------------------- */
int rnad_observe(const rnad_tree_t *tree, int64_t B, const int32_t *idx, int player, void *obs, int obs_half,
                 uint8_t *mask_bits, float *mask, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Policy head  --  nn/net.py:45-46 (forward) and :74-77 (forward_batch):
 *   policy = where(legal, exp(logits), 0) / max(sum, 1e-12);  log_policy = where(legal, logits - log(sum), 0)
 * legality comes from mask_bits (u8 [N]) or, if that is NULL, from mask (f32 [N,A], non-zero = legal).
 * ---------------------------------------------------------------------------------------------- */
int rnad_policy_head(int64_t N, int A, const float *logits, const uint8_t *mask_bits, const float *mask,
                     float *policy, float *log_policy, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Fused MLP  --  nn/net.py:40-43 (and :70-73): value = value_fc1(relu(value_fc0 x)),
 * logits = policy_fc1(relu(policy_fc0 x)), x = obs flattened to 2*A*A floats (fp32, or fp16 when
 * obs_half != 0).  The hidden layer stays in registers (fp32 MFMA: exact fp32 products and sums;
 * the summation order differs from a BLAS GEMM, results agree to ~1e-6 relative).
 *
 * rnad_mlp_pack lays the eight torch Linear tensors ([out, in] row-major fp32: vw0 [W,2A^2], vb0 [W],
 * vw1 [1,W], vb1 [1], pw0 [W,2A^2], pb0 [W], pw1 [A,W], pb1 [A]) out as the `packed` image of
 * rnad_mlp_packed_size(A, W) floats that the kernels copy into LDS with one coalesced pass; pack once
 * per weight update, reuse for every forward/backward with those weights.  W: multiple of 32.
 *
 * rnad_mlp_forward: logits [N,A] and/or value [N]; a NULL output means that head is not computed.
 * rnad_mlp_backward: gradients of the eight tensors given dL/dlogits [N,A] and dL/dvalue [N] (what
 * loss.backward() computes for the learner net, learn/rnad.py:425), hidden layer recomputed on chip,
 * weight gradients contracted over samples on the matrix cores.  g_* are written (not accumulated),
 * torch Linear layouts.  workspace: rnad_mlp_backward_workspace(N, A, W) bytes (per-block partials,
 * summed in a fixed order; -1 if the weight image does not fit the 160 KiB LDS).
 * ---------------------------------------------------------------------------------------------- */
int64_t rnad_mlp_packed_size(int A, int W);
int rnad_mlp_pack(int A, int W, const float *vw0, const float *vb0, const float *vw1, const float *vb1, const float *pw0,
                  const float *pb0, const float *pw1, const float *pb1, float *packed, void *stream);
/* n_nets (1..4) weight images in one launch: weights = HOST array of 8 * n_nets device pointers (net i: entries 8i .. 8i + 7 in the
 * order of rnad_mlp_pack), packed = HOST array of n_nets output images. */
int rnad_mlp_pack_multi(int n_nets, int A, int W, const float *const *weights, float *const *packed, void *stream);
int rnad_mlp_forward(int64_t N, int A, int W, const float *packed, const void *obs, int obs_half, float *logits,
                     float *value, void *stream);
/* n_nets (1..4) nets of the same shape on the same N inputs in one launch; packed / logits / value are HOST arrays of n_nets
 * device pointers (a NULL logits[i] or value[i]: that output is not stored).  What the tabular update uses for the learner,
 * target and two regularisation nets (learn/rnad.py:373-380) on the 2S observations of the tree. */
int rnad_mlp_forward_multi(int n_nets, int64_t N, int A, int W, const float *const *packed, const void *obs, int obs_half,
                           float *const *logits, float *const *value, void *stream);
int64_t rnad_mlp_backward_workspace(int64_t N, int A, int W);
int rnad_mlp_backward(int64_t N, int A, int W, const float *packed, const void *obs, int obs_half, const float *dlogits,
                      const float *dvalue, float *g_vw0, float *g_vb0, float *g_vw1, float *g_vb1, float *g_pw0,
                      float *g_pb0, float *g_pw1, float *g_pb1, float *workspace, void *stream);

/* Ragged trees: once an episode is absorbed (state 0) the reference keeps evaluating the nets on it (episode.py:194-212,
 * net.py:64-85 run on all of [T, B]) and masks the results (`valid`, rnad.py:369).  The *_rows variants evaluate only the
 * rows listed: sample s of the launch is row rows[s] of obs / logits / value / dlogits / dvalue; rows that are not listed
 * are neither read nor written.  The row count is read from device memory (*n_rows <= max_rows), so the list can be built
 * on the same stream with no host round trip.
 * rnad_compact_valid: rows = the positions r in [0, N) with indices[r] != 0, ascending, *n_rows = how many;
 * block_counts: scratch of rnad_compact_workspace(N) int32. */
int64_t rnad_compact_workspace(int64_t N);
int rnad_compact_valid(int64_t N, const int32_t *indices, int32_t *rows, int64_t *n_rows, int32_t *block_counts, void *stream);
int rnad_mlp_forward_rows(int64_t max_rows, const int32_t *rows, const int64_t *n_rows, int A, int W, const float *packed,
                          const void *obs, int obs_half, float *logits, float *value, void *stream);
int rnad_mlp_backward_rows(int64_t max_rows, const int32_t *rows, const int64_t *n_rows, int A, int W, const float *packed,
                           const void *obs, int obs_half, const float *dlogits, const float *dvalue, float *g_vw0,
                           float *g_vb0, float *g_vw1, float *g_vb1, float *g_pw0, float *g_pb0, float *g_pw1, float *g_pb1,
                           float *workspace, void *stream);

/* The legal fold.  An observation is [expected value A x A | legal mask A x A] (episode.py:62-68); on a tree whose states all have the
 * full A x A action set -- every configuration of BASELINE.json -- the legal plane is all ones in every row except the two rows of
 * the absorbing state, where it is e0 = [1, 0, ..., 0] (tree.py:133).  The first layer of nn/net.py:40-43 is then
 *     z = W_ev ev + (b0 + W_legal 1) + [absorbing row] (W_legal e0 - W_legal 1)
 * -- A^2 + 1 input features instead of 2 A^2 (5 MFMA k-steps instead of 9 at A = 3): the same function of the same weights, summed
 * in another order.  These entry points are the fused MLP kernels instantiated for it; observations keep their layout (row stride
 * 2 A^2; the kernels read the expected values and legal[0][1]) and the CALLER guarantees the premise for every row it passes
 * (rnad_hip.TreeHandle checks its observation table once).  A >= 2.  The packed image (rnad_mlp_fold_packed_size floats) keeps the
 * raw weights, so rnad_optimizer_step(mlp_fold = 1) maintains it element by element; the kernels fold when they load it.
 * rnad_mlp_forward_fold: rnad_mlp_forward_multi / _rows in one (rows, n_rows: NULL or a row list, N its capacity).
 * rnad_mlp_backward_fold: the gradients of the eight ORIGINAL tensors (exactly recoverable: dW_legal[h][0] = db0[h], dW_legal[h][j >= 1]
 * = db0[h] - the indicator column's gradient); workspace: rnad_mlp_backward_workspace(N, A, W) bytes. */
int64_t rnad_mlp_fold_packed_size(int A, int W);
int rnad_mlp_pack_fold_multi(int n_nets, int A, int W, const float *const *weights, float *const *packed, void *stream);
int rnad_mlp_forward_fold(int n_nets, int64_t N, const int32_t *rows, const int64_t *n_rows, int A, int W, const float *const *packed,
                          const void *obs, int obs_half, float *const *logits, float *const *value, void *stream);
/* A tabular ACTOR on (a row list of) the tree's 2S observations: the policy head of one net -> logits [2S, A] and, from the kernel's
 * epilogue, its policy rows [2S, rnad_bucket_policy_row_stride(A)] (net.py:45-46 under the mover's legal bits): the table
 * rnad_bucket_sort / rnad_bucket_play gather from (table_is_policy = 1) without a policy-head launch of their own.  fold: FOLD image. */
int rnad_mlp_forward_actor(const rnad_tree_t *tree, const int32_t *rows, const int64_t *n_rows, int W, int fold, const float *packed,
                           const void *obs, int obs_half, float *logits, float *policy_rows, void *stream);
int rnad_mlp_backward_fold(int64_t N, const int32_t *rows, const int64_t *n_rows, int A, int W, const float *packed, const void *obs,
                           int obs_half, const float *dlogits, const float *dvalue, float *g_vw0, float *g_vb0, float *g_vw1, float *g_vb1,
                           float *g_pw0, float *g_pb0, float *g_pw1, float *g_pb1, float *workspace, void *stream);

/* ------------------------------------------------------------------------------------------------
 * K3  sample  --  torch.multinomial(policy, 1) at nn/net.py:49: one draw from Cat(probs[b]) per lane.
 * noise (f32 [B,n]) given: torch's own n_sample == 1 algorithm on it, argmax_a(probs[a] / q[a]) with
 * q = noise ~ Exp(1), first maximum wins -- with the q the reference consumed, the reference's draw.
 * noise NULL: the SEEDED draw of include/rnad_rng.h -- the inverse CDF of the counter-based uniform of
 * decision (seed, lane0 + b, step): stream_id 0 = the action of env step `step`, 1 = the chance
 * draw of the game transition `step` belongs to.  Same distribution, one generator call per
 * transition, no logarithms or divisions.
 * ---------------------------------------------------------------------------------------------- */
int rnad_sample(int64_t B, int n, const float *probs, const float *noise, uint64_t seed, int64_t lane0, int step,
                int stream_id, int32_t *out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * K2  transition  --  environment/episode.py:102-123 (States.step, column branch):
 *   t ~ multinomial(chance[s,:,r,c]);  s' = index[s,t,r,c];  reward = value[s,t,r,c] * (s' == 0)
 * noise: f32 [B,C] Exp(1) noise for torch's race, or NULL: the seeded chance draw of env step `step`
 * (include/rnad_rng.h).  alive (optional, int32[1]) is incremented by the
 * number of lanes with s' != 0 (replaces the host sync of episode.py:124).
 * ---------------------------------------------------------------------------------------------- */
int rnad_transition(const rnad_tree_t *tree, int64_t B, const int32_t *idx, const int32_t *row_actions,
                    const int32_t *col_actions, const float *noise, uint64_t seed, int64_t lane0, int step,
                    int32_t *idx_out, float *reward, int32_t *alive, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Rollout driver  --  environment/episode.py:175-230 (Episodes.generate) without the per-step
 * clones, list appends and host syncs.  The caller owns preallocated [T_cap, B, ...] buffers:
 * ---------------------------------------------------------------------------------------------- */
typedef struct rnad_traj {
    int32_t T_cap;        /* capacity in env steps (>= 2 * max depth)                               */
    int32_t obs_half;     /* 0: observations fp32, 1: fp16                                          */
    int64_t B;            /* lanes                                                                   */
    int32_t *indices;     /* [T_cap + 1, B]  state id at the START of step t (row T_cap: after last) */
    void *observations;   /* [T_cap, B, 2, A, A]                                                     */
    uint8_t *mask_bits;   /* [T_cap, B]                                                              */
    float *policy;        /* [T_cap, B, A]   acting policy mu                                        */
    int32_t *actions;     /* [T_cap, B]      sampled action id                                       */
    float *rewards;       /* [T_cap, B]      row-player reward (episode.py:120-121)                  */
    float *values;        /* [T_cap, B]      actor value head (episode.py:207)                       */
    int32_t *alive;       /* [T_cap + 1]     #lanes with indices[t] != 0                             */
} rnad_traj_t;

/* indices[0,:] = 1 (root, episode.py:22), then K1 for t = 0. */
int rnad_rollout_begin(const rnad_tree_t *tree, const rnad_traj_t *traj, void *stream);

/* One env step t given the net outputs for observations[t] (nn/net.py:37-51):
 *   mode 0: `logits` [B,A] given      -> policy head, sample, record            (fast path)
 *   mode 1: `policy_in` [B,A] given   -> sample, record
 *   mode 2: `policy_in` and `actions_in` (int32 [B]) given -> record only       (generic nets that
 *           sample for themselves, net.py:49)
 * then (odd t) the transition of K2 with row action actions[t-1] and column action actions[t],
 * (even t) indices[t+1] = indices[t], rewards[t] = 0 (episode.py:99-101); and, if t + 1 < T_cap, K1 for
 * step t + 1 as a second launch.  The alive counters are NOT touched per step (one contended atomic word
 * caps at ~90 updates/us on gfx950): rnad_rollout_end counts them once.
 * noise_action [B,A] / noise_chance [B,C]: explicit Exp(1) noise (torch's race) or NULL for the seeded draws
 * (include/rnad_rng.h: the uniforms of lane lane0 + b at env step t, inverse CDF). */
int rnad_rollout_step(const rnad_tree_t *tree, const rnad_traj_t *traj, int t, int mode, const float *logits,
                      const float *policy_in, const int32_t *actions_in, const float *value,
                      const float *noise_action, const float *noise_chance, uint64_t seed, int64_t lane0,
                      void *stream);

/* The whole loop of Episodes.generate for a net that is the fused MLP of this library (nn/net.py:18-51): begin, then
 * for every t: rnad_mlp_forward(observations[t]) -> rnad_rollout_step(mode 0), then rnad_rollout_end -- enqueued from ONE
 * call (3 * T_cap + 2 launches, no host work in between).  packed: rnad_mlp_pack image; value_ws [B] scratch, or NULL:
 * the actor's value head is then not evaluated and traj->values is filled with zeros (the reference stores the actor's
 * values, episode.py:206,218, and never reads them: learn/rnad.py:373 recomputes v from the learner net);
 * logits_ws: [B,A] scratch with logits_step_stride = 0, or a [T_cap,B,A] buffer with logits_step_stride = B*A that
 * keeps the actor's raw logits of every step.  Seeded draws only.
 * live_rows [B] int32 / n_live [1] / block_counts [rnad_compact_workspace(B)]: all NULL = the actor runs on every lane at
 * every step, as the reference does; all given (logits_step_stride must be 0) = from step 1 on it runs only on the lanes
 * still in the tree (rnad_compact_valid + rnad_mlp_forward_rows); absorbed lanes then carry the logits / value of their
 * last live step instead of the net's output on state 0 -- masked everywhere downstream. */
int rnad_rollout_run(const rnad_tree_t *tree, const rnad_traj_t *traj, int W, const float *packed, float *logits_ws,
                     int64_t logits_step_stride, float *value_ws, uint64_t seed, int64_t lane0, int32_t *live_rows,
                     int64_t *n_live, int32_t *block_counts, void *stream);

/* Tabular actor.  A lane's observation is a function of (state, player to move) alone (episode.py:62-68), so an actor with
 * fixed weights can be evaluated once on the 2S distinct observations -- rnad_observe_all, then rnad_mlp_forward with
 * N = 2S -- instead of on B lanes at each of the T steps.  logits_table: [2, S, A], row = player * S + state.  The rollout
 * (policies, actions, indices, rewards, observations) is that of rnad_rollout_run bit for bit; value_table [2, S] (the actor's
 * value head on the same rows) is gathered into traj->values, or NULL: zeros. */
int rnad_rollout_run_tabular(const rnad_tree_t *tree, const rnad_traj_t *traj, const float *logits_table,
                             const float *value_table, uint64_t seed, int64_t lane0, void *stream);

/* alive[t] = #lanes with indices[t, :] != 0 for t in [0, T_cap]: one pass over the index buffer after the last
 * step.  The host reads it once to trim the trajectory to the reference's T (episode.py:194 stops when every lane
 * is absorbed) and to get the loss normalisers N_P = sum over t == P (mod 2) of alive[t]. */
int rnad_rollout_end(const rnad_tree_t *tree, const rnad_traj_t *traj, void *stream);

/* ------------------------------------------------------------------------------------------------
 * K4  process_policy  --  learn/vtrace.py:24-55.   policy, mask f32 [N,A] -> out f32 [N,A].
 * ---------------------------------------------------------------------------------------------- */
int rnad_process_policy(int64_t N, int A, const float *policy, const float *mask, int n_disc, float eps,
                        float *out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * K5  v_trace  --  learn/vtrace.py:207-352 for `player`, incl. _has_played (:141-177),
 * _policy_ratio (:180-204), _player_others (:70-87).  One lane per episode, backward in time.
 *   v, valid, reward f32 [T,B]; player_id int32 [T,B] or NULL (= t & 1); mu (acting), pi (merged),
 *   logpi (merged log policy) f32 [T,B,A]; actions: one-hot f32 [T,B,A] (actions_onehot != 0) or
 *   int32 [T,B].  Outputs v_target f32 [T,B], q (= learning_output) f32 [T,B,A], has_played int32 [T,B]
 *   (optional).
 * ---------------------------------------------------------------------------------------------- */
int rnad_vtrace(int T, int64_t B, int A, const float *v, const float *valid, const int32_t *player_id,
                const float *mu, const float *pi, const float *logpi, const void *actions, int actions_onehot,
                const float *reward, int player, float eta, float lambda_, float c, float rho, float gamma,
                float *v_target, int32_t *has_played, float *q, void *stream);

/* ------------------------------------------------------------------------------------------------
 * K6  losses + closed-form gradients, one list entry (player) per call like the reference's loops
 * --  learn/vtrace.py:377-393 (get_loss_v) and :396-431 (get_loss_nerd with
 * apply_force_with_threshold :355-367 and renormalize :370-374).
 *   rnad_mask_sum: out[0] (device f64) = sum(mask), the `normalization` of :373/:388.  A
 *   data-parallel caller all-reduces it over ranks before passing it on as `norm`.
 *   rnad_loss_v:   L = sum(mask * (v - v_target)^2) / max(norm, 1);  dv = weight * 2 mask (v - v_target) / norm
 *   rnad_loss_nerd: adv = clip(q - sum_a(pi q)); l = logit - mean_a(logit * legal);
 *                   f = (l > -thr) min(adv, 0) + (l < thr) max(adv, 0);  L = -sum(mask * sum_a legal l f) / norm;
 *                   dlogit = -weight * mask (w - legal * sum_a(w) / A) / norm,  w = legal * f  (f is detached)
 *   v, v_target, mask f32 [N]; logit, pi, q, legal f32 [N,A]; norm device f64[1].
 *   loss (device f64[1], optional) receives += weight-free LOCAL sum / norm.  accumulate == 0:
 *   gradients are written, != 0: added to (second player of the list).
 * ---------------------------------------------------------------------------------------------- */
int rnad_mask_sum(int64_t N, const float *mask, double *out, void *stream);
int rnad_loss_v(int64_t N, const float *v, const float *v_target, const float *mask, const double *norm, float weight,
                double *loss, float *dv, int accumulate, void *stream);
int rnad_loss_nerd(int64_t N, int A, const float *logit, const float *pi, const float *q, const float *mask,
                   const float *legal, const double *norm, float clip, float threshold, float weight, double *loss,
                   float *dlogit, int accumulate, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Fused learner post-processing  --  the tensor program of RNaD.__learn, learn/rnad.py:365-425,
 * between the four forward_batch calls and loss.backward(): policy heads of the learner and the
 * two regularisation nets, log_policy_reg (:382), process_policy (:374), both players' v_trace
 * (:384-406) and the loss gradients (:407-425), in ONE backward-in-time pass per lane.  Nothing
 * but dlogit/dv (and optional logging tensors) is written to HBM.
 *   inputs [T,B(,A)]: indices (valid = != 0, :369), mask_bits, actions, rewards, mu = acting policy,
 *   logit / v of the learner, v_target_net of the target net, logit_reg / logit_reg_ of the
 *   regularisation nets; alpha (:497); norm = device f64[2], #(valid & t&1 == P) for P = 0, 1
 *   (all-reduced over ranks by a data-parallel caller).
 *   outputs: dlogit [T,B,A], dv [T,B]; losses f64[2] partial sums / norm; optional
 *   pi_out, v_target_out [2,T,B], q_out [2,T,B,A] for logging and tests (may be NULL).
 * ---------------------------------------------------------------------------------------------- */
typedef struct rnad_learn_params {
    float alpha, one_minus_alpha;               /* rnad.py:382,:497; 1 - alpha is rounded from the double */
    float eta, lambda_, c, rho, gamma;          /* rnad.py:397-401 */
    float clip, threshold;                      /* neurd_clip, logit_clip (beta), rnad.py:420-421 */
    float w_v, w_n;                             /* value_loss_weight, neurd_loss_weight, rnad.py:424 */
    float eps_threshold;                        /* process_policy epsilon, rnad.py:374 */
    int32_t n_disc;
} rnad_learn_params_t;

int rnad_learn_fused(int T, int64_t B, int A, const int32_t *indices, const uint8_t *mask_bits,
                     const int32_t *actions, const float *rewards, const float *mu, const float *logit,
                     const float *v, const float *v_target_net, const float *logit_reg, const float *logit_reg_,
                     const double *norm, const rnad_learn_params_t *hp, double *losses, float *dlogit, float *dv,
                     float *pi_out, float *v_target_out, float *q_out, void *stream);

/* Tabular variants.  The four forward_batch calls of learn/rnad.py:373-380 evaluate nets on observations that depend on
 * (state, player to move) only: 2S distinct inputs for T*B slots (132 862 vs 12.6 M on configs[1]).  Here the nets were
 * evaluated once per distinct input -- tables [2, S, (A)], row = player * S + state (rnad_observe_all + rnad_mlp_forward) --
 * and each slot gathers its row: per-slot net outputs are the same bits as in rnad_learn_fused (same inputs, same kernel),
 * hence the same V-trace targets, losses and per-slot dL/dlogit, dL/dv.
 *
 * rnad_learn_fused_tabular additionally sums those per-slot gradients per row: the weight gradient is linear in dL/dout, so
 * ONE rnad_mlp_backward over the 2S inputs with dlogit_tab [2S,A], dv_tab [2S] gives the gradients of the per-slot backward
 * up to fp32 summation order.  The sums are taken in 64-bit fixed point (integer atomics: any order, same result; the states
 * of the top levels, which the whole batch passes through, first in a per-block LDS table), so they are reproducible.
 * workspace: rnad_learn_tabular_workspace(tree, T, B) bytes, 16-byte aligned.  B <= 2^21 per call. */
int64_t rnad_learn_tabular_workspace(const rnad_tree_t *tree, int T, int64_t B);
int rnad_learn_fused_tabular(const rnad_tree_t *tree, int T, int64_t B, const int32_t *indices, const uint8_t *mask_bits,
                             const int32_t *actions, const float *rewards, const float *mu, const float *logit_tab,
                             const float *v_tab, const float *v_target_tab, const float *logit_reg_tab,
                             const float *logit_reg_tab_, const double *norm, const rnad_learn_params_t *hp, double *losses,
                             void *workspace, float *dlogit_tab, float *dv_tab, void *stream);
/* The per-row sums alone, for per-slot gradients that were computed elsewhere (e.g. by autograd behind a table gather in
 * nn/net.py:64-85 forward_batch): dlogit_tab[row] = sum over the slots of that row of dlogit, dv_tab likewise, slots with
 * indices == 0 skipped, in the same reproducible fixed-point arithmetic.  workspace: rnad_row_sums_workspace(tree) bytes. */
int64_t rnad_row_sums_workspace(const rnad_tree_t *tree);
int rnad_row_sums(const rnad_tree_t *tree, int T, int64_t B, const int32_t *indices, const float *dlogit, const float *dv,
                  void *workspace, float *dlogit_tab, float *dv_tab, void *stream);

/* Same gathers, but dL/dlogit [T,B,A] and dL/dv [T,B] are written per slot (the bits of rnad_learn_fused): only the forward
 * evaluations are deduplicated, and a per-slot rnad_mlp_backward then gives bit-identical, reproducible weight gradients.
 * workspace: rnad_learn_gather_workspace(tree) bytes, 16-byte aligned (the five tables interleaved into one record per row,
 * so that a slot gathers 48 contiguous bytes at A = 3 instead of five scattered pieces). */
int64_t rnad_learn_gather_workspace(const rnad_tree_t *tree);
int rnad_learn_fused_gather(const rnad_tree_t *tree, int T, int64_t B, const int32_t *indices, const uint8_t *mask_bits,
                            const int32_t *actions, const float *rewards, const float *mu, const float *logit_tab,
                            const float *v_tab, const float *v_target_tab, const float *logit_reg_tab,
                            const float *logit_reg_tab_, const double *norm, const rnad_learn_params_t *hp, double *losses,
                            void *workspace, float *dlogit, float *dv, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Bucketed tabular pipeline (csrc/bucket.hip)  --  environment/episode.py:175-230 (Episodes.generate) and learn/rnad.py:365-425
 * again, for trees that are small next to the batch, organised so that the per-(player, state) sums of the tabular update need
 * no global atomics.  State ids are DFS pre-order (tree.py:311-330), so the subtree of s is the id range [s, s + size(s)).  The
 * tree is cut by subtree size: states whose subtree exceeds R rows are "upper"; the other children of an upper state, packed into
 * runs of consecutive siblings spanning at most R ids, are the "groups".  Lanes are sorted by the group they descend into
 * (bucket), the trajectory buffers are written in that order, and one workgroup per <= 256 lanes of a bucket adds its lanes'
 * gradients up in an LDS table indexed by (state - first id of the group); the upper rows above it are common to the workgroup.
 *
 * rnad_bucket_plan: out[0] = R (rows of a bucket's table), [1] = number of buckets (groups + one terminal bucket per upper state),
 * [2] = number of upper states, [3] = number of groups, [4] = capacity of the work-item list (items: int32 [out[4]][4] = first lane,
 * lanes, bucket, 1 if the bucket's only item), [5] = bytes of `scratch` for rnad_rollout_bucketed, [6] = bytes of `accumulators` for
 * rnad_learn_bucketed (zero them once; every update leaves them zero), [7] = LDS bytes of a learner workgroup, [8] = bytes of a relative
 * state of the compact trajectory (1 or 2), [9] = lanes per workgroup of the sort passes, [10] = lanes per work item (out must hold 11
 * values).  The last two follow the tuning knobs RNAD_SORT_TILE / RNAD_BUCKET_CHUNK, which size out[5] / out[4]: the entry points that
 * write `scratch` fail (status 1) when the knobs differ from every combination rnad_bucket_plan was asked the sizes of.  Non-zero return: this
 * tree / batch cannot be bucketed (use the entry points above).  rnad_bucket_map: the bucket of every state (host int32 [S]; < *n_groups:
 * a group, else n_groups + upper slot; -1: state 0 / unreachable) -- what tests and tools need to reproduce the lane order.
 *
 * rnad_rollout_bucketed: the rollout of rnad_rollout_run_tabular -- same tabular actor (table row = player * S + state,
 * table_stride floats apart; table_is_policy == 0: the actor's logits, whose policy head is then taken once per row;
 * != 0: the actor's policy rows, e.g. the pi columns of rnad_bucket_records), same seeded draws keyed by the GLOBAL lane
 * id lane0 + lane, hence the same episodes bit for bit -- with column j of every [T_cap, B] buffer holding lane lane_ids[j]
 * (a stable sort of the lanes by bucket).  traj->observations is not written (may be NULL; an observation is a function of
 * (t & 1, indices[t]): rnad_observe), traj->values only if non-NULL (value_table NULL: zeros).  items / n_items: the learner's
 * work list.  norm (optional, device f64[2]): the loss normalisers N_P = number of live slots of parity P (learn/vtrace.py:373,388).
 * T_cap <= 64, B <= 2^22.
 *
 * rnad_bucket_records: one record per (player, state) row from the five [2S, .] net-output tables of a tabular update, holding
 * everything of learn/rnad.py:373-382 that depends on the row alone (rnad_bucket_record_stride(A) floats, 16-byte aligned):
 *   logit[A] | v | v_target | process_policy(pi)[A] | log_policy_reg[A] | legal bits | pi[A] | pad
 * with pi, log_pi = the learner's policy head (net.py:74-77), log_policy_reg = log_pi - (alpha log_pi_reg + (1 - alpha) log_pi_reg_).
 * policy_rows (optional output, rnad_bucket_policy_row_stride(A) floats per row, 16-byte aligned): the pi columns once more, as a table
 * of their own -- 16 bytes per row at A <= 4, 2 MB on configs[1]: what the rollout kernels should gather the actor from (it stays in
 * an XCD's L2 where the 64-byte records do not).
 *
 * rnad_learn_bucketed: rnad_learn_fused_tabular on a bucket-ordered trajectory and those records: dlogit_tab [2S, A], dv_tab [2S] = per-row sums
 * of the per-slot gradients, accumulated in 64-bit fixed point with an a-priori scale (|dL/dlogit| <= 2 * clip / N_P by
 * construction; |v - v_target| < 2^10 is checked, the tables are NaN if it fails), normalised by norm (f64[2], batch-global
 * N_P) at the end.  Integer sums: reproducible bit for bit.  losses (f64[2], optional): loss_v, loss_nerd of this rank's slots.
 *
 * Compact trajectory (on-policy updates: the batch is learned from in the step that played it, learn/rnad.py:502-510 with the
 * default one-batch buffer).  Of a slot's record only the state cannot be recomputed: the mask and the acting policy are rows of
 * tables, the action takes 3 bits, and rewards *= (indices == 0) (episode.py:120-121) leaves one non-zero reward per episode.
 * rnad_rollout_bucketed_compact plays the same episodes as rnad_rollout_bucketed (same table arguments: e.g. the pi columns of
 * `records`: table = records + 3A + 3, table_stride = rnad_bucket_record_stride(A), table_is_policy = 1) and
 * writes states [T_cap + 1, B], alive, acts (uint64 [B]: action of step t in bits 3t .. 3t + 2) and final_reward (f32 [B]); T_cap <= 21.
 * `states` holds RELATIVE states, rnad_bucket_plan's out[8] bytes each (1 while the cut's tables have <= 255 rows, else 2): column j
 * belongs to a work item of some bucket b (items: {begin, count, bucket, single}); a lane of b sits in the bucket's own path state at
 * every env step above its group (and at the root of a group that is a single subtree) -- those rows of the column are neither
 * written nor read -- and below in bucket_lo[b] + (value - 1), value 0 = absorbed.  25 instead of 300 bytes per lane at A = 3,
 * T = 12 on configs[1] (5 state bytes + actions + reward + lane id).  rnad_bucket_indices rebuilds the reference's indices
 * (int32 [T1, B], episode.py:218) from it, rnad_bucket_pack_states is the inverse (a recorded bucket-ordered trajectory into the
 * compact layout; *mismatch is set when a column is not a lane of its item's bucket).  rnad_learn_bucketed_compact is rnad_learn_bucketed on that
 * trajectory with the actor's own pi as the acting policy (the very floats the rollout sampled from), so that every operand of a
 * slot's V-trace / NeuRD arithmetic except the carries is the ROW's: rnad_bucket_records writes them once per row into
 * fast_records (optional output; rnad_bucket_fast_record_stride(A) = 4 + 4A floats:
 *   v | v_target | -eta sum(pi_processed log_policy_reg) | legal and threshold-gate bits | pi_processed[A] | -eta log_policy_reg[A] |
 *   pi_processed[A] / pi[A] | 1 / pi[A])
 * with the operations learn/vtrace.py applies per slot, and the learner does the rest: same gradients bit for bit.  `records` is
 * only read when `losses` is asked for (the logits).  rnad_bucket_expand writes the dense [T, B] buffers of such a trajectory (mask_bits, policy, actions, rewards) when
 * something asks for them; slots of absorbed lanes get action 0 (the dense rollout keeps drawing there; nothing reads them).
 *
 * Lazy rows (trees that are large next to the batch: configs[3] has 1.9 M rows of which a 2^20-lane batch visits 5 %).  The compact
 * rollout only needs the actor's policy, so: the learner's POLICY head on all 2S rows (table = its logits, table_is_policy = 0: the
 * policy head is taken once per row into scratch, or table = policy rows) -> rnad_rollout_bucketed_compact with `visited` (int32
 * [2S], optional: set to 1 for every (player, state) row a live slot sits in, plus the two rows of the absorbing state; cleared
 * first) -> rnad_compact_valid(2S, visited, rows, n_rows) -> the value heads on the listed rows (rnad_mlp_forward_rows) ->
 * rnad_bucket_records / rnad_learn_bucketed_compact / rnad_bucket_finish with that row list (rows, n_rows: device memory, both NULL =
 * all rows; rows that are not listed are neither read nor written, and their accumulators are zero) -> rnad_mlp_backward_rows.
 *
 * alive == NULL in rnad_rollout_bucketed_compact: the per-workgroup alive counts stay un-summed in `scratch` (one launch less); either
 * the learner adds them up -- rnad_learn_bucketed_compact(..., rollout_scratch = that scratch, rollout_T_cap, alive, norm_out), whose first
 * T_cap + 1 workgroups then write alive [T_cap + 1] and add N_P into norm_out (cleared by the rollout) before k_bucket_finish reads it --
 * or rnad_bucket_alive does, on its own (a data-parallel caller that all-reduces N_P beside the learner kernel; anything that reads
 * `alive` before the update).  rollout_scratch == NULL: nothing of this.
 *
 * norm == NULL in rnad_learn_bucketed / rnad_learn_bucketed_compact: the sums stay in `accumulators` (losses, dlogit_tab, dv_tab are
 * not written) and rnad_bucket_finish completes the update -- so that a data-parallel caller's all-reduce of the normalisers
 * (learn/vtrace.py:373,388 are batch-global counts) runs beside the learner kernel instead of in front of it.
 * ---------------------------------------------------------------------------------------------- */
/* Per-step scalars in DEVICE memory (optional everywhere: NULL = use the immediate arguments).  With them a captured hipGraph of a
 * whole training step can be replayed step after step: the host only rewrites these 16 bytes. */
typedef struct rnad_step_params {
    uint64_t seed;                  /* noise seed of the rollout (replaces the `seed` argument) */
    float alpha, one_minus_alpha;   /* rnad.py:497 (replace the fields of rnad_learn_params_t) */
} rnad_step_params_t;
/* Groups of (player, state) rows with the same observation, those of more than one row (csrc/rows_dedup.hip; rnad_hip.TreeHandle.obs_dedup):
 * the rows of group g are order[start[g] .. start[g + 1]), ascending, the first of them its representative. */
typedef struct rnad_row_groups {
    int32_t n_groups;
    const int32_t *start;   /* device int32 [n_groups + 1] */
    const int32_t *order;   /* device int32 [start[n_groups]] */
    const int32_t *first;   /* optional (NULL: not given), device int32 [n_groups][4][64]: the group's first 256 rows once more, padded --
                             * first[g][k][l] = order[start[g] + 64 k + l] while that is a row of group g, else -1 -- so that k_bucket_finish
                             * requests a group's accumulators without walking start -> order first (r06) */
    int32_t rows_below_cut; /* non-zero: the caller's `rows` list holds no row above the buckets (those are converted by the wave-per-row
                             * workgroups whatever the list says), so the row threads skip the lookup of each row's bucket -- one dependent
                             * load less on their path (r06) */
} rnad_row_groups_t;

/* *device_params = {seed, alpha, one_minus_alpha}, enqueued on `stream` (the values travel as kernel arguments: safe to call again
 * before the GPU has consumed the previous values). */
int rnad_step_params_set(rnad_step_params_t *device_params, uint64_t seed, float alpha, float one_minus_alpha, void *stream);
/* The scalars of the NEXT steps, in device memory: a replayed step then needs no launch before it.  `live` is what the kernels of a
 * step read (a rnad_step_queue_t* is passed wherever a rnad_step_params_t* device_params is expected); rnad_step_queue_set writes n
 * entries (1 <= n <= RNAD_STEP_QUEUE, HOST array, copied at the call: they travel as kernel arguments), live = entries[0], cursor = 0;
 * rnad_optimizer_step(..., advance = the queue) -- the last launch of a step -- moves on: cursor += 1, live = ahead[cursor] while
 * cursor < n.  The caller keeps count: once it has replayed n steps, or when a step's scalars are not the ones it queued (a logging
 * step in between took a seed; alpha left the schedule), it sets the queue again. */
#define RNAD_STEP_QUEUE 32
typedef struct rnad_step_queue {
    rnad_step_params_t live;
    int64_t cursor, n;
    rnad_step_params_t ahead[RNAD_STEP_QUEUE];
} rnad_step_queue_t;
int rnad_step_queue_set(rnad_step_queue_t *device_queue, int n, const rnad_step_params_t *entries, void *stream);
int rnad_bucket_plan(const rnad_tree_t *tree, int64_t B, int64_t *out);
int rnad_bucket_map(const rnad_tree_t *tree, int64_t B, int32_t *bucket_of, int32_t *n_groups);
/* n_shared[b] (host int32 [out[1] buckets]): the env steps a lane of bucket b shares with the whole bucket (above its group, plus the two
 * at the root of a group that is one subtree) -- the rows of a column of the compact trajectory's `states` that are neither written nor read. */
int rnad_bucket_shared_steps(const rnad_tree_t *tree, int64_t B, int32_t *n_shared);
int rnad_rollout_bucketed(const rnad_tree_t *tree, const rnad_traj_t *traj, const float *table, int64_t table_stride,
                          int table_is_policy, const float *value_table, int64_t value_stride, uint64_t seed, int64_t lane0,
                          const rnad_step_params_t *device_params, void *scratch, int32_t *lane_ids, int32_t *items, int32_t *n_items,
                          double *norm, void *stream);
int rnad_rollout_bucketed_compact(const rnad_tree_t *tree, int T_cap, int64_t B, const float *table, int64_t table_stride,
                                  int table_is_policy, uint64_t seed, int64_t lane0, const rnad_step_params_t *device_params,
                                  void *scratch, int32_t *lane_ids, int32_t *items, int32_t *n_items, double *norm, void *states,
                                  int32_t *alive, uint64_t *acts, float *final_reward, int32_t *visited, void *stream);
/* rnad_rollout_bucketed_compact + rnad_rows_expand inside its own launches (distinct observations, csrc/rows_dedup.hip: `tables` were
 * evaluated on one representative row per observation; rep_of: int32 [2S]).  The keys pass reads the upper states' policy rows through
 * rep_of and the copies are made by extra workgroups of the sort's scan launch (r05; r04: by the keys pass itself), i.e. they are
 * complete before the rollout launch, which is the first to read the other rows; walks that
 * cannot carry them (global / hybrid tables) get a launch of rnad_rows_expand in front. */
int rnad_rollout_bucketed_compact_expand(const rnad_tree_t *tree, int T_cap, int64_t B, const float *table, int64_t table_stride,
                                         int table_is_policy, uint64_t seed, int64_t lane0, const rnad_step_params_t *device_params,
                                         void *scratch, int32_t *lane_ids, int32_t *items, int32_t *n_items, double *norm, void *states,
                                         int32_t *alive, uint64_t *acts, float *final_reward, int32_t *visited, const int32_t *rep_of,
                                         int n_tables, float *const *tables, const int32_t *floats_per_row, void *stream);
/* Rollout AND learner of the batch in one call (r04): rnad_rollout_bucketed_compact(_expand) (a table of policy rows; n_tables == 0: no
 * copies to carry) followed by rnad_learn_bucketed_compact of the batch it plays with T = T_cap (learn/rnad.py:503-505 + :365-425 on the
 * on-policy batch) -- keys, sort, then ONE launch in which the workgroup of a work item plays its lanes and adds up their update right
 * away, then the alive counts (`alive`, `norm`: required) and, flags & RNAD_PLAY_LEARN_FINISH, rnad_bucket_finish with the batch's own
 * normalisers (a data-parallel caller leaves the flag out, all-reduces `norm` and calls rnad_bucket_finish itself).  Trajectory, counts,
 * accumulators and gradient tables are those of the two calls, bit for bit.  No loss sums (a logging step takes the two calls), no
 * visited flags.  RNAD_PLAY_LEARN_DISTINCT: larger work items, and the learner half runs once per DISTINCT trajectory of an item -- a
 * state has one parent entry, so the state a lane was last alive in and the outcome it drew there fix its trajectory -- weighted with
 * the number of lanes that took it: 64-bit integer sums, the same bits.  Pays once the policy has sharpened (fewer distinct
 * trajectories per item), costs ~4 % under uniform policies; the environment variable RNAD_FUSED_DISTINCT=0/1 overrides the flag. */
/* norm_global (device f64 [2], optional, with RNAD_PLAY_LEARN_FINISH): the finish divides by THESE normalisers instead of the batch's
 * own -- the N_P of a global batch whose other lanes other ranks play.  On a tree whose episodes all last 2 * max_depth env steps
 * (rnad_tree_info(tree, 6)) they are known without a collective: N_0 = N_1 = global lanes * T_cap / 2. */
#define RNAD_PLAY_LEARN_FINISH 1
#define RNAD_PLAY_LEARN_DISTINCT 2
/* The learner on the tree's LEAF PATHS (r05; `leaf` of rnad_rollout_learn_bucketed_compact, NULL: the learner per lane).  A state has
 * exactly one parent entry (tree.py:311-330: ids are DFS pre-order), so the transition a lane leaves the tree by -- (state, row action,
 * column action, outcome) with index == 0 -- fixes its whole trajectory: states, actions, reward (episode.py:96-125).  The on-policy
 * update of learn/rnad.py:365-425 is a sum over the lanes of addends that depend on the trajectory alone, i.e.  sum over the tree's
 * terminal transitions of  (lanes that took it) x (the addends of that trajectory)  -- in the learner's 64-bit fixed point: the same
 * bits.  The caller builds, ONCE per tree and cut, the batch of those trajectories in the compact layout (one column per terminal
 * transition, sorted by bucket, with its work items: what rnad_rollout_bucketed_compact would leave had every trajectory been played
 * once; rnad_leaf_paths_pack writes the relative states); per step the rollout's launch leaves, per lane, the column of the transition it
 * left the tree by (col_of: terminal transition -> column; in `scratch`), and the learner runs on the n_cols columns -- a work item counts,
 * in LDS, the lanes of its bucket that fall into its columns (items of at most 256 columns) and skips the columns nobody played.
 * Its cost no longer grows with the batch, and lanes that share a trajectory are learned from once (configs[1]: 531 441 columns for
 * 2^20 .. 2^22 lanes).  Rollout and learner are two launches then.  Needs T_cap == 2 * depth (every lane has left the tree by the end
 * of the window). */
typedef struct rnad_leaf_paths {
    int64_t n_cols;            /* columns: one per terminal transition of a reachable state */
    int32_t rows, T_cap;       /* the cut (rnad_bucket_plan out[0]) and window the columns were packed for */
    const void *states;        /* relative states [T_cap + 1, n_cols], 1 or 2 bytes each (rnad_leaf_paths_pack) */
    const uint64_t *acts;      /* [n_cols] 3 bits of action per env step */
    const float *final_reward; /* [n_cols] value of the terminal transition */
    const int32_t *items;      /* [max_items][4] work items over the columns: begin, count (<= 256), bucket, single */
    const int32_t *n_items;    /* device int32: their number */
    int32_t max_items;
    const int32_t *col_of;     /* [S * A * A * C]: column of transition ((s * A + a0) * A + a1) * C + c, -1 if it is not terminal */
    /* r06, optional (NULL / 0: every bucket's lanes are counted by the learner's work items): a bucket that holds at least crowded_lanes
     * lanes of the batch (and at most 1024 columns) is counted by the ROLLOUT's work items instead -- a histogram over the bucket's columns
     * in LDS, its non-zero bins added to col_count -- so that a sharpened policy, whose lanes pile up in few buckets and on few
     * trajectories, does not make every learner work item of such a bucket scan all of its lanes. */
    int32_t *col_count;          /* device int32 [n_cols], zero before the first step; the learner leaves it zero */
    const int32_t *bucket_col0;  /* device int32 [buckets + 1]: first column of every bucket (columns are sorted by bucket) */
    int32_t crowded_lanes;
} rnad_leaf_paths_t;
/* relative states of the columns from their dense states (int32 [T1, n_cols], bucket order, 0 once the episode is over) under the cut
 * rnad_bucket_plan chooses for batches of plan_B lanes; mismatch (device int32, caller-zeroed): set if a column does not lie in its
 * item's bucket; rows_out / rel_bytes_out (host): the cut's table rows and the width of a relative state. */
int rnad_leaf_paths_pack(const rnad_tree_t *tree, int64_t plan_B, int T1, int64_t n_cols, const int32_t *indices, const int32_t *items,
                         const int32_t *n_items, int32_t max_items, void *states, int32_t *mismatch, int32_t *rows_out, int32_t *rel_bytes_out,
                         void *stream);
int rnad_rollout_learn_bucketed_compact(const rnad_tree_t *tree, int T_cap, int64_t B, const float *table, int64_t table_stride,
                                        uint64_t seed, int64_t lane0, const rnad_step_params_t *device_params, void *scratch,
                                        int32_t *lane_ids, int32_t *items, int32_t *n_items, double *norm, void *states, int32_t *alive,
                                        uint64_t *acts, float *final_reward, const int32_t *rep_of, int n_tables, float *const *tables,
                                        const int32_t *floats_per_row, const float *fast_records, const rnad_learn_params_t *hp,
                                        void *accumulators, int flags, const double *norm_global, float *dlogit_tab, float *dv_tab,
                                        const int32_t *rows, const int64_t *n_rows, const rnad_row_groups_t *groups,
                                        const rnad_leaf_paths_t *leaf, void *stream);
int rnad_bucket_indices(const rnad_tree_t *tree, int T1, int64_t B, const void *states, const int32_t *items, const int32_t *n_items,
                        int32_t *indices, void *stream);
int rnad_bucket_pack_states(const rnad_tree_t *tree, int T1, int64_t B, const int32_t *indices, const int32_t *items,
                            const int32_t *n_items, void *states, int32_t *mismatch, void *stream);
int rnad_bucket_expand(const rnad_tree_t *tree, int T, int64_t B, const int32_t *indices, const uint64_t *acts,
                       const float *final_reward, const float *records, uint8_t *mask_bits, float *policy, int32_t *actions,
                       float *rewards, void *stream);
int64_t rnad_bucket_record_stride(int A);
int64_t rnad_bucket_fast_record_stride(int A);
int64_t rnad_bucket_policy_row_stride(int A);
int rnad_bucket_records(const rnad_tree_t *tree, const float *logit_tab, const float *v_tab, const float *v_target_tab,
                        const float *logit_reg_tab, const float *logit_reg_tab_, const rnad_learn_params_t *hp,
                        const rnad_step_params_t *device_params, float *records, float *fast_records, float *policy_rows,
                        const int32_t *rows, const int64_t *n_rows, void *stream);
/* Distinct observations (csrc/rows_dedup.hip).  What a net contributes to a (player, state) row is a function of the row's observation
 * alone (nn/net.py:37-51), so rows with the same observation share their net outputs and records, and their gradients dL/dout can be
 * added up before the backward.  The caller groups the rows by the bits of their observation (rnad_hip.TreeHandle.obs_dedup) and
 *   rnad_rows_expand      after a table launch on one representative row per group: tables[k][r] = tables[k][rep_of[r]] for every row r
 *                         (rep_of: int32 [rows], the representative of r's group, itself for a representative; tables: up to 4 device
 *                         tables of floats_per_row[k] floats per row, multiples of 4, 16-byte aligned);
 *   rnad_rows_segment_sum after rnad_bucket_finish: for every group g with more than one row -- rows order[start[g] .. start[g + 1]),
 *                         ascending, the first its representative -- dlogit_tab / dv_tab of the representative <- the sums over the group
 *                         (one wave per group, a fixed reduction tree); the backward then runs on the representatives alone;
 *   `groups` of rnad_bucket_finish / rnad_learn_bucketed_compact (optional) does the same inside k_bucket_finish: rows / n_rows then list the
 *                         rows OUTSIDE the groups (the groups of one row), a wave per group converts and adds up its rows' int64 sums
 *                         into dlogit_tab / dv_tab of the representative -- the bits of finish on all rows + rnad_rows_segment_sum -- and
 *                         the table rows of the groups' other rows are not written.  The caller's promise: no row above the buckets
 *                         (rnad_bucket_map: bucket_of >= n_groups) is in a group of more than one row (those rows' sums live in the
 *                         replicas; rnad_hip.ObsDedup.groups_below_cut checks). */
int rnad_rows_expand(int64_t rows, const int32_t *rep_of, int n_tables, float *const *tables, const int32_t *floats_per_row, void *stream);
int rnad_rows_segment_sum(int n_groups, const int32_t *start, const int32_t *order, int A, float *dlogit_tab, float *dv_tab, void *stream);

/* rnad_mlp_rows_records (csrc/mlp_rows.hip): the table evaluations of a tabular update AND rnad_bucket_records in one launch -- replaces
 * learn/rnad.py:373,378 on the 2S rows (the learner net's two heads, the target net's value head; nn/net.py:40-43) followed by the
 * row-only arithmetic of :374,382 and learn/vtrace.py (see rnad_bucket_records).  packed_net / packed_target: weight images of
 * rnad_mlp_pack(_fold)_multi (fold != 0: the FOLD images; `obs` is then the tree's own observation table); logit_tab [2S, A], v_tab [2S],
 * v_target_tab [2S] are OUTPUTS (the same tables rnad_mlp_forward_multi would give, up to the order of the second-layer sums);
 * records (optional) / fast_records / policy_rows (optional) exactly as rnad_bucket_records writes them from those tables (same function,
 * same bits).  policy_from_table != 0: logit_tab is an INPUT (a staged actor wrote it) and only the two value heads are evaluated --
 * with rows / n_rows (device memory, both NULL = all 2S rows) the lazy-rows step of learn/rnad.py.  Width: a multiple of 32 up to 256;
 * A <= 3 with the policy head, A <= 5 (fold) / 4 without (rnad_mlp_rows_records_supported: other shapes take the two launches). */
int rnad_mlp_rows_records_supported(int A, int W, int fold, int policy_from_table);
/* rnad_mlp_rows_actor: rnad_mlp_forward_actor with the mapping of csrc/mlp_rows.hip (weights in registers, a wave per hidden tile) -- the
 * three staging launches of a large tree's actor, each on a short row list.  Same outputs up to the order of the second-layer sums. */
int rnad_mlp_rows_actor_supported(int A, int W, int fold);
int rnad_mlp_rows_actor(const rnad_tree_t *tree, const int32_t *rows, const int64_t *n_rows, int W, int fold, const float *packed, const void *obs,
                        int obs_half, float *logits, float *policy_rows, void *stream);
int rnad_mlp_rows_records(const rnad_tree_t *tree, int W, int fold, const float *packed_net, const float *packed_target, const void *obs,
                          int obs_half, const int32_t *rows, const int64_t *n_rows, int policy_from_table, float *logit_tab, float *v_tab,
                          float *v_target_tab, const float *logit_reg_tab, const float *logit_reg_tab_, const rnad_learn_params_t *hp,
                          const rnad_step_params_t *device_params, float *records, float *fast_records, float *policy_rows, void *stream);
int rnad_learn_bucketed(const rnad_tree_t *tree, int T, int64_t B, const int32_t *indices, const int32_t *actions,
                        const float *rewards, const float *mu, const float *records, const int32_t *items, const int32_t *n_items,
                        const double *norm, const rnad_learn_params_t *hp, void *accumulators, double *losses, float *dlogit_tab,
                        float *dv_tab, void *stream);
int rnad_learn_bucketed_compact(const rnad_tree_t *tree, int T, int64_t B, const void *states, const uint64_t *acts,
                                const float *final_reward, const float *fast_records, const float *records, const int32_t *items,
                                const int32_t *n_items, const double *norm, const rnad_learn_params_t *hp, void *accumulators,
                                double *losses, float *dlogit_tab, float *dv_tab, const int32_t *rows, const int64_t *n_rows,
                                const void *rollout_scratch, int rollout_T_cap, int32_t *alive, double *norm_out,
                                const rnad_row_groups_t *groups, void *stream);
int rnad_bucket_alive(const rnad_tree_t *tree, int T_cap, int64_t B, const void *scratch, int32_t *alive, double *norm, void *stream);
/* rnad_rollout_bucketed_compact in two calls, for an actor that is evaluated in stages (trees that are large next to the batch):
 * rnad_bucket_sort = the keys pass + the sort; it reads the actor's rows of the UPPER states of the cut and of the absorbing state
 * only (rnad_bucket_plan / rnad_bucket_map say which).  group_flags (int32 [2S], optional): 1 for both rows of every state inside a
 * group some lane descends into, 0 elsewhere -- after rnad_compact_valid, the rows the actor still has to be evaluated on.
 * staged_rows (int32 [2S] capacity) / n_staged (device int64), optional: the same rows as an ascending LIST with its length, written by the
 * sort's last kernel (no flags, no compaction: what a caller hands to rnad_mlp_forward_rows and to rnad_bucket_play as rows / n_rows).
 * visited (int32 [2S], optional): cleared here for the rollout (then pass visited_is_clear = 1 to rnad_bucket_play, which otherwise clears it
 * with a launch of its own).
 * rnad_bucket_play = the rollout itself (+ the alive counts), same scratch, same seed / lane0 / device_params, the lane_ids and
 * the work list (items / n_items) of the sort; with a logits table (table_is_policy == 0) `rows` / `n_rows` name the rows that were evaluated since (their policy head is
 * taken here), NULL = all. */
int rnad_bucket_sort(const rnad_tree_t *tree, int T_cap, int64_t B, const float *table, int64_t table_stride, int table_is_policy,
                     uint64_t seed, int64_t lane0, const rnad_step_params_t *device_params, void *scratch, int32_t *lane_ids,
                     int32_t *items, int32_t *n_items, double *norm, int32_t *group_flags, int32_t *staged_rows, int64_t *n_staged,
                     int32_t *visited, void *stage, int32_t *stage_rows0, void *stream);
/* Second staging level of a tabular actor on a tree that is large next to the batch (configs[3]: the non-empty groups hold 807 k rows, the
 * batch visits 103 k).  rnad_bucket_sort with `stage` (rnad_bucket_stage_bytes(tree, B) bytes, zeroed once; its first two int64 are the
 * lengths of the two row lists below) also records, per lane, the root of the group subtree it enters.  Then, on the same stream:
 *   rnad_bucket_stage_rows(level 0) -> rows = both players' rows of those roots (or hand rnad_bucket_sort `stage_rows0`: its last kernel
 *                                      then writes this list itself, one launch less);  evaluate the actor on them;
 *   rnad_bucket_stage_walk          -> draws every lane's transition at its root exactly as the rollout will (environment/episode.py:96-125
 *                                      with the seeded draws of rnad_rng.h: same counters, same rows, same outcome);
 *   rnad_bucket_stage_rows(level 1) -> rows = the rows of every subtree hanging below a state some lane was just drawn into;  evaluate the
 *                                      actor on them;  rnad_bucket_play then only reads rows that were evaluated.
 * configs[3]: 176 k rows instead of 807 k.  The lists are unordered (one launch each, a reservation per workgroup); `rows` holds up to 2S
 * entries; seed / device_params as in rnad_bucket_sort (they stamp the marks: nothing is cleared between steps). */
int64_t rnad_bucket_stage_bytes(const rnad_tree_t *tree, int64_t B);
int rnad_bucket_stage_rows(const rnad_tree_t *tree, int64_t B, int level, uint64_t seed, const rnad_step_params_t *device_params, void *stage,
                           int32_t *rows, void *stream);
int rnad_bucket_stage_walk(const rnad_tree_t *tree, int T_cap, int64_t B, const float *policy_rows, int64_t stride, uint64_t seed, int64_t lane0,
                           const rnad_step_params_t *device_params, const void *scratch, const int32_t *lane_ids, void *stage, void *stream);
int rnad_bucket_play(const rnad_tree_t *tree, int T_cap, int64_t B, const float *table, int64_t table_stride, int table_is_policy,
                     const int32_t *rows, const int64_t *n_rows, uint64_t seed, int64_t lane0, const rnad_step_params_t *device_params,
                     void *scratch, const int32_t *lane_ids, const int32_t *items, const int32_t *n_items, double *norm, void *states,
                     int32_t *alive, uint64_t *acts, float *final_reward, int32_t *visited, int visited_is_clear, void *stream);
int rnad_bucket_finish(const rnad_tree_t *tree, int64_t B, const double *norm, const rnad_learn_params_t *hp, void *accumulators,
                       double *losses, float *dlogit_tab, float *dv_tab, const int32_t *rows, const int64_t *n_rows,
                       const rnad_row_groups_t *groups, void *stream);

/* torch.nn.utils.clip_grad_norm_(parameters, max_norm) of learn/rnad.py:456 over one flat fp32 gradient bucket (all of a net's
 * .grad tensors back to back): g *= min(max_norm / (||g||_2 + 1e-6), 1), in place, one launch.  total_norm: optional device
 * float receiving ||g||_2. */
int rnad_clip_grad_norm(int64_t n, float *grads, float max_norm, float *total_norm, void *stream);

/* The tail of a training step in ONE launch  --  learn/rnad.py:456 (clip_grad_norm_), :514 (torch.optim.Adam.step: no weight decay,
 * no amsgrad, as constructed at rnad.py:232-237) and :516-523 (EMA target) for up to 8 parameter tensors whose gradients lie back to
 * back in one flat fp32 bucket `grads` (read only: the clipped values go straight into Adam).  sizes: HOST array of element counts; param / exp_avg / exp_avg_sq / step /
 * target: HOST arrays of device pointers (step: torch's per-tensor fp32 step counters on the device, incremented here; target may be
 * NULL: no EMA).  The state tensors are torch.optim.Adam's own, so checkpoints keep the reference format.
 * mlp_A > 0 (with mlp_W, packed_param, packed_target; either image may be NULL): the 8 tensors are the fused MLP's Linear tensors in
 * the order rnad_mlp_pack takes them, and every new weight / new target weight is ALSO written into its slot of that net's packed
 * image (rnad_mlp_pack's layout; mlp_fold != 0: rnad_mlp_pack_fold_multi's) -- the images stay current without a pack launch per
 * step.  mlp_A == 0: none of this.  advance (optional, device memory): see rnad_step_queue_t -- the workgroup that finishes last moves the
 * queue of per-step scalars on.  ticket (device, one uint32 the caller zero-initialises ONCE and then leaves alone): the kernel's
 * workgroups count themselves in it and the last one resets it -- one word per optimiser, so that two trainers stepping on two streams
 * of one device do not share a counter. */
typedef struct rnad_adam_params {
    float lr, beta1, beta2, eps;  /* rnad.py:232-237 */
    float max_norm;               /* grad_clip, rnad.py:456 */
    float ema;                    /* gamma_averaging, rnad.py:516-523 */
} rnad_adam_params_t;
int rnad_optimizer_step(int n_tensors, const int64_t *sizes, float *const *param, float *grads, float *const *exp_avg,
                        float *const *exp_avg_sq, float *const *step, float *const *target, const rnad_adam_params_t *hp,
                        float *total_norm, int mlp_A, int mlp_W, int mlp_fold, float *packed_param, float *packed_target, rnad_step_queue_t *advance,
                        uint32_t *ticket, void *stream);
/* ------------------------------------------------------------------------------------------------
 * NashConv  --  util/metric.py:93-175 (NashConvData.get_nashconv), level-batched on the GPU
 * instead of one Python frame per state.  joint_policy f32 [S,2A] (device) for every state below
 * `state_index`; root_policy f32 [2A] (device) is the row used AT state_index (the reference
 * recursion passes self.joint_policy to descendants, :148-151).  Outputs f32/int32 [S] (device):
 * only states reachable from state_index are written.
 * ---------------------------------------------------------------------------------------------- */
int rnad_nashconv(const rnad_tree_t *tree, const float *joint_policy, const float *root_policy,
                  int64_t state_index, float reach, float *row_best, float *col_best, float *reach_out,
                  int32_t *depth_out, void *stream);
/* Observations of EVERY state for both players (metric.py:66-81): obs_row, obs_col f32 [S,2,A,A]. */
int rnad_observe_all(const rnad_tree_t *tree, float *obs_row, float *obs_col, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Native tree generator (HOST code)  --  environment/tree.py:164-366 (generate, _transition_probs,
 * _solve) for trees too large for the Python recursion (66 431 states take > 100 s there).
 * Regular shape only: every state has A x A legal actions and depth_bound - 1 below it; chance
 * profiles are Dirichlet(1/C) thresholded and renormalised (tree.py:182-197); terminal payoffs are
 * drawn uniformly from terminal_values (tree.py:271-275); each state's matrix game is solved
 * exactly (Shapley-Snow enumeration, same sub-matrix order as tests/golden/_pygambit_stub.py);
 * ids are the reference's DFS pre-order with the absorbing state 0 prepended (tree.py:311-366).
 * prune_num/prune_den: each child's depth_bound is additionally lowered by 2 with probability
 * prune_num/prune_den (reference main.py:37).  Randomness: splitmix64(seed), NOT numpy's.
 * Call with outputs NULL to get the state count, then with buffers of that size.
 * ---------------------------------------------------------------------------------------------- */
int64_t rnad_tree_generate(int A, int C, int depth_bound, float transition_threshold, const float *terminal_values,
                           int n_terminal_values, int prune_num, int prune_den, uint64_t seed, int64_t capacity,
                           int64_t *index, float *value, float *chance, float *expected_value, float *legal,
                           float *root_value, float *solution);
/* tree.py:199-234: one zero-sum matrix game M [ra,ca] (row-major, row player maximises) ->
 * solution f32 [2*max_actions] (row strategy | column strategy), returns the game value via *value. */
int rnad_solve_matrix(const float *M, int ra, int ca, int max_actions, float *solution, float *value);

/* ------------------------------------------------------------------------------------------------
 * Kernel timing (bench.py's roofline leg): rnad_prof_enable(mask) brackets every launch of kernel k with a pair of
 * hipEvents on the launching stream when bit (1 << k) of `mask` is set (k: 0 = observe, 1 = act/transition,
 * 2 = learn_fused, 3 = mlp_forward, 4 = mlp_backward, 5 = bucket keys, 6 = bucket sort passes, 7 = bucket rollout,
 * 8 = bucket learner, 9 = bucket finish; -1 = all, 0 = off) and drops earlier measurements.  Each
 * bracket costs a few microseconds of dispatch latency, so bracket only what is being measured.
 * rnad_prof_read synchronises the device and returns launches and total milliseconds since the last enable.
 * ---------------------------------------------------------------------------------------------- */
int rnad_prof_enable(int mask);
int rnad_prof_read(int which, int64_t *launches, double *total_ms);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* RNAD_HIP_H */
