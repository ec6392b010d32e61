/*
 * rubiks_hip.h -- C ABI of librubiks_hip.so, the MI355X (gfx950) cube engine.
 *
 * The reference (peleiden/librubiks) is pure Python and has no FFI of its own; the boundary this
 * library replaces is the module surface of `librubiks.cube` (librubiks/cube/__init__.py:2) and
 * the expand-children sections of `librubiks.solving.agents`.  Each entry point below names the
 * reference function (file:line under /root/reference) whose arithmetic it performs.  The ctypes
 * stub a maintainer would add on the reference side is shown in INTEGRATION.md.
 *
 * Conventions
 *   - Every entry returns 0 on success or a negative RK_E* code; rk_last_error() gives the
 *     thread-local message of the last failure.  No entry ever falls back to a CPU path.
 *   - Pointers named d_* are DEVICE pointers (hipMalloc / tensor.data_ptr()); h_* are host
 *     pointers.  The library never frees or retains caller memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Device-pointer entries
 *     are stream-ordered and do not synchronise; *_host entries copy, launch, copy back and
 *     synchronise `stream` before returning.
 *   - A 20-byte state is int8[20]: 8 corner codes slot*3+ori then 12 edge codes slot*2+ori
 *     (cube.py:58-65).  State arrays are dense row-major (n, 20); their base must be 4-byte
 *     aligned (always true for rows of a 256-B aligned allocation).
 *   - An action index a in [0,12) means face a/2, direction 1-(a%2) (cube.py:33-34).  PRECONDITION of the
 *     device-pointer entries: every action code is < 12.  A larger code is treated as action 0 (the kernels never
 *     index past the move table) and the result for that row is meaningless; the *_host entries check and fail
 *     with RK_EINVAL, as the reference's table indexing would raise.
 *   - repr: RK_REPR_2024 = 20-byte cubie codes, RK_REPR_686 = int8 (6,8,6) one-hot, 288 bytes
 *     (cube.py:67-71).
 */
#ifndef RUBIKS_HIP_H
#define RUBIKS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RK_REPR_2024 0
#define RK_REPR_686  1

#define RK_OK          0
#define RK_EINVAL     -1   /* bad argument (null pointer, misaligned base, unknown repr, ...) */
#define RK_EHIP       -2   /* a HIP runtime call failed; message holds hipGetErrorString */
#define RK_ECAPACITY  -3   /* an engine ran out of the capacity it was created with */
#define RK_ESTATE     -4   /* call not valid in the engine's current state */

#define RK_OH_F32  0
#define RK_OH_F16  1
#define RK_OH_BF16 2
#define RK_OH_STATES 3   /* engines only: hand the net the 20-byte states themselves (first layer fused, rk_ohl_*) */

#define RK_OHL_GATHER 0  /* exact float32 gather-sum through an LDS-resident weight slice */
#define RK_OHL_MFMA   1  /* bf16 MFMA with the one-hot operand synthesised in registers; the form follows the batch size */
#define RK_OHL_MFMA_DIRECT 2  /* ... always the few-rows form: one 32 x 32 output tile per wave, weights straight from global memory */
#define RK_OHL_MFMA_TILED  3  /* ... always the many-rows form: a 64-column weight tile resident in LDS (same bits as the direct form) */
/* epilogue activation of the fused first layer (rk_ohl_set_epilogue) */
#define RK_OHL_ACT_NONE 0
#define RK_OHL_ACT_ELU  1  /* x > 0 ? x : alpha (exp(x) - 1)          nn.ELU, the reference's default, model.py:30 */
#define RK_OHL_ACT_RELU 2

/* ---- library ------------------------------------------------------------------------------ */
int         rk_version(void);
const char *rk_last_error(void);
/* Select `device` for the calling thread and check that it is a gfx950 part. */
int         rk_init(int device);

/* The large launches of the write-heavy kernels (fan-out, one-hot, 6x8x6 fan-out, multi_rotate) run in a PACED form: inputs read
 * first, every tile's stores released on a fixed-rate schedule.  Results never depend on it.  mode 0 switches
 * the form off for this process (the unpaced kernels run at every size), 1 on, -1 back to the environment's choice (RK_PACE,
 * default on).  bench.py uses it to time both forms on the same box in one run (`frac_ring_same_box`). */
int rk_set_pacing(int mode);
/* The fan-out's store schedule is measured once per device and process (rk_init does it; about 4 ms, 272 MB of scratch that is freed
 * again): the ring form and 2.0 / 2.1 / 2.2 / 2.4 ns per 64-parent tile on 1 Mi parents.  The compiled 2.1 ns stay unless another
 * candidate is more than 3 % faster -- a part whose HBM does not keep that schedule gets the one it does keep, the unpaced form
 * included.  rk_calibrate_pacing(force != 0) measures again.  Skipped when RK_PACE_TAU_PS, RK_PACE=0 or RK_PACE_CALIBRATE=0 say so.
 * rk_get_pacing: *tau_ps = picoseconds per tile in force on the current device (0 = unpaced), *source = 0 compiled default (not
 * calibrated yet), 1 calibrated, 2 fixed by the environment or calibration impossible; h_us (nullable, 5 floats) = measured
 * microseconds per 1 Mi-parent launch of {ring form, 2.0, 2.1, 2.2, 2.4 ns}, zeros if nothing was measured.  Any pointer may be null. */
int rk_calibrate_pacing(int force);
int rk_get_pacing(unsigned int *tau_ps, int *source, float *h_us);
/* Everything the paced forms keep is PER DEVICE (time-base cells, the measured schedule, the turn gate): this is the slot of `device` in
 * those tables -- the device's own index, or -1 for a device the tables have no room for (it runs the unpaced forms).  Pure host
 * arithmetic, no HIP call: a diagnostic, and what the CPU test pins ("two devices never share a slot"). */
int rk_pace_slot_of_device(int device);
/* Stream lifetime.  Paced launches on different streams take turns: the library remembers, per device, the
 * stream of the last paced launch and, when the next one arrives on another stream, makes it wait (event record + stream wait)
 * for that stream.  It only ever remembers a stream it may rely on: the null stream, and streams REGISTERED with
 * rk_stream_register(stream) -- the caller's promise that the stream stays alive until rk_stream_forget(stream), which must be
 * called before the stream is destroyed.  A paced launch on an unregistered stream takes no turn (it may overlap another paced
 * launch: slower, never wrong) and nothing about it is kept.  The Python shim registers torch's pooled streams, which are never
 * destroyed.  Both entries are cheap and idempotent; neither touches the stream. */
int rk_stream_register(void *stream);
int rk_stream_forget(void *stream);

/* Move tables, written to HOST memory.
 * RK_REPR_2024: uint8 (12,2,24) absolute table T[a][kind][v] = v + maps[dir][face][kind][v]
 *               (maps.py:107-145).   RK_REPR_686: uint8 (12,48) sticker-slot permutation,
 *               new[slot] = old[perm[a][slot]], slot = 8*face+pos (cube.py:330-347). */
int rk_tables(int repr, uint8_t *h_out);
/* The six face definitions the tables are generated from, in the order F, B, T, D, L, R, written to HOST memory as uint8 (6,14):
 * [0:4] ring of corner slots and [4:8] ring of side slots a positive turn cycles (ring[j] -> ring[j+1]), [8] the corner
 * orientation that stays while the other two swap, [9] 1 if the turn flips side orientations (maps.py:74-98 `Actions`), [10:14]
 * the face's four neighbours in positive direction (maps.py:149-156 `neighbors_686`).  What `librubiks.cube.maps` exposes as
 * `Actions` / `neighbors_686`; the drop-in's maps.py rebuilds both from this entry. */
int rk_face_definitions(uint8_t *h_out);
/* Solved state in HOST memory: int8[20] (cube.py:58-65) or int8[288] (cube.py:67-71). */
int rk_solved(int repr, int8_t *h_out);

/* ---- plain device memory helpers (for hosts that do not bring their own allocator) -------- */
int rk_malloc(void **d_ptr, size_t bytes);
int rk_free(void *d_ptr);
int rk_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream);
int rk_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream);
int rk_memset(void *d_dst, int value, size_t bytes, void *stream);
int rk_stream_synchronize(void *stream);

/* ---- cube hot path, device pointers -------------------------------------------------------- */

/* multi_rotate (cube.py:49-52, 256-263; 686: cube.py:349-361): d_out[i] = move d_actions[i]
 * applied to d_states[i].  d_out may equal d_states. */
int rk_multi_rotate(int repr, const int8_t *d_states, const uint8_t *d_actions, int8_t *d_out,
                    size_t n, void *stream);
/* multi_rotate + multi_is_solved of the MOVED states in one launch -- the pair the reference's per-row callers always run back
 * to back (agents.py:157-159 ValueSearch, :696-703 EGVM, train.py:277-281): d_out[i] = move d_actions[i] of d_states[i] (d_out
 * may equal d_states), d_flags[i] (nullable) = d_out[i] is solved, d_stats (nullable, int64[2], initialised by the caller as for
 * rk_expand12: [0] += solved rows, [1] = min(first solved row, previous value)); at least one of the two outputs.  20-byte
 * states: 21 B read + 21 B written per state, the moved states are never read back.  (6x8x6: two launches inside.) */
int rk_multi_rotate_solved(int repr, const int8_t *d_states, const uint8_t *d_actions, int8_t *d_out, uint8_t *d_flags,
                           long long *d_stats, size_t n, void *stream);
/* Same as rk_multi_rotate with the reference's (faces, directions) pair as two uint8 arrays (20-byte states only). */
int rk_multi_rotate_fd(int repr, const int8_t *d_states, const uint8_t *d_faces, const uint8_t *d_dirs,
                       int8_t *d_out, size_t n, void *stream);
/* The device-pointer entries (rk_multi_rotate, rk_multi_rotate_solved, rk_multi_rotate_fd, rk_apply_sequences) take action
 * codes as they are: a code >= 12 is treated as action 0 -- the kernels never index past the move table -- and leaves a sticky
 * mark on the device.  This reads the mark into *h_seen (1: some launch since the last call saw such a code) and clears it
 * in ONE atomic exchange on the device, and synchronises `stream`.  The mark is per device, not per stream: it reports the
 * launches of every stream.  The host entries (rk_*_host) validate their arrays and fail with RK_EINVAL instead, like the
 * reference's table indexing raises IndexError (cube.py:33-34, 256-263). */
int rk_bad_actions_seen(int *h_seen, void *stream);

/* The 12-child fan-out `multi_rotate(repeat(S,12), *iter_actions(n))` (agents.py:277-281, :513,
 * :605; train.py:285) fused with `multi_is_solved` of the children (cube.py:88-89; agents.py:321,
 * :540; train.py:292).  d_children is (12 n, state) parent-major / action-minor and must be 16-byte
 * aligned.  d_solved (nullable) gets one byte per child.  d_stats (nullable) is int64[2] that the
 * caller zero/initialises: [0] += number of solved children, [1] = min(index of a solved child,
 * previous value) -- initialise [1] to INT64_MAX.  The statistics are meant for RARE hits (a search
 * looking for its goal): every wave that has a solved child reports with two device-scope atomics
 * on these two words, and atomics on one address complete about one every 25 ns chip-wide -- 12 M
 * states of which 4 000 are solved take 0.24 ms with d_stats and 0.04 ms without.  Use d_solved
 * when hits are common. */
int rk_expand12(int repr, const int8_t *d_parents, int8_t *d_children, uint8_t *d_solved,
                long long *d_stats, size_t n, void *stream);

/* Structure-of-arrays form of the fan-out for device-resident pipelines (same arithmetic, different layout):
 * d_parents uint32 [5][n] (plane j = bytes 4j..4j+3 of every state), d_children uint32 [12][5][n] (child a of
 * parent p has its dword j at [(a*5 + j)*n + p]), d_solved uint8 [12][n] (nullable).  Every access of a wavefront
 * is one contiguous run, so the kernel needs no LDS transpose.  d_stats as rk_expand12 (the index it reports is
 * 12 p + a).  rk_states_to_soa / rk_states_from_soa convert between (n,20) int8 rows and the five planes. */
int rk_expand12_soa(const uint32_t *d_parents, uint32_t *d_children, uint8_t *d_solved, long long *d_stats, size_t n,
                    void *stream);
int rk_states_to_soa(const int8_t *d_states, uint32_t *d_planes, size_t n, void *stream);
int rk_states_from_soa(const uint32_t *d_planes, int8_t *d_states, size_t n, void *stream);

/* multi_is_solved (cube.py:88-89).  d_flags (nullable) one byte per state; d_stats as above. */
int rk_multi_is_solved(int repr, const int8_t *d_states, uint8_t *d_flags, long long *d_stats,
                       size_t n, void *stream);

/* Move sequences from the solved state: the device half of scramble (cube.py:206-216) and
 * sequence_scrambler (cube.py:218-232).  d_actions is (depth, games) uint8, row d = d-th move of
 * every game (the layout of the reference's faces/dirs draws).  Game g emits `rows` =
 * (with_solved ? 1 : 0) + moves states, where moves = depth - with_solved: optionally the solved
 * state, then the state after each of its first `moves` moves.  If only_last != 0 just the final
 * state of each game is written (d_out is (games, state)), else d_out is (games*rows, state)
 * game-major. */
int rk_apply_sequences(int repr, const uint8_t *d_actions, int depth, int games, int with_solved,
                       int only_last, int8_t *d_out, void *stream);

/* The cube part of one Autodidactic-Iteration rollout in ONE launch (train.py:277-292): d_actions uint8 (depth, games) as for
 * rk_apply_sequences (not only_last) ->
 *   d_states        (games*depth, 20)    the states along every game's walk, game-major            (cube.py:218-232; train.py:277)
 *   d_state_flags   (games*depth) uint8  1 where such a state is solved, nullable                  (train.py:281)
 *   d_children      (12*games*depth, 20) their children, parent-major, action-minor                (train.py:285)
 *   d_child_flags   (12*games*depth) uint8 1 where a child is solved                               (train.py:292)
 *   d_stats         int64[2], nullable, as rk_expand12's: [0] += solved children, [1] = min(first solved child, previous value)
 * The same bytes as rk_apply_sequences + rk_multi_is_solved + rk_expand12, without writing the states once and reading them twice. */
int rk_rollout_fanout(int repr, const uint8_t *d_actions, int depth, int games, int with_solved, int8_t *d_states, uint8_t *d_state_flags,
                      int8_t *d_children, uint8_t *d_child_flags, long long *d_stats, void *stream);
/* as_oh (cube.py:130-133, 265-277; 686: cube.py:363-369): one-hot encode n states into
 * (n, 480) [2024] or (n, 288) [686] elements of out_dtype (RK_OH_*); d_out 16-byte aligned. */
int rk_as_oh(int repr, const int8_t *d_states, void *d_out, int out_dtype, size_t n, void *stream);

/* Fused one-hot -> first Linear of the net (cube.py:265-277 + model.py:127,150): y = as_oh(states) @ W^T + b without the
 * one-hot ever reaching HBM.  rk_ohl_create copies nn.Linear(480, H)'s weight (H, 480) row-major and bias (H), both
 * float32 or both bfloat16 (w_dtype = RK_OH_F32 / RK_OH_BF16; bias may be NULL), into the layouts the two routes use;
 * H must be a multiple of 64.  rk_ohl_forward: d_out (n, H) row-major, 16-byte aligned.
 *   RK_OHL_GATHER  y = ((b + w_0) + w_1) + ... + w_19 with float32 adds in that order (w_i = row 24 i + state[i] of W^T):
 *                  exact and reproducible; out_dtype RK_OH_F32 or RK_OH_BF16 (rounded to nearest even at the end)
 *   RK_OHL_MFMA    bf16 weights, float32 accumulation on the matrix cores, out_dtype RK_OH_BF16.  Two forms with identical
 *                  results: up to 768 rows (a search step's batch) every wave computes one output tile from weights it
 *                  reads itself; beyond, workgroups keep a weight tile in LDS.  RK_OHL_MFMA_DIRECT / _TILED force one. */
typedef struct rk_ohl rk_ohl_t;
int rk_ohl_create(rk_ohl_t **out, const void *d_weight, int w_dtype, const void *d_bias, int H, void *stream);
int rk_ohl_destroy(rk_ohl_t *h);
int rk_ohl_forward(rk_ohl_t *h, const int8_t *d_states, void *d_out, int out_dtype, size_t n, int route, void *stream);
/* Optional epilogue of every later rk_ohl_forward:  y = scale * act(x W^T + b) + shift  per output column -- the
 * activation (model.py:157) and the eval-mode BatchNorm1d (model.py:158-159: scale = gamma / sqrt(var + eps), shift =
 * beta - mean * scale) that follow the layer, computed in float32 on the accumulators.  d_scale / d_shift: H float32
 * values on the device, copied; both null = no affine part.  act RK_OHL_ACT_NONE and null pointers restore the plain layer. */
int rk_ohl_set_epilogue(rk_ohl_t *h, int act, float alpha, const float *d_scale, const float *d_shift, void *stream);

/* The net's other end (model.py:124-125,128-129: policy_net / value_net end in activation -> Linear(K, 12) / Linear(K, 1), and the
 * engines read those 12 + 1 numbers per row): y = act(x) @ W^T + b in ONE pass over x for a last layer of M <= 16 outputs, instead of an
 * elementwise kernel over (n, K) plus a GEMM with a 12-column output.  d_x (n, K) bfloat16 with row stride ldx elements (a multiple
 * of 8, rows 16-byte aligned), d_weight (M, K) bfloat16 row-major, d_bias M bfloat16 or NULL, d_out (n, M) bfloat16 row-major.
 * K is 512, 1024 or 2048; act RK_OHL_ACT_* (alpha for ELU), applied in float32 and rounded to bfloat16 as torch's activation kernel
 * stores it; float32 accumulation (v_dot2c_f32_bf16), the result rounded to bfloat16 (nearest even). */
int rk_tail_linear(const void *d_x, size_t n, int K, size_t ldx, const void *d_weight, const void *d_bias, int M, int act, float alpha,
                   void *d_out, void *stream);

/* as_correct (cube.py:371-380): 686 one-hot int8 (n,288) -> float32 (n,48) of +1/-1. */
int rk_as_correct686(const int8_t *d_states, float *d_out, size_t n, void *stream);

/* ---- batch weighted A* (agents.py:171-413): device-resident open set / closed set ---------------------------
 * Replaces the expand-children loop of AStar.search / expand_batch / relax_seen_states.  The engine owns:
 *   states (cap+1, 20) int8, G, parents, parent_actions            (agents.py:202-205; index 0 unused, root = 1)
 *   an open-addressing hash table state -> index                    (the `indices` dict, agents.py:201)
 *   the open queue as a few sorted runs (capacities 4K, 16K, 64K ... records, K = 12 * expansions): pop = the
 *   globally smallest (cost, index) records in heappop order, push = one multi-way merge into the first run that
 *   holds the result -- O(K log) queue traffic per iteration                       (the heapq, agents.py:185)
 * and EVERY size that varies (states, nodes popped, new states, won, out of budget) in device memory, so that one
 * reference iteration (agents.py:236-252 + 254-331) is two stream-ordered calls around the net forward that PyTorch
 * owns, neither of which synchronises with the host:
 *   rk_astar_step_expand : loop guard `len + 12 N <= max_states` (:236), pop the <= N best nodes (:238-239), 12-child
 *                          fan-out (:277-282), membership + first-occurrence de-duplication in parent-major batch order
 *                          (np.unique semantics, :286-295), append the unseen states with G / parent / action
 *                          (:299-313), goal test of the new states (:321-323), and the one-hot rows of the new states
 *                          (:379) into d_onehot (12 N, 480) -- rows past the number of new states are left untouched;
 *                          out_dtype RK_OH_STATES writes the (12 N, 20) int8 states instead (first layer fused)
 *   rk_astar_step_commit : d_values (12 N) from the net; cost = lambda*G + (-value) in float64 (:383), push (:316-317),
 *                          relaxation of the already-seen children (:326-329, 333-367; skipped once won, as the
 *                          reference returns before relaxing), bookkeeping and the next pop list
 *   rk_astar_status      : synchronises; h_status[8] = done, won, n_states, iterations, open-queue length, index of the
 *                          solved state, error, nodes the next iteration pops
 * Six launches per iteration (plus merge passes when 12 N > 2048), fixed shapes: an iteration can be captured in a
 * hipGraph.  Once `done` (won, out of budget, queue empty) further steps are no-ops.  Results are identical to the
 * reference's arrays (same index numbering, G, parents, parent_actions) whenever the value net returns the same
 * numbers.  An engine handle is not thread-safe: one host thread drives it, on one stream at a time. */
typedef struct rk_astar rk_astar_t;
int rk_astar_create(rk_astar_t **out, size_t capacity, int max_expansions);
int rk_astar_destroy(rk_astar_t *h);
int rk_astar_reset(rk_astar_t *h, const int8_t *h_start_state, double lambda, void *stream);
/* max_states of agents.py:236 (default: the capacity). */
int rk_astar_set_budget(rk_astar_t *h, long long max_states, void *stream);
/* increase_stack_size (agents.py:396-402, called from :273-274): grows the node pool to new_capacity states IN PLACE -- new
 * arrays, device-to-device copies, one kernel that rebuilds the larger hash table; the open queue's sorted runs stay as they
 * are (only the top run, which can hold the whole pool, moves to a larger buffer).  States, G, parents, actions, the open
 * queue, the next pop list and every counter survive: a search that stopped at its loop guard because the pool was its
 * budget continues after rk_astar_set_budget, and ends with the arrays of a search that started in the larger pool.  Call
 * between iterations; synchronises `stream`; a hipGraph captured from this engine must be captured again (it holds the old
 * arrays).  RK_ECAPACITY when the device has no room (the engine is untouched then). */
int rk_astar_grow(rk_astar_t *h, size_t new_capacity, void *stream);
int rk_astar_step_expand(rk_astar_t *h, void *d_onehot, int out_dtype, void *stream);
int rk_astar_step_commit(rk_astar_t *h, const float *d_values, void *stream);
/* The value vectors handed to rk_astar_step_commit / rk_astar_commit / rk_astar_shard_push are float32 (default) or, after
 * rk_astar_set_values_dtype(h, RK_OH_BF16), bfloat16 as a bf16 net emits them (same pointer arguments; bf16 -> float32 is
 * exact, so the cost lambda * G - value is the same number; saves one conversion kernel per iteration).  Sticky until changed. */
int rk_astar_set_values_dtype(rk_astar_t *h, int dtype);
int rk_astar_status(rk_astar_t *h, long long *h_status /* [8] */, void *stream);
/* The same iteration as three calls for hosts that want to feed the net exactly the new states: rk_astar_expand
 * synchronises, h_info = {popped, new, won, solved_index, n_states} (popped == 0: the engine is done -- budget, nothing open --
 * and nothing is pending: there is no iteration to commit); rk_astar_new_states_oh writes the one-hot of the
 * `new` states in index order; rk_astar_commit takes their values. */
int rk_astar_expand(rk_astar_t *h, int n_expand, long long *h_info /* [5] */, void *stream);
int rk_astar_new_states_oh(rk_astar_t *h, void *d_out, int out_dtype, void *stream);
int rk_astar_commit(rk_astar_t *h, const float *d_values, void *stream);
/* Number of stored states (len(agent), agents.py:409-410) and of open-queue entries.  Synchronise. */
long long rk_astar_size(const rk_astar_t *h);
long long rk_astar_open_size(const rk_astar_t *h);
/* Copy rows [first, first+count) of the node arrays to HOST buffers (any may be NULL): states int8 (count,20),
 * G float64, parents int64, parent_actions int64 -- the reference's dtypes (agents.py:390-393). */
int rk_astar_export(rk_astar_t *h, size_t first, size_t count, int8_t *h_states, double *h_G, long long *h_parents,
                    long long *h_parent_actions, void *stream);
/* Action indices from the root to node `index`: the parents are walked on the device (agents.py:244-251).  Returns
 * the path length (>= 0) or a negative error; writes at most `max_len` actions. */
long long rk_astar_path(rk_astar_t *h, long long index, long long *h_actions, size_t max_len, void *stream);
/* Index of a state in the closed set or 0 (the `indices` dict lookup); host state in, synchronises. */
long long rk_astar_lookup(rk_astar_t *h, const int8_t *h_state, void *stream);
/* The open queue in pop order: up to max_len (cost, index) pairs to HOST arrays; returns the count written.
 * Inspection only (gathers every run and sorts on the host). */
long long rk_astar_export_open(rk_astar_t *h, double *h_costs, long long *h_indices, size_t max_len, void *stream);
/* The node indices the NEXT iteration pops, in heappop order (agents.py:238-239). */
long long rk_astar_next_pops(rk_astar_t *h, long long *h_indices, size_t max_len, void *stream);

/* ---- batched A*: S independent searches in lock-step, no host synchronisation inside an iteration -------------
 * Every search is a complete rk_astar_* engine (agents.py:171-413: its own pool, hash table, open queue, counter block);
 * the batch launches the same kernels with a second grid dimension (search), so one iteration of ALL searches is the
 * six launches of one search around one net forward on the padded (S * 12 N, 480) batch -- capturable in a hipGraph:
 *   rk_astarb_step_expand : loop guard + pop + fan-out + membership / first-occurrence / append + goal test for every
 *                           search, then the net's rows of the new states into d_onehot: search s owns rows
 *                           s * 12 N ... (one-hot of out_dtype, or the 20-byte states with RK_OH_STATES); rows past a
 *                           search's new states keep what they held
 *   rk_astarb_step_commit : d_values (S * 12 N) from the net (float32, or bfloat16 after rk_astarb_set_values_dtype);
 *                           cost, push, relaxation, bookkeeping, next pop lists
 *   rk_astarb_status      : synchronises; h_status (S, 7) int64 = done, won (2 = start already solved), n_states,
 *                           iterations, open-queue length, index of the solved state, error
 * h_max_states of rk_astarb_reset: per-search state budgets (null = the capacity). */
typedef struct rk_astarb rk_astarb_t;
int rk_astarb_create(rk_astarb_t **out, int n_searches, size_t capacity_per_search, int max_expansions);
int rk_astarb_destroy(rk_astarb_t *h);
int rk_astarb_reset(rk_astarb_t *h, const int8_t *h_start_states, const long long *h_max_states, double lambda, void *stream);
int rk_astarb_set_values_dtype(rk_astarb_t *h, int dtype, void *stream);
int rk_astarb_step_expand(rk_astarb_t *h, void *d_onehot, int out_dtype, void *stream);
/* The same step with the net's rows COMPACTED across the searches: only the NEW states of every search, one search after
 * the other (each search's first row rounded up to a multiple of 4 rows), and their total copied asynchronously into
 * (pinned) host memory at h_total.  The caller waits for the count (an event recorded behind this call), runs the net on
 * that many rows and commits values laid out the same way: the net never sees a padded row -- what the sequential
 * `AStar` does for a float32 net (agents.py:315, :369-383 evaluate exactly the new states).  One host wait per iteration;
 * not capturable in a hipGraph (the batch size varies). */
int rk_astarb_step_expand_compact(rk_astarb_t *h, void *d_rows, int out_dtype, int *h_total, void *stream);
int rk_astarb_step_commit(rk_astarb_t *h, const float *d_values, void *stream);
int rk_astarb_status(rk_astarb_t *h, long long *h_status, void *stream);
int rk_astarb_export(rk_astarb_t *h, int search, size_t first, size_t count, int8_t *h_states, double *h_G,
                     long long *h_parents, long long *h_parent_actions, void *stream);
long long rk_astarb_path(rk_astarb_t *h, int search, long long index, long long *h_actions, size_t max_len, void *stream);

/* ---- hash-sharded A* across the GPUs of a node (BASELINE config 5; no counterpart in the reference) ------------
 * One engine per GPU/rank holds the states it owns, owner(state) = rk_shard_owner(state, world).  An iteration is two
 * collectives with fixed-size device buffers and no host synchronisation in between:
 *   all-gather   rk_astar_shard_gather_ptr: 8 + N doubles per rank = {pool size, won, solved index, error, candidates,
 *                seconds since the reset on the rank's DEVICE clock (rank 0's decides "out of time"), 0, 0, the rank's N cheapest
 *                open costs ascending, +inf} -- the engine writes all of it, the host nothing: the iteration is capturable
 *   rk_astar_shard_select (gathered)  identical stop decision on every rank (won / budget / a pool could overflow /
 *                time / error / nothing open) and the global top-N by (cost, rank, position); expands this rank's share
 *                and buckets the 32-byte child records by owner (stable) into d_send
 *   all-to-all   equal splits of rk_astar_shard_block_bytes() per peer: {32-byte header = record count, offer count;
 *                12 N records of 32 B; 12 N shortcut offers of 16 B} -- counts travel inside the blocks
 *   rk_astar_shard_insert (d_recv)  applies the offers received (relaxation case 2 of the PREVIOUS iteration on the
 *                parents' owner), then membership / first-occurrence / append / goal test / relaxation case 1 in arrival
 *                order and the one-hot rows of the new states (12 N rows at most: all ranks together pop N nodes)
 *   rk_astar_shard_push  values -> cost, push; builds this iteration's offers into d_send for the next all-to-all;
 *                bookkeeping, next candidates, next all-gather contribution
 * rk_astar_shard_decision (synchronises) lets the host learn the stop decision -- every iteration or every few.
 * After a stop without a win: one more select + all-to-all, then rk_astar_shard_flush applies the pending offers.
 * With world = 1 (send buffer = receive buffer) this reproduces the single-GPU engine exactly.
 * librubiks_amd/solving/sharded.py is the driver. */
int rk_astar_create_sharded(rk_astar_t **out, size_t capacity, int max_expansions, int rank, int world);
int rk_shard_owner(const int8_t *h_state, int world);
long long rk_astar_shard_block_bytes(const rk_astar_t *h);
long long rk_astar_shard_gather_len(const rk_astar_t *h);
void *rk_astar_shard_gather_ptr(rk_astar_t *h);
/* Make the engine write its all-gather contribution into caller memory (8 + N doubles, 8-byte aligned). */
int rk_astar_shard_bind(rk_astar_t *h, void *d_gather);
int rk_astar_shard_reset(rk_astar_t *h, const int8_t *h_start_state, double lambda, void *d_send, void *stream);
int rk_astar_shard_select(rk_astar_t *h, const void *d_gathered, double time_limit, double max_states, void *d_send, void *stream);
/* h_out[8] = {stop reason (0 none, 1 won, 2 budget, 3 capacity, 4 time, 5 nothing open, 6 error), winner rank, winner
 * index, total states, this rank's pops, iterations, this rank's states, the largest error code any rank reported (1 pool
 * capacity, 2 look-back chain, 3 more new states than net rows: rk_astar_shard_push_rows)}. */
int rk_astar_shard_decision(rk_astar_t *h, long long *h_out, void *stream);
int rk_astar_shard_insert(rk_astar_t *h, const void *d_recv, void *d_send, void *d_onehot, int out_dtype, void *stream);
/* Between insert and push: asynchronous copy of this iteration's new-state count (<= 12 N: all ranks together pop at most N
 * nodes) into (pinned) host memory; does not synchronise.  Lets the driver run the net on the rows that exist instead of
 * on the whole padded batch (librubiks_amd/solving/sharded.py). */
int rk_astar_shard_new_count(rk_astar_t *h, int *h_out, void *stream);
int rk_astar_shard_push(rk_astar_t *h, const float *d_values, const void *d_recv, void *d_send, void *stream);
/* The same with the promise the driver can actually keep without a host round trip: d_values holds the net's values of the FIRST
 * `rows` new states only (0 < rows <= 12 N; a rank expects 12 N / world new states, the driver evaluates a fixed number a little
 * above that).  An iteration with more new states than rows sets error 3 in this rank's all-gather contribution: every rank stops
 * together at the next rk_astar_shard_select with stop reason 6 and h_out[7] = 3, and the driver repeats the search with
 * rows = 12 N.  Nothing is copied to the host and nothing waits: insert -> net(rows) -> push_rows is a fixed-shape sequence. */
int rk_astar_shard_push_rows(rk_astar_t *h, const float *d_values, int rows, const void *d_recv, void *d_send, void *stream);
int rk_astar_shard_flush(rk_astar_t *h, const void *d_recv, void *stream);
int rk_astar_shard_clear_send(rk_astar_t *h, void *d_send, int records, int offers, void *stream);
/* h_out = {parent rank, parent index, action} of node `index` on this rank (for the cross-rank path walk). */
int rk_astar_shard_parent(rk_astar_t *h, long long index, long long *h_out, void *stream);
/* Rows [first, first+count) of the parents' owner ranks (int64, HOST): with rk_astar_export's states / G / parents / actions
 * the whole shard of this rank. */
int rk_astar_shard_export_ranks(rk_astar_t *h, size_t first, size_t count, long long *h_parent_ranks, void *stream);

/* ---- transport of the hash-sharded search over RCCL / xGMI, for callers without torch.distributed ----------------
 * No counterpart in the reference (single process); SURVEY.md 8(b) `rk_comm_*`.  One communicator per process (= per GPU,
 * the device current at rk_comm_create).  Rank 0 calls rk_comm_unique_id and hands the RK_COMM_ID_BYTES bytes to the other
 * ranks by any means (file, socket, MPI); every rank then calls rk_comm_create with the same bytes (a collective call).
 * The three transfers are exactly what rk_astar_shard_* needs, on the engine's own fixed-size DEVICE buffers, enqueued on
 * the caller's stream and ordered with the engine's kernels (nothing synchronises):
 *   rk_comm_all_gather   bytes_per_rank from every rank, rank-major, into d_recv (world * bytes_per_rank)
 *   rk_comm_all_to_all   block p of d_send goes to rank p, block q of d_recv comes from rank q (equal blocks: the record
 *                        and offer counts travel inside them, rk_astar_shard_block_bytes); d_send != d_recv
 *   rk_comm_broadcast    in place, from `root` (the three integers of one hop of the path walk)
 * librccl is opened at first use (RK_RCCL_LIB overrides the name); RK_EHIP with RCCL's message if a call fails.
 * librubiks_amd/solving/sharded.py::RcclTransport drives ShardedAStar through these; torch.distributed stays the default. */
#define RK_COMM_ID_BYTES 128
typedef struct rk_comm rk_comm_t;
int rk_comm_unique_id(void *out_128_bytes);
int rk_comm_create(rk_comm_t **out, const void *id_128_bytes, int rank, int world);
int rk_comm_destroy(rk_comm_t *c);
int rk_comm_rank(const rk_comm_t *c);
int rk_comm_world(const rk_comm_t *c);
int rk_comm_all_gather(rk_comm_t *c, const void *d_send, void *d_recv, size_t bytes_per_rank, void *stream);
int rk_comm_all_to_all(rk_comm_t *c, const void *d_send, void *d_recv, size_t bytes_per_peer, void *stream);
int rk_comm_broadcast(rk_comm_t *c, void *d_buf, size_t bytes, int root, void *stream);

/* ---- Monte Carlo tree search (agents.py:415-645): T independent trees, all state in HBM ----------------------
 * Replaces MCTS.expand_leaf / find_leaf for a whole batch of searches (the reference runs one tree at a time).
 * Per tree the engine owns the reference's arrays: states, neighbors (cap,12), leaves, P, V, N, W, L
 * (agents.py:419-427; index 0 unused, root = 1), a hash table state -> index, and the current path
 * (indices_visited / actions_taken, agents.py:474-475).  One wavefront serves one tree.
 * One simulation of every live tree (agents.py:476-490) = two stream-ordered calls around the net forward:
 *   rk_mcts_expand        loop guard `len + 12 <= max_states` (:476); the leaf's 12 children (:513), membership and
 *                         new indices in action order (:517-529), neighbor links both ways and leaf flag (:533-536),
 *                         goal test of ALL children, first solved wins (:540-543).  Leaves the 12 children of every
 *                         tree in a (T*12, 20) buffer.
 *   rk_mcts_children_oh   one-hot of that buffer, (T*12, 480): the fixed-shape net batch (out_dtype RK_OH_STATES: the
 *                         (T*12, 20) int8 states themselves, for a net whose first layer is fused, rk_ohl_*)
 *   rk_mcts_backup_select P, V of the new children (:556-557), W[leaf] = V[neighbors], W[new] = v, max-backup of
 *                         max(v_new) along the path (:559-562), N += 1 once per distinct (node, action) of the path,
 *                         virtual loss cleared (:567-570); then, unless the tree just solved, the next descent
 *                         `find_leaf` (:575-595): U = c P sqrt(sum N)/(1+N), Q = W - L, first argmax, L += nu on both
 *                         ends of the chosen edge.  d_probs (T*12, 12) are softmaxed policy rows, d_values (T*12).
 * Nothing synchronises, shapes are fixed and finished trees are skipped on the device, so a simulation step can be
 * captured in a hipGraph and replayed.  float64 statistics, no fused multiply-add: identical to NumPy. */
typedef struct rk_mcts rk_mcts_t;
int rk_mcts_create(rk_mcts_t **out, int n_trees, size_t capacity_per_tree, size_t max_path);
int rk_mcts_destroy(rk_mcts_t *h);
/* h_start_states (T,20) host; h_max_states (T) host or NULL (= capacity); c exploration constant, nu virtual loss. */
int rk_mcts_reset(rk_mcts_t *h, const int8_t *h_start_states, const long long *h_max_states, double c, double nu, void *stream);
/* One-hot of the T root states (T, 480) for the net; then hand back softmaxed policy (T,12) and value (T). */
int rk_mcts_roots_oh(rk_mcts_t *h, void *d_out, int out_dtype, void *stream);
int rk_mcts_set_root_pv(rk_mcts_t *h, const float *d_probs, const float *d_values, void *stream);
int rk_mcts_expand(rk_mcts_t *h, void *stream);
/* rk_mcts_expand and hipGraphs: whether the path's leaf still has to be expanded is decided ON THE DEVICE (a per-tree word the
 * expanding kernel sets and the backup consumes).  Called eagerly, rk_mcts_expand skips its launch when the previous backup +
 * select call has expanded ahead; called on a stream that is being captured it always records the kernel, which leaves at once
 * for trees that are expanded already -- so a step captured at ANY point (straight after rk_mcts_reset, or in the middle of
 * a search) replays correctly from any state.  A captured step may also leave rk_mcts_expand out altogether once expand-ahead
 * is on and one eager step has run (what MCTSBatch does: one launch less per replay); a backup that finds no expansion pending
 * stops its tree with error 3 in rk_mcts_status instead of backing up stale children.
 * Expand ahead: with sim_limit != 0 the backup + select launch ends by expanding the leaf it has just found (the first half
 * of the NEXT simulation's expand_leaf, agents.py:505-543), and the rk_mcts_expand call that follows it is then a no-op
 * without a launch: one launch and its dependent start-up less per simulation, same order of steps (select, expand, net,
 * backup).  sim_limit > 0: only while the simulation just backed up has a number below sim_limit (so that a search of
 * sim_limit simulations ends exactly where the reference's would); < 0: always; 0 (default): off.  A driver that stops for
 * another reason (time) sets the limit to 0 and runs one more step, which completes the pending expansions. */
int rk_mcts_set_expand_ahead(rk_mcts_t *h, long long sim_limit);
int rk_mcts_children_oh(rk_mcts_t *h, void *d_out, int out_dtype, void *stream);
int rk_mcts_backup_select(rk_mcts_t *h, const float *d_probs, const float *d_values, void *stream);
/* The same step from the net's RAW outputs: d_logits, T*12 rows of 12 logits `logits_stride` elements apart, and d_values,
 * T*12 values `values_stride` elements apart (12 and 1 for separate contiguous heads; 13 and 13 for one (T*12, 13) tensor of
 * merged heads with d_values = d_logits + 12 elements), both float32 (dtype RK_OH_F32) or both bfloat16 (RK_OH_BF16); the
 * softmax of agents.py:551 (exp(x - max) / sum in float32) runs inside the kernel, which saves the conversion, softmax and
 * copy kernels of every simulation. */
int rk_mcts_backup_select_logits(rk_mcts_t *h, const void *d_logits, int logits_stride, const void *d_values, int values_stride,
                                 int dtype, void *stream);
/* The same for the trees first_tree ... first_tree + n_trees - 1 only; d_logits / d_values hold THEIR n_trees*12 rows (row 0 = child
 * 0 of tree first_tree; their children lie at rk_mcts_children() + first_tree*12*20 bytes).  Trees are independent searches
 * (agents.py:415-645 runs one at a time), so a step may advance the batch in parts, on different streams: while one part's
 * latency-bound descent runs, the other part's net forward has the chip (MCTSBatch overlap_halves).  Every tree still sees exactly
 * the reference's sequence select, expand, net, backup. */
int rk_mcts_backup_select_logits_range(rk_mcts_t *h, int first_tree, int n_trees, const void *d_logits, int logits_stride,
                                       const void *d_values, int values_stride, int dtype, void *stream);
/* Device pointer to the (T*12, 20) int8 child states of the pending simulation (what rk_mcts_children_oh encodes): a net
 * whose first layer reads states (rk_ohl_forward) can take them where they lie. */
const int8_t *rk_mcts_children(rk_mcts_t *h);
/* Synchronises.  h_status is (T, 6) int64: done, solved, n_states, simulations, path_len, error (1: path longer than max_path,
 * 2: broken neighbour link, 3: backup without a pending expansion). */
int rk_mcts_status(rk_mcts_t *h, long long *h_status, void *stream);
/* Rows [first, first+count) of one tree's arrays to HOST buffers in the reference's dtypes (any may be NULL):
 * states int8 (count,20), neighbors int64 (count,12), leaves uint8, P/W/L float64 (count,12), V float64, N int64. */
int rk_mcts_export(rk_mcts_t *h, int tree, size_t first, size_t count, int8_t *h_states, long long *h_neighbors,
                   uint8_t *h_leaves, double *h_P, double *h_V, long long *h_N, double *h_W, double *h_L, void *stream);
/* The tree's action queue: the solving actions if it solved (agents.py:483), else the actions of its current
 * descent (agents.py:492).  Also the visited node indices if h_nodes != NULL.  Returns the number of actions. */
long long rk_mcts_path(rk_mcts_t *h, int tree, long long *h_actions, long long *h_nodes, size_t max_len, void *stream);

/* increase_stack_size (agents.py:450-460, called from :503-504): every tree's pool grows to new_capacity states (and the path
 * arrays to new_max_path entries) IN PLACE: new arrays, device-to-device copies, the hash tables rebuilt by one kernel.
 * h_max_states (T, host, nullable = new_capacity) are the new state budgets; a tree that had stopped only at the loop guard
 * `len + 12 <= max_states` (agents.py:476) and has room again continues with the expansion it was about to do.  Synchronises;
 * a hipGraph captured from this engine must be captured again. */
int rk_mcts_grow(rk_mcts_t *h, size_t new_capacity, size_t new_max_path, const long long *h_max_states, void *stream);
/* The graph post-processing of a solved search (agents.py:483-486) on the device, for every tree that has solved:
 * _complete_graph (agents.py:597-611: every leaf linked, both ways, to those of its 12 children that are in the graph -- one
 * launch: fan-out, hash probe, neighbour writes) and the breadth-first search of _shorten_action_queue (agents.py:613-633)
 * from the root to the solved state's index, one workgroup per tree, level by level in the reference's queue order (so the
 * path found is the reference's, not just one of the same length).  Stream-ordered; rk_mcts_export afterwards shows the
 * completed `neighbors`.  rk_mcts_graph_path (synchronises) returns the shortened action queue of `tree` -- its length, or -1
 * when there is none (tree not solved: the queue of rk_mcts_path stands). */
int rk_mcts_search_graph(rk_mcts_t *h, void *stream);
long long rk_mcts_graph_path(rk_mcts_t *h, int tree, long long *h_actions, size_t max_len, void *stream);

/* ---- host-pointer conveniences (allocate scratch, copy, launch, copy back, synchronise) ----
 * What reference code sees when it calls cube.rotate / multi_rotate / is_solved / scramble with NumPy arrays
 * (cube.py:41-56, :85-89, :206-216).  SMALL calls -- inputs and outputs together up to 1 MiB, i.e. about 25 000 20-byte
 * states through rk_multi_rotate_host or 3 800 parents through rk_expand12_host, without h_stats -- go ZERO-COPY: the arrays
 * pass through one page-locked, device-mapped buffer per host thread, the kernel reads and writes it over the bus, and the
 * call costs one launch and one stream synchronisation (one state: 17-21 us from Python against 29-37 us with staged copies,
 * profiles/r04_reference_protocol.json, r04_latency.json).  Larger calls, and calls that want the counters (atomics, kept in
 * device memory), stage through device scratch as before.  Results are the same either way. */
int rk_multi_rotate_host(int repr, const int8_t *h_states, const uint8_t *h_actions, int8_t *h_out,
                         size_t n, void *stream);
int rk_expand12_host(int repr, const int8_t *h_parents, int8_t *h_children, uint8_t *h_solved,
                     long long *h_stats, size_t n, void *stream);
int rk_multi_is_solved_host(int repr, const int8_t *h_states, uint8_t *h_flags, long long *h_stats,
                            size_t n, void *stream);
int rk_apply_sequences_host(int repr, const uint8_t *h_actions, int depth, int games, int with_solved,
                            int only_last, int8_t *h_out, void *stream);
/* cube.as_oh as reference code calls it (cube.py:130-133, :265-277): states from HOST memory, the one-hot written to DEVICE
 * memory (`d_out`, 16-byte aligned, n x 480 or n x 288 elements of `out_dtype`), where the net reads it.  Synchronises. */
int rk_as_oh_host(int repr, const int8_t *h_states, void *d_out, int out_dtype, size_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RUBIKS_HIP_H */
