// Device-resident batch weighted A* (reference: librubiks/solving/agents.py:171-413), sync-free and fused.
//
// What lives in HBM (capacity C states, N = expansions per iteration, K = 12 N children per iteration):
//   states  int8 (C+1, 20)   node pool, index 0 unused, root = 1              (agents.py:202, :390)
//   G       int32 (C+1)      path cost (whole numbers; exported as float64)    (agents.py:203, :393)
//   parents int32, pact uint8, prank uint8 (owner rank of the parent, sharded mode)   (agents.py:204-205)
//   table   uint32 (T)       open-addressing hash table state -> index, T = pow2 >= 2C   (the `indices` dict)
//   mark    uint32 (C+1)     per-node scratch: batch position of a seen state's first occurrence
//   open    the open queue as a small log-structured set of SORTED runs ("levels", capacities 4 K, 16 K, 64 K, ...)
//   ctr     int32[32]        every size that varies: states, popped, new, won, done, budget, iterations, ...
//
// The open queue.  The reference pops with heapq from a heap of (cost, idx) tuples and never re-pushes a node, so
// "the N smallest (cost, idx) pairs in ascending order" is exactly what it expands (agents.py:238-239).  Here every
// level is an array sorted by (cost, idx) with a head pointer.  Pop = the N globally smallest records among the first
// N of every level: each candidate finds its global rank with one binary search per other level (keys are distinct),
// so the pop order is exact.  Push = sort the <= K new records and rank-merge them with levels 0..t into level t,
// where t is the first level whose capacity holds them all -- the classic logarithmic method: a record takes part in
// O(log(|open| / K)) merges, so an iteration moves O(K log) queue bytes instead of re-merging all of |open|
// (round 1 merged the whole queue, 16 B x |open|, every iteration).
//
// One iteration (agents.py:236-252 + 254-331) is SIX launches around the net forward, none of which synchronises:
//   k_expand_lookup   pop list -> parents -> 12 children each (agents.py:277-282), goal flag, membership test and
//                     in-batch first-occurrence election through the hash table (agents.py:286-295)
//   k_append          first_unseen / first_seen flags, order-preserving compaction across workgroups (tickets +
//                     look-back), append with G / parent / action (agents.py:299-313), goal test of the new states
//                     (:321-323), read half of relaxation case 1 (:354)
//   k_new_rows        the net's input: one-hot rows (or, with a fused first layer, the raw states) of the new states, on a
//                     grid as wide as the batch; write half of relaxation case 1 (:357-359)
//   [net forward on the fixed (12 N, 480) batch -- PyTorch]
//   k_records_sort    cost = lambda*G + (-value) in float64 (agents.py:383); merge sort by rank in LDS, one workgroup per chunk:
//                     runs of 256 (K <= 2048) or chunks of 2048 (which k_merge_pass merges into one run only when there
//                     are more than eight of them: K > 16 384)
//   k_queue_insert    the multi-way rank merge described above (heappush, :316-317); read half of case 2 (:362)
//   k_end             write half of case 2 (:365-367), queue bookkeeping, loop guard (:236), and the NEXT pop list
//                     (with levels * N candidates beyond one workgroup's reach the list comes from k_pop_wide, a grid)
// All shapes are fixed by N, so an iteration can be captured in a hipGraph and replayed; the host polls `ctr` now and
// then (rk_astar_status).  Kernels are no-ops once `done` is set (won, out of budget, queue empty).
//
// Order-dependent semantics that are reproduced exactly:
//   * children are generated parent-major / action-minor in pop order                      (agents.py:277-282)
//   * np.unique(..., return_index=True) keeps the FIRST occurrence of a state in batch order; only first
//     occurrences are appended (unseen) or relaxed (seen)                                  (agents.py:291-295)
//   * new indices are handed out in batch order                                            (agents.py:300)
//   * relaxation is two vectorised passes, each reading all of G before writing, the second seeing the first's
//     writes; duplicate targets in the second pass resolve to the LAST one in batch order  (agents.py:353-367)
//
// Batched mode (S searches in lock-step: the kb_* wrappers and rk_astarb_* at the end of this file) and
// hash-sharded mode (one engine per GPU, owner(state) = owner_of(state, world); no counterpart in the reference): see
// the section "hash-sharded search" below and librubiks_amd/solving/sharded.py.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/rubiks_hip.h"
#include "rk_device.h"
#include "rk_error.h"
#include "rk_kernels.h"
#include "rk_search_dev.h"

namespace rk {

enum {
	C_NSTATES = 0, C_NBEFORE, C_NPOP, C_NNEW, C_WON, C_SOLVED, C_DONE, C_BUDGET, C_ITERS, C_ERROR, C_OPEN, C_NCAND, C_NEXP, C_NIN,
	C_NOFF, C_EPOCH, C_TICKET0 = 16, C_TICKET1, C_TICKET2, C_CLOCK0 = 20 /* and 21: the search's start on the device clock */,
	C_DECIDE_ACC = 22, C_DECIDE_TICKET = 23 /* k_shard_decide: pops counted so far, workgroups done */, C_COUNT = 32
};
enum { ERR_NONE = 0, ERR_CAPACITY = 1, ERR_CHAIN = 2, ERR_NET_ROWS = 3 /* sharded: more new states than net rows were evaluated */ };

constexpr int QL = 12;                          // maximal number of queue levels
constexpr int SORT_CHUNK = 2048;                // records sorted per workgroup in LDS (32 KB) when K > 2048
constexpr int SMALL_CHUNK = 256;                // ... and when K <= 2048: up to eight 256-record runs, sorted by eight workgroups in parallel
constexpr int MAX_NEW_RUNS = SORT_CHUNK / SMALL_CHUNK;
enum { Q_HEAD = 0, Q_LEN = 1, Q_CUR = 2, Q_TAKE = 3 };

struct QueueDev {
	Rec *buf[QL][2];
	uint32_t cap[QL];
	int levels;
	int32_t *meta;                              // [4][QL]: head, len, current buffer, records taken by the pending pop
};

struct AstarDev {
	uint32_t mask, cap1;
	int N, K, Kpad, chunk;                      // expansions, 12 N, K rounded up to the sort chunk, sort chunk (256 or 2048)
	int world, rank, KI;                        // sharded: ranks, this rank, incoming child slots = world * K
	double lambda;
	int values_bf16;                            // the net's values arrive as bfloat16 instead of float32 (rk_astar_set_values_dtype)
	uint32_t *states; int32_t *G, *parents; uint8_t *pact, *prank; uint32_t *table, *mark;
	int32_t *ctr;
	QueueDev q;
	int32_t *exp_idx; uint64_t *cand_key; uint8_t *cand_level;       // the pop list (rank order)
	uint32_t *children; uint8_t *solved;
	int32_t *seen; uint32_t *child_slot; uint8_t *flags; int32_t *rank_local;
	uint8_t *newway, *shortcut; int32_t *val1, *val2;
	Rec *rec0, *rec1;
	unsigned long long *chain0, *chain1, *chain2;                    // look-back words {epoch, total} of the in-launch compactions
	uint8_t *hit;
	double *gather_in;                                               // sharded: this rank's all-gather contribution
};

__device__ __forceinline__ int32_t *qmeta(const QueueDev &q, int which) { return q.meta + which * QL; }

// ---- queue merge plan: identical on every thread that computes it from the same meta ----------------------------
struct MergePlan {
	int t;                       // target level (-1: nothing to merge)
	int n_runs;
	const Rec *run[QL + MAX_NEW_RUNS];
	int len[QL + MAX_NEW_RUNS];
	int n_new_runs;              // the first n_new_runs runs are the sorted chunks of the new records
	int total;
	Rec *dst;
};

// How the sorted new records reach the queue insert: runs of 256 (K <= 2048), the 2048-record chunks as they are
// (K <= 16 384: the insert's merge is multi-way, no merge pass needed), or one run (larger K: k_merge_pass merges the
// chunks first).  0 = one run.
__device__ __host__ __forceinline__ int new_chunk_of(int chunk, int Kpad)
{
	return chunk == SMALL_CHUNK ? SMALL_CHUNK : (Kpad <= MAX_NEW_RUNS * SORT_CHUNK ? SORT_CHUNK : 0);
}

// The plan's target level and size alone (the end-of-iteration kernel needs nothing else of it; a whole MergePlan as a local
// of one thread is 272 bytes of SCRATCH memory, written and read back through the memory system: 2 us of that kernel).
// t = -1: nothing to merge.
__device__ __forceinline__ void plan_target(const QueueDev &q, const int32_t *meta, int n_new, int &t_out, int &total)
{
	t_out = -1; total = 0;
	if (n_new <= 0) return;
	int sum = n_new, t = 0;
	for (; t < q.levels; t++) {
		sum += meta[Q_LEN * QL + t] - meta[Q_HEAD * QL + t] - meta[Q_TAKE * QL + t];
		if ((uint32_t)sum <= q.cap[t]) break;
	}
	if (t >= q.levels) t = q.levels - 1;       // cannot happen: the top level holds the whole pool
	t_out = t; total = sum;
}

// meta = pointer to [4][QL] ints (global or LDS copy).  live range of level j after the pending pop: [head+take, len)
// new_chunk: the new records are sorted runs of this length (SMALL_CHUNK), or one run (0)
__device__ __forceinline__ void make_plan(const QueueDev &q, const int32_t *meta, const Rec *newrun, int n_new, int new_chunk, MergePlan &p)
{
	p.t = -1; p.n_runs = 0; p.total = 0; p.dst = nullptr; p.n_new_runs = 0;
	if (n_new <= 0) return;
	int sum = n_new, t = 0;
	for (; t < q.levels; t++) {
		sum += meta[Q_LEN * QL + t] - meta[Q_HEAD * QL + t] - meta[Q_TAKE * QL + t];
		if ((uint32_t)sum <= q.cap[t]) break;
	}
	if (t >= q.levels) t = q.levels - 1;       // cannot happen: the top level holds the whole pool
	p.t = t;
	if (new_chunk > 0) {
		for (int at = 0; at < n_new; at += new_chunk) {
			p.run[p.n_runs] = newrun + at;
			p.len[p.n_runs] = n_new - at < new_chunk ? n_new - at : new_chunk;
			p.n_runs++;
		}
	} else {
		p.run[0] = newrun; p.len[0] = n_new; p.n_runs = 1;
	}
	p.n_new_runs = p.n_runs;
	for (int j = 0; j <= t; j++) {
		const int start = meta[Q_HEAD * QL + j] + meta[Q_TAKE * QL + j];
		const int live = meta[Q_LEN * QL + j] - start;
		if (live > 0) {
			p.run[p.n_runs] = q.buf[j][meta[Q_CUR * QL + j]] + start;
			p.len[p.n_runs] = live;
			p.n_runs++;
		}
	}
	p.total = sum;
	p.dst = q.buf[t][meta[Q_CUR * QL + t] ^ 1];
}

// The same plan written by ONE WAVE (all 64 lanes of it call this): lane j looks at queue level j, lane i at new run i; the
// prefix sums, the target level and the run list come out of a few cross-lane steps instead of one thread's loops over LDS
// and kernel-argument memory (2.1 us at the head of every workgroup of k_queue_insert, measured with the device clock).
// cap[j], buf[2 j + b]: the queue's q.cap[j] and q.buf[j][b] (LDS copies: as kernel-argument memory indexed by the lane they are two
// more round trips inside the plan)
__device__ __forceinline__ void make_plan_wave(int levels, const uint32_t *cap, Rec *const *buf, const int32_t *meta, const Rec *newrun, int n_new, int new_chunk, MergePlan &p, int lane)
{
	static_assert(QL <= 16, "the levels' prefix sums run over 16 lanes");
	if (n_new <= 0) {
		if (lane == 0) { p.t = -1; p.n_runs = 0; p.total = 0; p.dst = nullptr; p.n_new_runs = 0; }
		return;
	}
	const bool is_level = lane < levels;
	const int lv = is_level ? lane : 0;
	const int start = meta[Q_HEAD * QL + lv] + meta[Q_TAKE * QL + lv];
	const int live = is_level ? meta[Q_LEN * QL + lv] - start : 0;
	const int cur = meta[Q_CUR * QL + lv];
	int incl = live;                                                    // inclusive prefix sum over the levels (QL <= 16 lanes)
	#pragma unroll
	for (int off = 1; off < 16; off <<= 1) {
		const int v = __shfl_up(incl, off);
		if (lane >= off) incl += v;
	}
	const int sum = n_new + incl;
	const unsigned long long fits = __ballot(is_level && (uint32_t)sum <= cap[lv]);
	const int t = fits ? __ffsll((long long)fits) - 1 : levels - 1;     // (no level fits: cannot happen, the top level holds the whole pool)
	const int total = __shfl(sum, t);
	const int cur_t = __shfl(cur, t);
	const int n_chunks = new_chunk > 0 ? (n_new + new_chunk - 1) / new_chunk : 1;
	if (lane < n_chunks) {
		const int at = lane * new_chunk;
		p.run[lane] = newrun + at;
		p.len[lane] = new_chunk > 0 ? (n_new - at < new_chunk ? n_new - at : new_chunk) : n_new;
	}
	const bool has = is_level && lane <= t && live > 0;
	const unsigned long long m = __ballot(has);
	if (has) {
		const int at = n_chunks + __popcll(m & ((1ull << lane) - 1ull));
		p.run[at] = buf[2 * lv + cur] + start;
		p.len[at] = live;
	}
	if (lane == 0) {
		p.t = t; p.n_new_runs = n_chunks; p.n_runs = n_chunks + __popcll(m); p.total = total;
		p.dst = buf[2 * t + (cur_t ^ 1)];
	}
}

// ---------------------------------------------------------------------------------------------------------------
__global__ void k_astar_root(AstarDev d, const uint32_t *root, int insert)
{
	const int tid = threadIdx.x;
	if (tid < C_COUNT) d.ctr[tid] = 0;
	if (tid < 4 * QL) d.q.meta[tid] = 0;
	__syncthreads();
	if (tid != 0) return;
	d.ctr[C_BUDGET] = (int32_t)(d.cap1 - 1);
	d.ctr[C_NEXP] = d.N;
	if (!insert) return;                          // sharded: a rank that does not own the root starts empty
	uint32_t s[5];
	load5(root, s);
	#pragma unroll
	for (int j = 0; j < 5; j++) d.states[5 + j] = s[j];
	d.G[1] = 0; d.parents[1] = 0; d.pact[1] = 0; d.prank[1] = (uint8_t)d.rank;
	d.table[hash_state(s) & d.mask] = 1u;
	d.q.buf[0][0][0] = Rec{sortable_key(0.0), 1ull};       // heappush(open_queue, (0, 1))   agents.py:234
	d.q.meta[Q_LEN * QL + 0] = 1;
	d.ctr[C_NSTATES] = 1; d.ctr[C_NBEFORE] = 1; d.ctr[C_OPEN] = 1;
	d.exp_idx[0] = 1; d.cand_key[0] = sortable_key(0.0); d.cand_level[0] = 0;
	d.ctr[C_NCAND] = 1;
	d.ctr[C_NPOP] = 1;
}

// The pop list of the next iteration: the n_cand = min(N, |open|) globally smallest records in ascending order.
// `meta` is the committed queue state (LDS copy); candidates are the first N live records of every level.  When they
// fit (levels * N <= POP_LDS records) the heads are staged in LDS first, so that the binary searches -- a dozen dependent
// reads per candidate -- stay on the CU; `s_heads` may be null (then everything is read from global memory).
constexpr int POP_LDS = 6144;                   // 96 KB of 16-byte records

// Candidates are handled by threads first, first + stride, ...: one workgroup (end-of-iteration kernel) or a grid (k_pop_wide).
// (sharded engines too, round 5: every rank offers its N cheapest whatever the world size -- N = 5 600 in the weak-scaling run at 8 ranks --
//  and one workgroup walking levels * N candidates through global memory was 130 us of a rank's iteration, benchmarks/sharded_sim8.py)
__device__ __forceinline__ bool pop_is_wide(const AstarDev &d) { return d.q.levels * d.N > POP_LDS; }

// cand / n for 0 <= cand < 2^23: a float multiply and a fix-up instead of the 30-instruction integer division (twice per candidate)
__device__ __forceinline__ int div_small(int cand, int n, float inv_n)
{
	int j = (int)((float)cand * inv_n);
	if ((j + 1) * n <= cand) j++;
	else if (j * n > cand) j--;
	return j;
}
__device__ __forceinline__ void pop_select(const AstarDev &d, const int32_t *meta, int n_cand, int n_exp, Rec *s_heads, int first, int stride)
{
	const QueueDev &q = d.q;
	const float inv_n = 1.0f / (float)(n_exp > 0 ? n_exp : 1);
	const bool staged = s_heads != nullptr && q.levels * n_exp <= POP_LDS;
	if (staged) {
		for (int cand = threadIdx.x; cand < q.levels * n_exp; cand += blockDim.x) {
			const int j = div_small(cand, n_exp, inv_n), i = cand - j * n_exp;
			const int head = meta[Q_HEAD * QL + j];
			if (i < meta[Q_LEN * QL + j] - head) s_heads[cand] = q.buf[j][meta[Q_CUR * QL + j]][head + i];
		}
		__syncthreads();
	}
	for (int cand = first; cand < q.levels * n_exp; cand += stride) {
		const int j = div_small(cand, n_exp, inv_n), i = cand - j * n_exp;
		const int head = meta[Q_HEAD * QL + j], live = meta[Q_LEN * QL + j] - head;
		if (i >= live) continue;
		const Rec x = staged ? s_heads[cand] : q.buf[j][meta[Q_CUR * QL + j]][head + i];
		int rank = i;
		for (int j2 = 0; j2 < q.levels; j2++) {
			if (j2 == j) continue;
			const int h2 = meta[Q_HEAD * QL + j2];
			int m = meta[Q_LEN * QL + j2] - h2;
			m = m < n_exp ? m : n_exp;
			if (m > 0) rank += lower_bound_rec(staged ? s_heads + j2 * n_exp : q.buf[j2][meta[Q_CUR * QL + j2]] + h2, m, x);
		}
		if (rank < n_cand) {
			d.exp_idx[rank] = (int32_t)x.idx;
			d.cand_key[rank] = x.key;
			d.cand_level[rank] = (uint8_t)j;
		}
	}
}

// child c of this iteration, recomputed from the pop list (used when a hash slot holds another child's tentative claim)
__device__ __forceinline__ void child_of(const AstarDev &d, const u32x4 *s_act, int c, uint32_t s[5])
{
	const int i = c / 12;
	load5(d.states + (size_t)d.exp_idx[i] * 5, s);
	uint32_t tab[12];
	load_action_table(s_act, (uint32_t)(c - 12 * i), tab);
	move5(s, tab);
}

// membership test + in-batch first-occurrence election through the hash table                    agents.py:286-295
// other(c', buf): state of batch position c' (recomputed or loaded).  Returns through seen / child_slot.
template <typename Other>
__device__ __forceinline__ void lookup_elect(const AstarDev &d, const uint32_t s[5], int c, Other other)
{
	uint32_t slot = hash_state(s) & d.mask;
	for (;;) {
		// (the claim IS the probe: an empty slot -- two children in three at N = 1000 are new states -- costs one round trip to the
		//  table instead of a load and then the compare-and-swap; an occupied slot answers with its occupant either way)
		const uint32_t e = atomicCAS(&d.table[slot], 0u, TENT | (uint32_t)c);
		if (e == 0u) { d.seen[c] = 0; d.child_slot[c] = slot; return; }
		if (e & TENT) {
			uint32_t o[5];
			other((int)(e & ~TENT), o);
			if (((s[0] ^ o[0]) | (s[1] ^ o[1]) | (s[2] ^ o[2]) | (s[3] ^ o[3]) | (s[4] ^ o[4])) == 0u) {
				atomicMin(&d.table[slot], TENT | (uint32_t)c);        // all claimants hold the same state: smallest position wins
				d.seen[c] = 0; d.child_slot[c] = slot;
				return;
			}
		} else if (equal5(s, d.states + (size_t)e * 5)) {
			d.seen[c] = (int32_t)e;
			atomicMin(&d.mark[e], (uint32_t)c);
			return;
		}
		slot = (slot + 1) & d.mask;
	}
}

// pop + gather + 12-child fan-out + goal flag + membership / election: one thread per child
__device__ __forceinline__ void expand_lookup_body(const AstarDev &d)
{
	__shared__ u32x4 s_act[36];
	__shared__ int s_take[QL];
	stage_action_tables(s_act, threadIdx.x);
	if (threadIdx.x < QL) s_take[threadIdx.x] = 0;
	__syncthreads();
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	const int n_pop = d.ctr[C_NPOP];
	if (c < 12 * n_pop) {
		const int i = c / 12, a = c - 12 * i;
		// the queue learns which level the node leaves -- counted per workgroup in LDS: one global atomic per popped node
		// put N operations on a single address (10 000 at the reference's largest N: half of this kernel's time)
		if (a == 0) atomicAdd(&s_take[d.cand_level[i]], 1);
		uint32_t s[5];
		load5(d.states + (size_t)d.exp_idx[i] * 5, s);
		uint32_t tab[12];
		load_action_table(s_act, (uint32_t)a, tab);
		move5(s, tab);
		#pragma unroll
		for (int j = 0; j < 5; j++) d.children[(size_t)c * 5 + j] = s[j];
		d.solved[c] = is_solved5(s) ? 1 : 0;
		lookup_elect(d, s, c, [&](int c2, uint32_t o[5]) { child_of(d, s_act, c2, o); });
	}
	__syncthreads();
	if (threadIdx.x < QL && s_take[threadIdx.x] > 0) atomicAdd(&qmeta(d.q, Q_TAKE)[threadIdx.x], s_take[threadIdx.x]);
}
__global__ __launch_bounds__(256)
void k_expand_lookup(AstarDev d)
{
	expand_lookup_body(d);
}
__global__ __launch_bounds__(256)
void kb_expand_lookup(const AstarDev *__restrict__ devs)
{
	const AstarDev &d = devs[blockIdx.y];              // search blockIdx.y of the batch (rk_astarb_*)
	expand_lookup_body(d);
}


// Sharded mode: incoming child slots.  The receive buffer is `world` blocks of {32-byte header, K records of 32 B,
// K shortcut offers of 16 B}.  Batch position c counts the records in ARRIVAL ORDER -- grouped by sending rank, each group in the
// sender's order -- over the records that are there: c = (records of the peers before `src`) + pos.  All ranks together pop N nodes,
// so a rank receives at most K = 12 N records (and K offers) whatever the world size: every receive-side kernel is a grid over K
// positions.  (Rounds 2-4 used c = src * K + pos over world * K slots, 98 % of them empty at 8 ranks: the receive side's five
// kernels and their ticketed look-back chains ran over 2 100 workgroups where 263 carry records -- benchmarks/sharded_sim8.py.)
// Record: {state[5], parent_idx, g | action<<16 | parent_rank<<24, pad}.
__device__ __forceinline__ size_t shard_block_bytes(int K) { return 32 + (size_t)K * 48; }
__device__ __forceinline__ const uint32_t *shard_hdr(const uint8_t *buf, int K, int peer)
{
	return reinterpret_cast<const uint32_t *>(buf + (size_t)peer * shard_block_bytes(K));
}
constexpr int SHARD_MAX_WORLD = 64;
// Per workgroup, once: s_pref[p] = records (which = 0) or offers (which = 1) of the peers before p, s_pref[world] = their total.
// Contains a barrier: call from ALL threads, before any early exit.
__device__ __forceinline__ void shard_stage_prefix(int *s_pref /* world + 1 */, const uint8_t *buf, int K, int world, int which)
{
	if ((int)threadIdx.x < world) s_pref[threadIdx.x + 1] = (int)shard_hdr(buf, K, (int)threadIdx.x)[which];
	__syncthreads();
	if (threadIdx.x == 0) {
		int sum = 0;
		s_pref[0] = 0;
		for (int p = 0; p < world; p++) { sum += s_pref[p + 1]; s_pref[p + 1] = sum; }
	}
	__syncthreads();
}
__device__ __forceinline__ bool shard_valid(const int *s_pref, int world, int c) { return c < s_pref[world]; }
// (peer, pos) of batch position c < s_pref[world]
__device__ __forceinline__ void shard_locate(const int *s_pref, int world, int c, int &peer, int &pos)
{
	int p = 0;
	while (p + 1 < world && c >= s_pref[p + 1]) p++;
	peer = p; pos = c - s_pref[p];
}
__device__ __forceinline__ const uint32_t *shard_rec(const uint8_t *buf, int K, const int *s_pref, int world, int c)
{
	int peer, pos;
	shard_locate(s_pref, world, c, peer, pos);
	return reinterpret_cast<const uint32_t *>(buf + (size_t)peer * shard_block_bytes(K) + 32 + (size_t)pos * 32);
}

// The net's input rows for the new states of this iteration, (n_new, 480) one-hot or (n_new, 20) raw states, written by a
// grid as wide as the batch (fused into the append kernel it was the longest kernel of the iteration: a handful of
// workgroups writing 2 MB).  Rows past n_new are left untouched.  ELEM_BYTES = 0: no encoding at all, the rows are the
// 20-byte states themselves (for a net whose first layer reads states: rk_ohl_forward, librubiks_amd/oh_linear.py).
template <int ELEM_BYTES, bool SHARDED>
__device__ __forceinline__ void new_rows_body(const AstarDev &d, u32x4 *out, uint32_t one_bits, const uint8_t *recv)
{
	// relaxation case 1, write half (agents.py:357-359): first-seen children that found a shorter way to an old node.
	// It only needs the append kernel's results and touches old nodes, so it rides here, off the critical path after the
	// net.  The reference returns before relaxing once it has won (agents.py:321-323).
	__shared__ int s_pref[SHARDED ? SHARD_MAX_WORLD + 1 : 1];
	if (SHARDED) shard_stage_prefix(s_pref, recv, d.K, d.world, 0);
	if (!d.ctr[C_WON]) {
		const int K = SHARDED ? s_pref[d.world] : 12 * d.ctr[C_NPOP];
		for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < K; c += gridDim.x * blockDim.x) {
			if (!(d.flags[c] & 2) || !d.newway[c]) continue;
			const int32_t t = d.seen[c];
			d.G[t] = d.val1[c];
			if (SHARDED) {
				const uint32_t *r = shard_rec(recv, d.K, s_pref, d.world, c);
				d.pact[t] = (uint8_t)((r[6] >> 16) & 0xFFu);
				d.parents[t] = (int32_t)r[5];
				d.prank[t] = (uint8_t)(r[6] >> 24);
			} else {
				d.pact[t] = (uint8_t)(c % 12);
				d.parents[t] = d.exp_idx[c / 12];
			}
		}
	}
	if (out == nullptr) return;
	const int n_new = d.ctr[C_NNEW];
	const uint32_t *pool = d.states + ((size_t)d.ctr[C_NBEFORE] + 1) * 5;
	if (ELEM_BYTES == 0) {
		uint32_t *dst = reinterpret_cast<uint32_t *>(out);
		for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n_new * 5; q += gridDim.x * blockDim.x) dst[q] = pool[q];
		return;
	}
	constexpr int EB = ELEM_BYTES == 0 ? 4 : ELEM_BYTES;
	constexpr int E = 16 / EB, CPR = 480 / E, CPC = 24 / E;
	const uint8_t *bytes = reinterpret_cast<const uint8_t *>(pool);
	for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n_new * CPR; q += gridDim.x * blockDim.x) {
		const int r = q / CPR, g = q - r * CPR;
		const int cubie = g / CPC, base = (g - cubie * CPC) * E;
		const int rel = (int)bytes[r * STATE_BYTES + cubie] - base;
		u32x4 val = {0u, 0u, 0u, 0u};
		if (ELEM_BYTES == 4) {
			val.x = rel == 0 ? one_bits : 0u; val.y = rel == 1 ? one_bits : 0u;
			val.z = rel == 2 ? one_bits : 0u; val.w = rel == 3 ? one_bits : 0u;
		} else if (rel >= 0 && rel < 8) {
			const uint32_t one = one_bits << (16 * (rel & 1));
			val.x = (rel >> 1) == 0 ? one : 0u; val.y = (rel >> 1) == 1 ? one : 0u;
			val.z = (rel >> 1) == 2 ? one : 0u; val.w = (rel >> 1) == 3 ? one : 0u;
		}
		out[q] = val;
	}
}
template <int ELEM_BYTES, bool SHARDED>
__global__ __launch_bounds__(256)
void k_new_rows(AstarDev d, u32x4 *out, uint32_t one_bits, const uint8_t *recv)
{
	new_rows_body<ELEM_BYTES, SHARDED>(d, out, one_bits, recv);
}
// row_off (batched engines, compact mode): row_off[s] = first row of search s in the shared net batch when only the NEW rows
// of every search are laid out, one search after the other (exclusive prefix of the searches' new-state counts; row_off[S] =
// their total).  Null: search s owns rows s K ... s K + K - 1 (padded).  A row offset must keep 16-byte alignment: offsets
// are rounded up to multiples of 4 rows, which it does for every row type (20-byte states: 80 bytes).
__global__ __launch_bounds__(1024)
void kb_row_offsets(const AstarDev *__restrict__ devs, int S, int32_t *__restrict__ row_off)
{
	__shared__ int s_part[1024];
	__shared__ int s_carry;
	const int tid = threadIdx.x;
	if (tid == 0) s_carry = 0;
	__syncthreads();
	for (int base = 0; base < S; base += 1024) {
		const int s = base + tid;
		const int mine = s < S ? ((devs[s].ctr[C_NNEW] + 3) & ~3) : 0;
		s_part[tid] = mine;
		__syncthreads();
		for (int off = 1; off < 1024; off <<= 1) {                          // inclusive scan (Hillis-Steele): S is small
			const int v = tid >= off ? s_part[tid - off] : 0;
			__syncthreads();
			s_part[tid] += v;
			__syncthreads();
		}
		if (s < S) row_off[s] = s_carry + s_part[tid] - mine;
		__syncthreads();
		if (tid == 1023) s_carry += s_part[1023];
		__syncthreads();
	}
	if (tid == 0) row_off[S] = s_carry;
}

template <int ELEM_BYTES, bool SHARDED>
__global__ __launch_bounds__(256)
void kb_new_rows(const AstarDev *__restrict__ devs, u32x4 *out, uint32_t one_bits, const uint8_t *recv, const int32_t *__restrict__ row_off)
{
	const AstarDev &d = devs[blockIdx.y];              // search blockIdx.y of the batch (rk_astarb_*)
	// this search's rows of the shared net batch: 480 elements (30 * ELEM_BYTES 16-byte chunks) or 20 bytes (5/4 chunks) each
	const size_t first_row = row_off != nullptr ? (size_t)row_off[blockIdx.y] : (size_t)blockIdx.y * d.K;
	if (out != nullptr) out += ELEM_BYTES == 0 ? first_row * 5 / 4 : first_row * 30 * ELEM_BYTES;
	new_rows_body<ELEM_BYTES, SHARDED>(d, out, one_bits, recv);
}


__global__ __launch_bounds__(256)
void k_shard_lookup(AstarDev d, const uint8_t *recv)
{
	__shared__ int s_pref[SHARD_MAX_WORLD + 1];
	shard_stage_prefix(s_pref, recv, d.K, d.world, 0);
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (d.ctr[C_DONE] || !shard_valid(s_pref, d.world, c)) return;
	uint32_t s[5];
	load5(shard_rec(recv, d.K, s_pref, d.world, c), s);
	lookup_elect(d, s, c, [&](int c2, uint32_t o[5]) { load5(shard_rec(recv, d.K, s_pref, d.world, c2), o); });
}

// flags + order-preserving compaction (tickets + look-back) + append + goal test + relaxation case 1 (read half).
// SHARDED = false: child c belongs to popped node c/12, action c%12.   SHARDED = true: child slots of the receive buffer.
template <bool SHARDED>
__device__ __forceinline__ void append_body(const AstarDev &d, const uint8_t *recv)
{
	__shared__ int s_wave[4];
	__shared__ int s_ticket, s_base;
	__shared__ int s_pref[SHARDED ? SHARD_MAX_WORLD + 1 : 1];
	if (SHARDED) shard_stage_prefix(s_pref, recv, d.K, d.world, 0);
	const int b = scan_ticket(&d.ctr[C_TICKET0], &s_ticket);
	const bool live = !d.ctr[C_DONE] || !SHARDED;                       // (single mode: K is 0 once done)
	const int K = !live ? 0 : SHARDED ? s_pref[d.world] : 12 * d.ctr[C_NPOP];
	// the grid covers the largest batch; the tickets past the batch's last workgroup have nothing to add to the scan and leave
	// (a rank of a sharded search receives about 1 / world of it: seven workgroups in eight)
	const int last = K > 0 ? (K - 1) / ASCAN : 0;
	if (b > last) return;
	const int c = b * ASCAN + threadIdx.x;
	const bool valid = c < K;
	int fu = 0, fs = 0;
	int32_t sidx = 0;
	if (valid) {
		sidx = d.seen[c];
		if (sidx == 0) fu = __hip_atomic_load(&d.table[d.child_slot[c]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (TENT | (uint32_t)c);
		else fs = __hip_atomic_load(&d.mark[sidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)c;
	}
	if (c < d.K) d.flags[c] = (uint8_t)(fu | (fs << 1));
	int total;
	const int r = block_rank256(fu != 0, s_wave, &total);
	const uint32_t epoch = (uint32_t)d.ctr[C_EPOCH] + 1u;
	const int base = scan_lookback(d.chain0, b, total, epoch, &s_base);
	const uint32_t n_before = (uint32_t)d.ctr[C_NBEFORE];
	if (b == last && threadIdx.x == 0) {                                // the batch's last ticket holds the grand total
		d.ctr[C_NNEW] = base + total;
		d.ctr[C_NSTATES] = (int32_t)n_before + base + total;
	}
	if (valid) {
		const uint32_t *cs = SHARDED ? shard_rec(recv, d.K, s_pref, d.world, c) : d.children + (size_t)c * 5;
		int32_t p, g;
		uint8_t act, pr = (uint8_t)d.rank;
		if (SHARDED) {
			p = (int32_t)cs[5];
			g = (int32_t)(cs[6] & 0xFFFFu);
			act = (uint8_t)((cs[6] >> 16) & 0xFFu);
			pr = (uint8_t)(cs[6] >> 24);
		} else {
			p = d.exp_idx[c / 12];
			g = d.G[p] + 1;
			act = (uint8_t)(c % 12);
		}
		if (fu) {
			const uint32_t idx = n_before + 1u + (uint32_t)(base + r);
			uint32_t s[5];
			load5(cs, s);
			#pragma unroll
			for (int j = 0; j < 5; j++) d.states[(size_t)idx * 5 + j] = s[j];
			d.G[idx] = g;
			d.parents[idx] = p;
			d.pact[idx] = act;
			d.prank[idx] = pr;
			d.table[d.child_slot[c]] = idx;
			if (is_solved5(s)) { d.ctr[C_WON] = 1; d.ctr[C_SOLVED] = (int32_t)idx; }     // agents.py:321-323
		}
		uint8_t nw = 0;
		if (fs) {
			nw = g < d.G[sidx];                                         // agents.py:354
			d.val1[c] = g;
		}
		d.newway[c] = nw;
	}
}
template <bool SHARDED>
__global__ __launch_bounds__(ASCAN)
void k_append(AstarDev d, const uint8_t *recv)
{
	append_body<SHARDED>(d, recv);
}
template <bool SHARDED>
__global__ __launch_bounds__(ASCAN)
void kb_append(const AstarDev *__restrict__ devs, const uint8_t *recv)
{
	const AstarDev &d = devs[blockIdx.y];              // search blockIdx.y of the batch (rk_astarb_*)
	append_body<SHARDED>(d, recv);
}


// cost record of new state j: cost = lambda * G + (-value), float64, no fused multiply-add (agents.py:380-383).
// Padding records carry distinct maximal keys.
__device__ __forceinline__ Rec cost_record(const AstarDev &d, const float *values, int j, int n_new, uint32_t n_before)
{
	if (j >= n_new) return Rec{~0ull, 0xFFFFFFFF00000000ull + (uint64_t)j};
	const uint32_t idx = n_before + 1u + (uint32_t)j;
	const float val = d.values_bf16 ? __builtin_bit_cast(float, (uint32_t)reinterpret_cast<const uint16_t *>(values)[j] << 16) : values[j];
	const double hv = (double)(-val);
	const double lg = d.lambda * (double)d.G[idx];
	return Rec{sortable_key(lg + hv), (uint64_t)idx};
}

// The new records are sorted in chunks, one workgroup each: CHUNK = 2048 (K > 2048; k_merge_pass then merges the chunks)
// or CHUNK = 256 (K <= 2048: up to eight runs sorted on eight CUs at once, which the queue insert merges directly --
// measured against sorting inside the insert kernel, where every workgroup repeats the sort: 9 us vs 20 us).
// The sort is a merge sort by RANK, two records per thread held in registers: in every pass a record finds its slot as
// its offset plus its lower bound in the sibling run (keys are distinct).  The first seven passes (runs of 1 ... 64)
// stay inside a wave's own 128 records and need no barrier (a wave's LDS operations execute in order); the others take
// two barriers each.  (Ordering the 64-record runs by counting -- 64 broadcast reads per record, no dependent chain --
// was VALU-bound with 16 waves on the CU: 17.9 us per 2048-record chunk.)
// A bitonic network over the same chunk needs 66 barrier steps (36 for 256 records), each moving every record through
// LDS twice: 27.2 us per 2048-record chunk against 16.6 us for this one (9.6 against 8.6 us for 256 records).
// Round 5: the records are sorted by KEY ALONE, as a stable merge.  A new record's index is n_before + 1 + its batch position
// (cost_record; the padding's likewise), so "ties by index" is "ties by position", and a merge of two neighbouring runs is stable
// when a record of the left run counts the right run's keys BELOW its own and a record of the right run the left run's keys
// up to and including its own.  LDS then holds 8-byte keys and 4-byte positions apart: a search step reads 8 bytes instead of a
// 16-byte record (the sort was bound by LDS cycles: 2048 x 66 record reads), the searches have a fixed trip count (runs are
// powers of two), and a thread's two searches advance together.  17.1 -> see profiles/r05_search_legs.json.
__device__ __forceinline__ void rank_pow2_x2(const uint64_t *a0, const uint64_t *a1, int L, const uint64_t x[2], bool incl0, bool incl1, int pos[2])
{
	pos[0] = pos[1] = 0;
	for (int step = L >> 1; step > 0; step >>= 1) {
		const uint64_t k0 = a0[pos[0] + step - 1], k1 = a1[pos[1] + step - 1];
		pos[0] += (incl0 ? k0 <= x[0] : k0 < x[0]) ? step : 0;
		pos[1] += (incl1 ? k1 <= x[1] : k1 < x[1]) ? step : 0;
	}
	const uint64_t k0 = a0[pos[0]], k1 = a1[pos[1]];
	pos[0] += (incl0 ? k0 <= x[0] : k0 < x[0]) ? 1 : 0;
	pos[1] += (incl1 ? k1 <= x[1] : k1 < x[1]) ? 1 : 0;
}
template <int CHUNK>
__device__ __forceinline__ void records_sort_body(const AstarDev &d, const float *values)
{
	__shared__ uint64_t sk[CHUNK];
	__shared__ uint32_t sp[CHUNK];
	constexpr int T = CHUNK / 2;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int n_new = min(d.ctr[C_NNEW], d.Kpad);                       // (never more than the launches were sized for: see shard_push_impl)
	const uint32_t n_before = (uint32_t)d.ctr[C_NBEFORE];
	const int base = blockIdx.x * CHUNK;
	if (base >= n_new) return;                                          // uniform for the workgroup
	const int cnt = n_new - base < CHUNK ? n_new - base : CHUNK;
	int P = 128;                                                        // merge only the power of two that holds the chunk's records
	while (P < cnt) P <<= 1;                                            // (records past it are padding and already in place)
	uint64_t x[2];
	uint32_t xp[2];
	int dst[2], rk[2];
	uint64_t *wk = sk + wv * 128;                                       // this wave's 128 records
	uint32_t *wp = sp + wv * 128;
	#pragma unroll
	for (int t = 0; t < 2; t++) {
		const int e = wv * 128 + t * 64 + lane;
		wk[t * 64 + lane] = cost_record(d, values, base + e, n_new, n_before).key;
		wp[t * 64 + lane] = (uint32_t)e;
	}
	wave_lds_fence();
	#pragma unroll
	for (int lg = 0; lg <= 6; lg++) {                                   // runs of 1, 2, ... 64 -> 128 sorted records per wave, no barrier
		const int L = 1 << lg;
		const int e0 = lane, e1 = 64 + lane;
		const int r0 = e0 >> lg, r1 = e1 >> lg;
		x[0] = wk[e0]; x[1] = wk[e1];
		xp[0] = wp[e0]; xp[1] = wp[e1];
		rank_pow2_x2(wk + ((r0 ^ 1) << lg), wk + ((r1 ^ 1) << lg), L, x, r0 & 1, r1 & 1, rk);
		dst[0] = ((r0 & ~1) << lg) + (e0 & (L - 1)) + rk[0];
		dst[1] = ((r1 & ~1) << lg) + (e1 & (L - 1)) + rk[1];
		wave_lds_fence();                                               // every lane has read before any lane writes
		wk[dst[0]] = x[0]; wp[dst[0]] = xp[0];
		wk[dst[1]] = x[1]; wp[dst[1]] = xp[1];
		wave_lds_fence();
	}
	__syncthreads();
	for (int lg = 7; (1 << lg) < P; lg++) {
		const int L = 1 << lg;
		const int e0 = tid, e1 = tid + T;
		const bool on0 = e0 < P, on1 = e1 < P;                          // (P >= 128 is a multiple of the run length: an active record's sibling run is whole)
		const int r0 = on0 ? e0 >> lg : 0, r1 = on1 ? e1 >> lg : 0;     // idle threads search run 1 from run 0's side: in bounds, result unused
		x[0] = sk[on0 ? e0 : 0]; x[1] = sk[on1 ? e1 : 0];
		xp[0] = sp[on0 ? e0 : 0]; xp[1] = sp[on1 ? e1 : 0];
		rank_pow2_x2(sk + ((r0 ^ 1) << lg), sk + ((r1 ^ 1) << lg), L, x, r0 & 1, r1 & 1, rk);
		dst[0] = ((r0 & ~1) << lg) + (e0 & (L - 1)) + rk[0];
		dst[1] = ((r1 & ~1) << lg) + (e1 & (L - 1)) + rk[1];
		__syncthreads();
		if (on0) { sk[dst[0]] = x[0]; sp[dst[0]] = xp[0]; }
		if (on1) { sk[dst[1]] = x[1]; sp[dst[1]] = xp[1]; }
		__syncthreads();
	}
	#pragma unroll
	for (int t = 0; t < 2; t++) {
		const int e = tid + t * T;
		const int j = base + (int)sp[e];
		const uint64_t idx = j < n_new ? (uint64_t)(n_before + 1u + (uint32_t)j) : 0xFFFFFFFF00000000ull + (uint64_t)j;   // = cost_record's
		d.rec0[base + e] = Rec{sk[e], idx};
	}
}
template <int CHUNK>
__global__ __launch_bounds__(CHUNK / 2)
void k_records_sort(AstarDev d, const float *values)
{
	records_sort_body<CHUNK>(d, values);
}
template <int CHUNK>
__global__ __launch_bounds__(CHUNK / 2)
void kb_records_sort(const AstarDev *__restrict__ devs, const float *values, const int32_t *__restrict__ row_off)
{
	const AstarDev &d = devs[blockIdx.y];              // search blockIdx.y of the batch (rk_astarb_*)
	const size_t first_row = row_off != nullptr ? (size_t)row_off[blockIdx.y] : (size_t)blockIdx.y * d.K;
	values = reinterpret_cast<const float *>(reinterpret_cast<const char *>(values) + first_row * (d.values_bf16 ? 2 : 4));
	records_sort_body<CHUNK>(d, values);
}


// merge neighbouring sorted runs of length L over the padded new-record array (all records distinct)
__device__ __forceinline__ void merge_pass_body(const AstarDev &d, int L, int from)
{
	const int e = blockIdx.x * blockDim.x + threadIdx.x;
	const int n_new = min(d.ctr[C_NNEW], d.Kpad);
	if (e >= d.Kpad || n_new <= SORT_CHUNK) return;                     // a single chunk is already sorted (only launched when K > 2048)
	const int used = ((n_new + SORT_CHUNK - 1) / SORT_CHUNK) * SORT_CHUNK;
	const Rec *src = from ? d.rec1 : d.rec0;
	Rec *dst = from ? d.rec0 : d.rec1;
	if (e >= used) return;
	const int r = e / L, i = e - r * L;
	const int base = (r & ~1) * L, pstart = (r ^ 1) * L;
	int plen = used - pstart;
	plen = plen < 0 ? 0 : (plen > L ? L : plen);
	const Rec x = src[e];
	dst[base + i + lower_bound_rec(src + pstart, plen, x)] = x;
}
__global__ void k_merge_pass(AstarDev d, int L, int from)
{
	merge_pass_body(d, L, from);
}
__global__ void kb_merge_pass(const AstarDev *__restrict__ devs, int L, int from)
{
	const AstarDev &d = devs[blockIdx.y];              // search blockIdx.y of the batch (rk_astarb_*)
	merge_pass_body(d, L, from);
}


// push (agents.py:316-317): multi-way rank merge of the sorted new records with queue levels 0..t into level t's
// other buffer; read half of relaxation case 2 (agents.py:362), which also clears the marks this batch set.
// Every record finds its output slot as its own offset plus one binary search per other run.  With K <= 2048 the new
// records are up to eight 256-record runs: each workgroup stages them in LDS first (32 KB), so those searches -- most of
// them -- never leave the CU.  Runs that stay in global memory -- the queue levels, and with 2048 < K <= 16 384 the
// new records' 2048-record chunks -- get a coarse index in LDS (below).
constexpr int SAMPLES = 256;
constexpr int POOL_RECS = SORT_CHUNK + 3 * SAMPLES;                     // 45 056 B: staged new records + 3 indexed runs, or 11 indexed runs
constexpr int MAX_SAMPLED = POOL_RECS / SAMPLES;
constexpr int MERGE_SLOTS = 8;                                        // lanes per record in the insert's merge (a power of two <= 64)

template <bool SHARDED>
__device__ __forceinline__ void queue_insert_body(const AstarDev &d, int new_in_rec1)
{
	__shared__ MergePlan s_plan;
	__shared__ Rec s_pool[POOL_RECS];
	__shared__ int32_t s_qmeta[4 * QL];
	__shared__ uint32_t s_cap[QL];
	__shared__ Rec *s_buf[2 * QL];
	__shared__ int s_nnew;
	const int nc = new_chunk_of(d.chunk, d.Kpad);
	const bool small = nc == SMALL_CHUNK;
	Rec *const s_newrecs = s_pool;
	const int stride = gridDim.x * blockDim.x;
	if (!SHARDED) {
		// Read half of relaxation case 2 and the marks this batch set: independent of the merge, so it comes first and in the
		// LAST workgroups of the grid, which have no records to merge most of the time (behind the merge in the first workgroups it
		// was 1.7 us at the end of the kernel).  Sharded engines: case 2 travels as offers (k_shard_offers).
		const int K = 12 * d.ctr[C_NPOP];
		const bool won = d.ctr[C_WON] != 0;
		for (int c = (gridDim.x - 1 - blockIdx.x) * blockDim.x + threadIdx.x; c < K; c += stride) {
			uint8_t sc = 0;
			if (d.flags[c] & 2) {
				const int32_t t = d.seen[c];
				d.mark[t] = NO_MARK;
				if (!won) {
					const int32_t g = d.G[t] + 1;
					sc = g < d.G[d.exp_idx[c / 12]];
					d.val2[c] = g;
				}
			}
			d.shortcut[c] = sc;
		}
	}
	if (threadIdx.x < 4 * QL) s_qmeta[threadIdx.x] = d.q.meta[threadIdx.x];     // one parallel load instead of a dependent chain
	else if (threadIdx.x == 64) s_nnew = min(d.ctr[C_NNEW], d.Kpad);            // (in the same round trip, and so are the queue's capacities and buffers)
	else if (threadIdx.x >= 128 && threadIdx.x < 128 + QL) s_cap[threadIdx.x - 128] = d.q.cap[threadIdx.x - 128];
	else if (threadIdx.x >= 192 && threadIdx.x < 192 + 2 * QL) s_buf[threadIdx.x - 192] = d.q.buf[(threadIdx.x - 192) >> 1][(threadIdx.x - 192) & 1];
	__syncthreads();
	__shared__ int s_step[MAX_SAMPLED], s_nsamp[MAX_SAMPLED];
	if (threadIdx.x < 64) {
		const int n_new = s_nnew;
		// after merge passes (one run out of more than eight chunks) the result ping-pongs; otherwise the sorted run(s) are in rec0
		make_plan_wave(d.q.levels, s_cap, s_buf, s_qmeta, (nc == 0 && n_new > SORT_CHUNK && new_in_rec1) ? d.rec1 : d.rec0, n_new, nc, s_plan, threadIdx.x);
		wave_lds_fence();
		// the coarse index's geometry (below), one sampled run per lane: the two divisions per run cost every thread of the
		// workgroup 2 us of instructions when each of them worked them out for all eleven runs
		if (threadIdx.x < MAX_SAMPLED) {
			const int k = threadIdx.x, r = (small ? s_plan.n_new_runs : 0) + k;
			const bool have = s_plan.total > 0 && k < (small ? 3 : MAX_SAMPLED) && r < s_plan.n_runs;
			const int len = have ? s_plan.len[r] : 0, step = (len + SAMPLES - 1) / SAMPLES;
			s_step[k] = step;
			s_nsamp[k] = step > 1 ? (len + step - 1) / step : 0;              // short runs are searched directly
		}
	}
	__syncthreads();
	if (small && s_plan.total > 0) {
		// K <= 2048: the sorted runs of new records (16-32 KB) are staged in LDS, so the merge searches them on the CU
		const int n_new = s_nnew;
		for (int i = threadIdx.x; i < n_new; i += blockDim.x) s_newrecs[i] = d.rec0[i];
		__syncthreads();
		if ((int)threadIdx.x < s_plan.n_new_runs) s_plan.run[threadIdx.x] = s_newrecs + threadIdx.x * SMALL_CHUNK;
		__syncthreads();
	}
	const MergePlan &p = s_plan;
	// Long runs that stay in global memory (the queue levels) get a coarse index in LDS: every `step`-th record, at most
	// SAMPLES per run, so that a binary search spends its first steps on the CU and only log2(step) of them in memory.
	Rec *const s_samples = small ? s_pool + SORT_CHUNK : s_pool;          // [sampled runs][SAMPLES]
	const int sampled_runs = small ? 3 : MAX_SAMPLED;
	const int first_global = small ? p.n_new_runs : 0;
	if (p.total > 0) {
		// The samples of all runs are loaded TOGETHER, then stored (as a loop of load-and-store per run every run's load waited for
		// the one before: 4.2 us in front of every workgroup's merge, measured with the device clock).  Two things make that happen
		// on this compiler: the loaded records are named values (an array of them indexed by the unrolled loop lands in scratch
		// memory), and the loads are GLOBAL loads -- the sampled runs always are global memory, and a flat load counts on the LDS
		// counter too, so the LDS reads of the next run's length would wait for it.  A thread without a sample in run k reads
		// one of the new records' slots (always there, always global memory; another one in every workgroup) and drops it.
		static_assert(MAX_SAMPLED == 11, "one SAMPLE_LOAD / SAMPLE_STORE per sampled run");
		auto sample_load = [&](int k, int &ns) -> u32x4 {
			ns = s_nsamp[k];                                                  // (0 for a run that is not there or not sampled)
			const bool mine = (int)threadIdx.x < ns;
			const Rec *src = mine ? p.run[first_global + k] + (size_t)threadIdx.x * s_step[k] : d.rec0 + (blockIdx.x * 16 + k) % d.Kpad;
			return *(const __attribute__((address_space(1))) u32x4 *)(uintptr_t)src;
		};
		#define SAMPLE_LOAD(k) int n##k; const u32x4 v##k = sample_load(k, n##k);
		#define SAMPLE_STORE(k) if ((int)threadIdx.x < n##k) reinterpret_cast<u32x4 *>(s_samples)[k * SAMPLES + threadIdx.x] = v##k;
		SAMPLE_LOAD(0) SAMPLE_LOAD(1) SAMPLE_LOAD(2) SAMPLE_LOAD(3) SAMPLE_LOAD(4) SAMPLE_LOAD(5)
		SAMPLE_LOAD(6) SAMPLE_LOAD(7) SAMPLE_LOAD(8) SAMPLE_LOAD(9) SAMPLE_LOAD(10)
		SAMPLE_STORE(0) SAMPLE_STORE(1) SAMPLE_STORE(2) SAMPLE_STORE(3) SAMPLE_STORE(4) SAMPLE_STORE(5)
		SAMPLE_STORE(6) SAMPLE_STORE(7) SAMPLE_STORE(8) SAMPLE_STORE(9) SAMPLE_STORE(10)
		#undef SAMPLE_LOAD
		#undef SAMPLE_STORE
		__syncthreads();
	}
	// A record's searches in the other runs are independent: `slots` (up to MERGE_SLOTS) neighbouring lanes share one record, lane `slot`
	// searches runs slot, slot + slots, ..., and the partial ranks are added across the lanes.  One thread per record walked through
	// all the other runs -- 7 to 9 dependent binary searches, 7.9 us of this kernel with one wave per SIMD on half the CUs
	// (device clock) -- where now every lane does one or two and eight times as many waves hide each other's latencies.
	// As many lanes per record as leave the grid ONE pass over the merge (a second pass in some workgroups is the kernel's tail).
	int slots = MERGE_SLOTS;
	while (slots > 1 && (long long)p.total * slots > (long long)gridDim.x * blockDim.x) slots >>= 1;
	const int slot = threadIdx.x & (slots - 1);
	const int per_pass = blockDim.x / slots;
	for (int e0 = blockIdx.x * per_pass; e0 < p.total; e0 += gridDim.x * per_pass) {      // (uniform for the workgroup)
		const int e = e0 + (int)(threadIdx.x / slots);
		const bool on = e < p.total;
		int r = 0, off = on ? e : 0;
		while (off >= p.len[r]) { off -= p.len[r]; r++; }
		const Rec x = p.run[r][off];
		int pos = 0;
		if (on) {
			for (int r2 = slot; r2 < p.n_runs; r2 += slots) {
				if (r2 == r) continue;
				const int k = r2 - first_global;
				if (k >= 0 && k < sampled_runs && s_nsamp[k] > 0) {
					const int sp = lower_bound_rec(s_samples + k * SAMPLES, s_nsamp[k], x);   // first sample >= x
					const int lo = sp > 0 ? (sp - 1) * s_step[k] : 0;
					const int hi = sp < s_nsamp[k] ? sp * s_step[k] : p.len[r2];
					pos += lo + lower_bound_rec(p.run[r2] + lo, hi - lo, x);
				} else {
					pos += lower_bound_rec(p.run[r2], p.len[r2], x);
				}
			}
		}
		for (int m = 1; m < slots; m <<= 1) pos += __shfl_xor(pos, m);
		if (on && slot == 0) p.dst[off + pos] = x;
	}
}
template <bool SHARDED>
__global__ __launch_bounds__(256)
void k_queue_insert(AstarDev d, int new_in_rec1)
{
	queue_insert_body<SHARDED>(d, new_in_rec1);
}
template <bool SHARDED>
__global__ __launch_bounds__(256)
void kb_queue_insert(const AstarDev *__restrict__ devs, int new_in_rec1)
{
	const AstarDev &d = devs[blockIdx.y];              // search blockIdx.y of the batch (rk_astarb_*)
	queue_insert_body<SHARDED>(d, new_in_rec1);
}


// End of an iteration, one workgroup: write half of relaxation case 2 (agents.py:365-367: one thread per expanded
// parent walks its 12 children in order, so the last shortcut child in batch order wins, as NumPy's fancy assignment
// with repeated indices does); queue bookkeeping; loop guard of the next iteration (agents.py:236); next pop list.
template <bool SHARDED>
__device__ __forceinline__ void end_body(const AstarDev &d, int new_in_rec1, int count_iteration, int rows_evaluated = 0)
{
	__shared__ int32_t s_meta[4 * QL], s_old[4 * QL], s_ctr[C_COUNT];
	__shared__ uint32_t s_cap[QL];
	__shared__ int s_ncand, s_nexp;
	__shared__ double s_elapsed;
	__shared__ Rec s_heads[POP_LDS];
	const int tid = threadIdx.x;
	// counters and queue state come in with two parallel loads and go back the same way: the bookkeeping thread below
	// works on LDS only (as dependent global round trips it cost more than every other kernel of the iteration)
	if (tid < C_COUNT) s_ctr[tid] = d.ctr[tid];
	else if (tid >= 64 && tid < 64 + 4 * QL) s_old[tid - 64] = d.q.meta[tid - 64];
	else if (tid >= 128 && tid < 128 + QL) s_cap[tid - 128] = d.q.cap[tid - 128];   // (kernel-argument memory, indexed by the lane: in the same round trip)
	__syncthreads();
	const int n_pop = s_ctr[C_NPOP];
	if (!SHARDED && !s_ctr[C_WON]) {
		for (int i = tid; i < n_pop; i += blockDim.x) {
			// the parent's twelve shortcut flags arrive as three dwords in ONE round trip (as twelve byte loads, each behind a
			// conditional store the compiler must not move it across, they were twelve dependent round trips of this kernel);
			// only the LAST set flag matters: a later assignment to the same parent overwrites an earlier one
			const uint32_t *f = reinterpret_cast<const uint32_t *>(d.shortcut + 12 * (size_t)i);
			const uint32_t w0 = f[0], w1 = f[1], w2 = f[2];
			const int32_t p = d.exp_idx[i];
			int a = -1;
			if (w2) a = 8 + ((31 - __clz((int)w2)) >> 3);
			else if (w1) a = 4 + ((31 - __clz((int)w1)) >> 3);
			else if (w0) a = (31 - __clz((int)w0)) >> 3;
			if (a >= 0) {
				const int c = 12 * i + a;
				d.G[p] = d.val2[c];
				d.pact[p] = (uint8_t)(a ^ 1);              // rev_action                                 cube.py:197-200
				d.parents[p] = d.seen[c];
			}
		}
	}
	// The queue's new state, one level per lane of the first wave (what k_queue_insert's plan of this iteration said: same inputs), then
	// the counters by its first lane.  (One thread walking the levels through LDS was 1.9 us of this kernel, and 4 us while it kept a
	// whole MergePlan -- 272 bytes of scratch memory -- for the two numbers it needs of it.)
	if (tid < 64) {
		const int lane = tid, levels = d.q.levels;
		const int n_new_all = s_ctr[C_NNEW];
		const int n_new = min(n_new_all, d.Kpad);                          // what the sort / insert launches of this iteration covered
		const bool is_level = lane < levels;
		const int lv = is_level ? lane : 0;
		int head = s_old[Q_HEAD * QL + lv] + s_old[Q_TAKE * QL + lv], len = s_old[Q_LEN * QL + lv], cur = s_old[Q_CUR * QL + lv];
		int t = -1, total = 0;                                             // the merge's target level and size (make_plan)
		if (n_new > 0) {
			int incl = is_level ? len - head : 0;
			#pragma unroll
			for (int off = 1; off < 16; off <<= 1) {
				const int v = __shfl_up(incl, off);
				if (lane >= off) incl += v;
			}
			const int sum = n_new + incl;
			const unsigned long long fits = __ballot(is_level && (uint32_t)sum <= s_cap[lv]);
			t = fits ? __ffsll((long long)fits) - 1 : levels - 1;
			total = __shfl(sum, t);
		}
		if (!is_level) { head = 0; len = 0; cur = 0; }
		else {
			if (lane < t) { head = 0; len = 0; }
			else if (lane == t) { head = 0; len = total; cur ^= 1; }
			if (head >= len) { head = 0; len = 0; }
		}
		if (lane < QL) { s_meta[Q_HEAD * QL + lane] = head; s_meta[Q_LEN * QL + lane] = len; s_meta[Q_CUR * QL + lane] = cur; s_meta[Q_TAKE * QL + lane] = 0; }
		int open = len - head;
		#pragma unroll
		for (int m = 1; m < 16; m <<= 1) open += __shfl_xor(open, m);     // (QL <= 16 levels, the other lanes hold 0)
	  if (lane == 0) {
		if (SHARDED) {
			// The driver evaluates the net on a FIXED number of rows (its expected share of the 12 N children plus a margin,
			// librubiks_amd/solving/sharded.py net_rows) instead of waiting for this count on the host.  More new states than rows:
			// the values of the rows beyond were never computed -- an error every rank stops on together at the next decision
			// (the driver then repeats the search with the full-width batch).
			if (rows_evaluated > 0 && n_new_all > rows_evaluated && s_ctr[C_ERROR] == ERR_NONE) s_ctr[C_ERROR] = ERR_NET_ROWS;
			// Rank 0's clock decides "out of time" for everybody; it is the DEVICE's constant 100 MHz clock, started by the first
			// k_end after the reset -- the host writes nothing per iteration, so the iteration can be replayed as a hipGraph.
			unsigned long long t0 = (unsigned long long)(uint32_t)s_ctr[C_CLOCK0] | ((unsigned long long)(uint32_t)s_ctr[C_CLOCK0 + 1] << 32);
			const unsigned long long now = __builtin_amdgcn_s_memrealtime();
			if (t0 == 0) { t0 = now ? now : 1; s_ctr[C_CLOCK0] = (int32_t)(uint32_t)t0; s_ctr[C_CLOCK0 + 1] = (int32_t)(uint32_t)(t0 >> 32); }
			s_elapsed = (double)(now - t0) * 1e-8;
		}
		const bool ran = SHARDED ? s_ctr[C_DONE] == 0 : n_pop > 0;
		if (ran && count_iteration) s_ctr[C_ITERS] += 1;
		s_ctr[C_EPOCH] += 1;                                             // look-back words of this launch sequence expire
		const int n_states = s_ctr[C_NSTATES];
		s_ctr[C_NBEFORE] = n_states;
		s_ctr[C_OPEN] = open;
		s_ctr[C_NNEW] = 0;
		s_ctr[C_NIN] = 0;
		s_ctr[C_TICKET0] = 0; s_ctr[C_TICKET1] = 0; s_ctr[C_TICKET2] = 0;
		const int n_exp = s_ctr[C_NEXP];
		int done = s_ctr[C_DONE];
		if (s_ctr[C_WON]) done = 1;
		if (!SHARDED && (n_states + 12 * n_exp > s_ctr[C_BUDGET] || open == 0)) done = 1;      // loop guard, agents.py:236
		s_ctr[C_DONE] = done;
		const int n_cand = open < n_exp ? open : n_exp;
		s_ctr[C_NCAND] = n_cand;
		if (!SHARDED) s_ctr[C_NPOP] = done ? 0 : n_cand;
		else s_ctr[C_NPOP] = 0;                                          // decided after the all-gather (k_shard_decide)
		s_ncand = n_cand; s_nexp = n_exp;
	  }
	}
	__syncthreads();
	if (tid < C_COUNT) d.ctr[tid] = s_ctr[tid];
	else if (tid >= 64 && tid < 64 + 4 * QL) d.q.meta[tid - 64] = s_meta[tid - 64];
	__syncthreads();
	// With many expansions (levels * N candidates beyond what one workgroup can stage) the selection runs as its own
	// grid-wide kernel right behind this one: on a single workgroup it was 64 us of the iteration at N = 10 000.
	if (!pop_is_wide(d)) pop_select(d, s_meta, s_ncand, s_nexp, s_heads, tid, blockDim.x);
	if (SHARDED) {
		// this rank's contribution to the all-gather: pool size, win flag, solved index, error, then the candidate costs
		__syncthreads();
		double *g = d.gather_in;
		if (tid == 0) {
			g[0] = (double)s_ctr[C_NSTATES]; g[1] = (double)s_ctr[C_WON]; g[2] = (double)s_ctr[C_SOLVED]; g[3] = (double)s_ctr[C_ERROR];
			g[4] = (double)s_ncand; g[5] = s_elapsed; g[6] = 0.0; g[7] = 0.0;   // g[5] = seconds since the reset on this rank's device clock (rank 0's decides)
		}
		// (wide selection: the candidates are not known yet -- k_pop_wide follows, then k_shard_heads writes these)
		if (!pop_is_wide(d)) for (int i = tid; i < d.N; i += blockDim.x) g[8 + i] = i < s_ncand ? key_to_double(d.cand_key[i]) : INFINITY;
	}
}
template <bool SHARDED>
__global__ __launch_bounds__(1024)
void k_end(AstarDev d, int new_in_rec1, int count_iteration, int rows_evaluated = 0)
{
	end_body<SHARDED>(d, new_in_rec1, count_iteration, rows_evaluated);
}
template <bool SHARDED>
__global__ __launch_bounds__(1024)
void kb_end(const AstarDev *__restrict__ devs, int new_in_rec1, int count_iteration)
{
	const AstarDev &d = devs[blockIdx.y];              // search blockIdx.y of the batch (rk_astarb_*)
	end_body<SHARDED>(d, new_in_rec1, count_iteration);
}


// standalone pop selection (only when the host changes the number of expansions between iterations)
__global__ __launch_bounds__(1024)
void k_pop_select_only(AstarDev d, int n_exp)
{
	__shared__ int32_t s_meta[4 * QL];
	__shared__ int s_ncand;
	for (int i = threadIdx.x; i < 4 * QL; i += blockDim.x) s_meta[i] = d.q.meta[i];
	if (threadIdx.x == 0) {
		const int open = d.ctr[C_OPEN];
		const int n_cand = open < n_exp ? open : n_exp;
		d.ctr[C_NEXP] = n_exp;
		d.ctr[C_NCAND] = n_cand;
		int done = d.ctr[C_WON] ? 1 : 0;
		if (d.ctr[C_NSTATES] + 12 * n_exp > d.ctr[C_BUDGET] || open == 0) done = 1;
		d.ctr[C_DONE] = done;
		d.ctr[C_NPOP] = done ? 0 : n_cand;
		s_ncand = n_cand;
	}
	__syncthreads();
	if (!pop_is_wide(d)) pop_select(d, s_meta, s_ncand, n_exp, nullptr, threadIdx.x, blockDim.x);
}

// the pop selection as a grid (single-GPU engines with levels * N > POP_LDS): launched behind k_end / k_pop_select_only
__device__ __forceinline__ void pop_wide_body(const AstarDev &d)
{
	__shared__ int32_t s_meta[4 * QL];
	if (threadIdx.x < 4 * QL) s_meta[threadIdx.x] = d.q.meta[threadIdx.x];
	__syncthreads();
	pop_select(d, s_meta, d.ctr[C_NCAND], d.ctr[C_NEXP], nullptr, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}
__global__ __launch_bounds__(256)
void k_pop_wide(AstarDev d)
{
	pop_wide_body(d);
}
// sharded engines with a wide selection: the rank's candidate costs for the next all-gather, behind k_pop_wide
__global__ __launch_bounds__(256)
void k_shard_heads(AstarDev d)
{
	const int n_cand = d.ctr[C_NCAND];
	double *g = d.gather_in;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.N; i += gridDim.x * blockDim.x) g[8 + i] = i < n_cand ? key_to_double(d.cand_key[i]) : INFINITY;
}
__global__ __launch_bounds__(256)
void kb_pop_wide(const AstarDev *__restrict__ devs)
{
	const AstarDev &d = devs[blockIdx.y];              // search blockIdx.y of the batch (rk_astarb_*)
	pop_wide_body(d);
}


__global__ void k_set_budget(AstarDev d, int budget)
{
	if (threadIdx.x != 0) return;
	d.ctr[C_BUDGET] = budget;
	if (d.world == 1 && !d.ctr[C_WON]) {
		const int done = (d.ctr[C_NSTATES] + 12 * d.ctr[C_NEXP] > budget || d.ctr[C_OPEN] == 0) ? 1 : 0;
		d.ctr[C_DONE] = done;
		d.ctr[C_NPOP] = done ? 0 : d.ctr[C_NCAND];
	}
}


// After a growth (rk_astar_grow): every stored state back into the larger, cleared hash table.  Between iterations no slot is
// tentative, so this is a plain insert of indices 1..n; the slot a state lands in may differ from the one it would have had
// in a pool created at this size, which no result depends on (look-ups compare states).
__global__ __launch_bounds__(256)
void k_astar_rehash(AstarDev d)
{
	const int n = d.ctr[C_NSTATES];
	for (int idx = 1 + blockIdx.x * blockDim.x + threadIdx.x; idx <= n; idx += gridDim.x * blockDim.x) {
		uint32_t s[5];
		load5(d.states + (size_t)idx * 5, s);
		uint32_t slot = hash_state(s) & d.mask;
		while (atomicCAS(&d.table[slot], 0u, (uint32_t)idx) != 0u) slot = (slot + 1) & d.mask;
	}
}

// action indices from the root to node `index` by walking parents on the device (agents.py:244-251)
__global__ void k_astar_walk(AstarDev d, int index, int32_t *out /* [0] = length or -1, then actions root -> node */, int max_len)
{
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	int len = 0, i = index;
	while (i != 1 && len <= (int)d.cap1) { i = d.parents[i]; len++; if (i < 1 || (uint32_t)i >= d.cap1) { out[0] = -1; return; } }
	if (i != 1) { out[0] = -1; return; }
	out[0] = len;
	i = index;
	for (int k = len - 1; k >= 0; k--) {
		if (k < max_len) out[1 + k] = d.pact[i];
		i = d.parents[i];
	}
}

__global__ void k_astar_find(const uint32_t *query, const uint32_t *states, const uint32_t *table, uint32_t mask, int32_t *out)
{
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	uint32_t s[5];
	load5(query, s);
	uint32_t slot = hash_state(s) & mask;
	for (;;) {
		const uint32_t e = table[slot];
		if (e == 0u) { *out = 0; return; }
		if (!(e & TENT) && equal5(s, states + (size_t)e * 5)) { *out = (int32_t)e; return; }
		slot = (slot + 1) & mask;
	}
}

// ---------------------------------------------------------------------------------------------------------------
// Hash-sharded search.  Per iteration, every rank (all in lock-step, nothing below synchronises with the host):
//   [all-gather]       (8 + N) doubles per rank, written by k_end: pool size, win flag, solved index, error, elapsed
//                      time of rank 0, and the rank's N cheapest open costs in ascending order (+inf padded)
//   k_shard_decide     identical on every rank: stop conditions (won, budget, capacity, time, error, nothing open) and
//                      the global top-N by (cost, rank, position) -> how many of its own candidates this rank pops
//   k_shard_expand     expand those, build the 32-byte child records, bucket them by owner into the send blocks with a
//                      stable (batch-order) partition: per-owner ticket + look-back scan across workgroups, all in ONE launch.
//                      The send block of a peer also carries the shortcut offers of the PREVIOUS iteration.
//   [all-to-all]       equal splits of one block per peer: {header, <= K records, <= K offers}; the counts travel in
//                      the header, so there is no count exchange and no host involvement
//   k_shard_offers_in  relaxation case 2 on the parents' owner: evaluate every received offer against G as it stands,
//                      the LAST hit per parent in arrival order wins (NumPy's fancy assignment, agents.py:365-367)
//   k_shard_lookup, k_append<true>, net, k_records_sort<true>, k_queue_insert<true>   as in the single-GPU engine
//   k_shard_offers     a first-seen child whose own G is at least two below its would-be parent's offers the parent a
//                      shortcut: 16-byte records bucketed by the parent's rank into the send blocks (next all-to-all)
//   k_end<true>        queue bookkeeping, candidates of the next iteration, the all-gather contribution
// Deferring the offers to the next all-to-all changes nothing: between the end of an iteration and the next insert
// nobody reads the G of a node that was already expanded.  When a search ends without a win the host flushes the
// pending offers with one more all-to-all so that the final arrays equal the reference's (world = 1).
// ---------------------------------------------------------------------------------------------------------------
enum { D_STOP = 0, D_WINNER_RANK, D_WINNER_IDX, D_TOTAL, D_NPOP, D_ITERS, D_NSTATES, D_ERROR, D_COUNT = 8 };
enum { STOP_NO = 0, STOP_WON = 1, STOP_BUDGET = 2, STOP_CAPACITY = 3, STOP_TIME = 4, STOP_EMPTY = 5, STOP_ERROR = 6 };

// A grid of 256-thread workgroups, one thread per candidate of this rank (round 5; rounds 2-4: ONE workgroup -- at the weak-scaling
// run's N = 5 600 on 8 ranks every thread walked five or six candidates through seven binary searches of thirteen dependent loads:
// 130 us of the iteration, benchmarks/sharded_sim8.py).  Every workgroup derives the stop decision for itself (eight header reads),
// counts its candidates among the global top N, and the last one to finish (a ticket) publishes the pop count and the decision.
__global__ __launch_bounds__(256)
void k_shard_decide(AstarDev d, const double *gathered, double time_limit, double max_states, long long *decision)
{
	__shared__ int s_mine, s_stop, s_winner, s_last;
	__shared__ int s_grank[256];
	__shared__ double s_total, s_maxerr;
	const int tid = threadIdx.x, W = d.world, N = d.N, stride = 8 + N;
	if (tid == 0) {
		s_mine = 0;
		double total = 0, biggest = 0, any_err = 0, max_err = 0;
		int winner = -1, cands = 0;
		for (int r = 0; r < W; r++) {
			const double *g = gathered + (size_t)r * stride;
			total += g[0];
			biggest = g[0] > biggest ? g[0] : biggest;
			if (winner < 0 && g[1] != 0.0) winner = r;
			any_err += g[3];
			max_err = g[3] > max_err ? g[3] : max_err;
			cands += (int)g[4];
		}
		int stop = STOP_NO;
		if (any_err != 0.0) stop = STOP_ERROR;
		else if (winner >= 0) stop = STOP_WON;
		else if (gathered[5] >= time_limit) stop = STOP_TIME;
		// All ranks together pop at most N nodes per iteration (grank < N below), so the search as a whole -- and therefore
		// any one rank's pool -- grows by at most 12 N states: the reference's own guard (agents.py:236), not W times it.
		else if (total + (double)(12 * N) > max_states) stop = STOP_BUDGET;
		else if (biggest + (double)(12 * N) > (double)(d.cap1 - 1)) stop = STOP_CAPACITY;       // a rank's pool could overflow
		else if (cands == 0) stop = STOP_EMPTY;
		s_stop = stop; s_winner = winner; s_total = total; s_maxerr = max_err;
	}
	__syncthreads();
	if (s_stop == STOP_NO) {
		// my candidates' global ranks by (cost, rank, position); the n globally cheapest are popped.  One thread per (candidate, rank)
		// pair: the W - 1 binary searches of a candidate run side by side and meet in an LDS counter, so the dependent chain is ONE
		// search (ten loads), not W - 1 of them back to back (seven searches in a row were 20 us of the select phase at world 8).
		const double *mine = gathered + (size_t)d.rank * stride + 8;
		const int n_mine = (int)gathered[(size_t)d.rank * stride + 4];
		const int cpb = (int)blockDim.x / W;                                // candidates per workgroup
		const int li = tid / W, r = tid - li * W;                            // local candidate, the rank whose list this thread searches
		const int i = blockIdx.x * cpb + li;
		if (tid < cpb) s_grank[tid] = 0;
		__syncthreads();
		if (li < cpb && i < n_mine && r != d.rank) {
			const double x = mine[i];
			const double *o = gathered + (size_t)r * stride + 8;
			int lo = 0, hi = (int)gathered[(size_t)r * stride + 4];
			while (lo < hi) {                                               // ranks below mine win ties, ranks above lose them
				const int mid = (lo + hi) >> 1;
				if (r < d.rank ? o[mid] <= x : o[mid] < x) lo = mid + 1; else hi = mid;
			}
			if (lo) atomicAdd(&s_grank[li], lo);
		}
		__syncthreads();
		const bool in = tid < cpb && blockIdx.x * cpb + tid < n_mine && blockIdx.x * cpb + tid + s_grank[tid] < N;
		const int cnt = __popcll(__ballot(in));
		if ((tid & 63) == 0 && cnt) atomicAdd(&s_mine, cnt);
	}
	__syncthreads();
	if (tid == 0) {
		if (s_mine) atomicAdd(&d.ctr[C_DECIDE_ACC], s_mine);
		__threadfence();
		s_last = atomicAdd(&d.ctr[C_DECIDE_TICKET], 1) == (int)gridDim.x - 1;
	}
	__syncthreads();
	if (s_last && tid == 0) {                                          // every workgroup's count is in: publish
		__threadfence();
		const int pops = s_stop == STOP_NO ? __hip_atomic_load(&d.ctr[C_DECIDE_ACC], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
		d.ctr[C_DECIDE_ACC] = 0; d.ctr[C_DECIDE_TICKET] = 0;
		d.ctr[C_NPOP] = pops;
		decision[D_STOP] = s_stop;
		decision[D_WINNER_RANK] = s_winner;
		decision[D_WINNER_IDX] = s_winner >= 0 ? (long long)gathered[(size_t)s_winner * stride + 2] : 0;
		decision[D_TOTAL] = (long long)s_total;
		decision[D_NPOP] = pops;
		decision[D_ITERS] = d.ctr[C_ITERS];
		decision[D_NSTATES] = d.ctr[C_NSTATES];
		decision[D_ERROR] = (long long)s_maxerr;                         // the largest error code any rank reported (ERR_*)
		if (s_stop != STOP_NO) d.ctr[C_DONE] = 1;
	}
}

// expand this rank's share and bucket the child records by owner, stable, in one launch
__global__ __launch_bounds__(ASCAN)
void k_shard_expand(AstarDev d, uint8_t *send)
{
	__shared__ u32x4 s_act[36];
	__shared__ int s_wave[4];
	__shared__ int s_ticket;
	__shared__ int s_take[QL];
	stage_action_tables(s_act, threadIdx.x);
	if (threadIdx.x < QL) s_take[threadIdx.x] = 0;
	const int b = scan_ticket(&d.ctr[C_TICKET1], &s_ticket);                // (contains the barrier that publishes s_take)
	const int n_pop = d.ctr[C_NPOP], K = 12 * n_pop;
	const int last = K > 0 ? (K - 1) / ASCAN : 0;                        // tickets past the last workgroup with children leave (see append_body)
	if (b > last) return;
	const int c = b * ASCAN + threadIdx.x;
	const bool valid = c < K;
	uint32_t s[5] = {0, 0, 0, 0, 0}, meta6 = 0, p = 0, owner = 0xFFFFFFFFu;
	if (valid) {
		const int i = c / 12, a = c - 12 * i;
		if (a == 0) atomicAdd(&s_take[d.cand_level[i]], 1);                 // per workgroup, see k_expand_lookup
		p = (uint32_t)d.exp_idx[i];
		load5(d.states + (size_t)p * 5, s);
		uint32_t tab[12];
		load_action_table(s_act, (uint32_t)a, tab);
		move5(s, tab);
		meta6 = ((uint32_t)(d.G[p] + 1) & 0xFFFFu) | ((uint32_t)a << 16) | ((uint32_t)d.rank << 24);
		owner = owner_of(s, (uint32_t)d.world);
	}
	const uint32_t epoch = (uint32_t)d.ctr[C_EPOCH] + 1u;
	__shared__ int s_tot[SHARD_MAX_WORLD], s_bases[SHARD_MAX_WORLD];
	const int at = scan_lookback_classes(d.chain1, (int)gridDim.x, b, d.world, owner, epoch, s_wave, s_tot, s_bases);
	if (valid) {
		u32x4 *dst = reinterpret_cast<u32x4 *>(send + (size_t)owner * shard_block_bytes(d.K) + 32 + (size_t)at * 32);
		dst[0] = u32x4{s[0], s[1], s[2], s[3]};
		dst[1] = u32x4{s[4], p, meta6, (uint32_t)c};
	}
	if (b == last && (int)threadIdx.x < d.world)                          // the batch's last ticket holds every owner's grand total
		reinterpret_cast<uint32_t *>(send + (size_t)threadIdx.x * shard_block_bytes(d.K))[0] = (uint32_t)(s_bases[threadIdx.x] + s_tot[threadIdx.x]);
	if (threadIdx.x < QL && s_take[threadIdx.x] > 0) atomicAdd(&qmeta(d.q, Q_TAKE)[threadIdx.x], s_take[threadIdx.x]);
}

// Shortcut offer (16 B): {parent_idx, new G for the parent, index of the child on this rank, this rank | rev(action) << 8}
__device__ __forceinline__ const u32x4 *shard_offer(const uint8_t *buf, int K, const int *s_pref, int world, int o)
{
	int peer, pos;
	shard_locate(s_pref, world, o, peer, pos);
	return reinterpret_cast<const u32x4 *>(buf + (size_t)peer * shard_block_bytes(K) + 32 + (size_t)K * 32 + (size_t)pos * 16);
}

// phase 0: evaluate (read all of G first) and elect the last hit per parent; phase 1: the elected offer applies itself and clears the
// parent's mark (every marked parent has exactly one elected offer, and a mark is only ever set by a hit: nothing else needs clearing --
// rounds 2-4 cleared in a third launch)
__global__ void k_shard_offers_in(AstarDev d, const uint8_t *recv, int phase)
{
	__shared__ int s_pref[SHARD_MAX_WORLD + 1];
	shard_stage_prefix(s_pref, recv, d.K, d.world, 1);                  // offers: header word 1
	const int o = blockIdx.x * blockDim.x + threadIdx.x;
	if (!shard_valid(s_pref, d.world, o)) return;
	const u32x4 r = *shard_offer(recv, d.K, s_pref, d.world, o);
	if (phase == 0) {
		const bool h = (int32_t)r.y < d.G[r.x];
		d.hit[o] = h;
		if (h) atomicMin(&d.mark[r.x], ~(uint32_t)o);
	} else {
		if (!d.hit[o] || __hip_atomic_load(&d.mark[r.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ~(uint32_t)o) return;
		d.G[r.x] = (int32_t)r.y;
		d.parents[r.x] = (int32_t)r.z;
		d.prank[r.x] = (uint8_t)(r.w & 0xFFu);
		d.pact[r.x] = (uint8_t)((r.w >> 8) & 0xFFu);
		d.mark[r.x] = NO_MARK;
	}
}

// receiver side of relaxation case 2: build the offers of this iteration's first-seen children, bucketed by the rank
// that owns the parent (stable), into the offer areas of the send blocks; also clears the marks this batch set.
__global__ __launch_bounds__(ASCAN)
void k_shard_offers(AstarDev d, const uint8_t *recv, uint8_t *send)
{
	__shared__ int s_wave[4];
	__shared__ int s_ticket;
	__shared__ int s_pref[SHARD_MAX_WORLD + 1];
	shard_stage_prefix(s_pref, recv, d.K, d.world, 0);
	const int b = scan_ticket(&d.ctr[C_TICKET2], &s_ticket);
	const int n_in = d.ctr[C_DONE] ? 0 : s_pref[d.world];
	const int last = n_in > 0 ? (n_in - 1) / ASCAN : 0;                  // tickets past the last workgroup with records leave (see append_body)
	if (b > last) return;
	const int c = b * ASCAN + threadIdx.x;
	const bool live = !d.ctr[C_DONE] && !d.ctr[C_WON];
	uint32_t dst_rank = 0xFFFFFFFFu;
	u32x4 rec = {0u, 0u, 0u, 0u};
	if (!d.ctr[C_DONE] && shard_valid(s_pref, d.world, c) && (d.flags[c] & 2)) {
		const uint32_t *r = shard_rec(recv, d.K, s_pref, d.world, c);
		const int32_t t = d.seen[c];
		d.mark[t] = NO_MARK;
		const int32_t g_parent = (int32_t)(r[6] & 0xFFFFu) - 1;
		const int32_t g_new = d.G[t] + 1;
		if (live && g_new < g_parent) {
			dst_rank = r[6] >> 24;
			rec = u32x4{r[5], (uint32_t)g_new, (uint32_t)t, (uint32_t)d.rank | ((((r[6] >> 16) & 0xFFu) ^ 1u) << 8)};
		}
	}
	const uint32_t epoch = (uint32_t)d.ctr[C_EPOCH] + 1u;
	__shared__ int s_tot[SHARD_MAX_WORLD], s_bases[SHARD_MAX_WORLD];
	const int at = scan_lookback_classes(d.chain2, (int)gridDim.x, b, d.world, dst_rank, epoch, s_wave, s_tot, s_bases);
	if (dst_rank < (uint32_t)d.world)
		*reinterpret_cast<u32x4 *>(send + (size_t)dst_rank * shard_block_bytes(d.K) + 32 + (size_t)d.K * 32 + (size_t)at * 16) = rec;
	if (b == last && (int)threadIdx.x < d.world)
		reinterpret_cast<uint32_t *>(send + (size_t)threadIdx.x * shard_block_bytes(d.K))[1] = (uint32_t)(s_bases[threadIdx.x] + s_tot[threadIdx.x]);
}

// zero the counts of every send block (after a flush, or before the first iteration)
__global__ void k_shard_clear_send(AstarDev d, uint8_t *send, int what)
{
	const int w = threadIdx.x;
	if (w >= d.world) return;
	uint32_t *h = reinterpret_cast<uint32_t *>(send + (size_t)w * shard_block_bytes(d.K));
	if (what & 1) h[0] = 0;
	if (what & 2) h[1] = 0;
}

}  // namespace rk

using namespace rk;

struct rk_astar {
	AstarDev d{};
	size_t cap = 0;
	int max_exp = 0;
	uint32_t *root_dev = nullptr;
	int32_t *walk = nullptr;
	long long *decision = nullptr;
	int n_exp = 0;                // expansions of the pending / next iteration (host view)
	bool ready = false, pending = false;
	bool budget_explicit = false; // rk_astar_set_budget was called since the last reset
	int last_n_new = 0, last_n_before = 0;    // sizes reported by the last rk_astar_expand
	int32_t *ctr_host = nullptr;  // page-locked landing place of the counter block: a status poll is one direct copy, no staging
	std::vector<void *> allocs;
};

namespace {

template <typename T>
int dev_alloc(rk_astar *h, T **p, size_t count)
{
	void *q = nullptr;
	RK_HIP(hipMalloc(&q, count * sizeof(T) + 64));
	h->allocs.push_back(q);
	*p = static_cast<T *>(q);
	return RK_OK;
}

inline unsigned blocks(size_t n, unsigned per = 256) { return (unsigned)((n + per - 1) / per); }
inline bool pop_is_wide_host(const AstarDev &d) { return d.q.levels * d.N > POP_LDS; }

constexpr int WALK_MAX = 1 << 16;

int read_ctr(rk_astar *h, int32_t *out, hipStream_t st)
{
	int32_t *dst = h->ctr_host != nullptr ? h->ctr_host : out;      // (pageable memory costs a staged copy per poll)
	RK_HIP(hipMemcpyAsync(dst, h->d.ctr, C_COUNT * sizeof(int32_t), hipMemcpyDeviceToHost, st));
	RK_HIP(hipStreamSynchronize(st));
	if (dst != out) memcpy(out, dst, C_COUNT * sizeof(int32_t));
	return RK_OK;
}

// records an iteration is expected to push: what the open queue's level capacities are multiples of
static inline int queue_inflow(const AstarDev &d) { return d.world == 1 ? d.K : (d.K + d.world - 1) / d.world; }

// sharded engines whose levels * N candidates do not fit one workgroup's LDS: the selection as a grid, then the candidate costs of the
// next all-gather (k_end<true> has written the eight header doubles)
static void launch_shard_wide_selection(const AstarDev &d, hipStream_t st)
{
	if (!pop_is_wide_host(d)) return;
	hipLaunchKernelGGL(k_pop_wide, dim3(blocks((size_t)d.q.levels * d.N)), dim3(256), 0, st, d);
	hipLaunchKernelGGL(k_shard_heads, dim3(blocks((size_t)d.N)), dim3(256), 0, st, d);
}

template <bool SHARDED>
void launch_append(rk_astar *h, const uint8_t *recv, void *d_onehot, int out_dtype, hipStream_t st)
{
	const AstarDev &d = h->d;
	const size_t kin = (size_t)d.K;                                    // (sharded too: at most K records arrive, see shard_stage_prefix)
	hipLaunchKernelGGL((k_append<SHARDED>), dim3(blocks(kin, ASCAN)), dim3(ASCAN), 0, st, d, recv);
	// the net's input rows (about one 16-byte store per thread: the grid covers the largest possible batch) + relaxation 1
	const size_t chunks = d_onehot == nullptr ? kin : kin * (out_dtype == RK_OH_F32 ? 120 : out_dtype == RK_OH_STATES ? 2 : 60);
	const unsigned grid = std::min<unsigned>(blocks(chunks), 8192u);
	if (d_onehot != nullptr && out_dtype == RK_OH_F32)
		hipLaunchKernelGGL((k_new_rows<4, SHARDED>), dim3(grid), dim3(256), 0, st, d, (u32x4 *)d_onehot, 0x3F800000u, recv);
	else if (d_onehot != nullptr && out_dtype == RK_OH_STATES)
		hipLaunchKernelGGL((k_new_rows<0, SHARDED>), dim3(grid), dim3(256), 0, st, d, (u32x4 *)d_onehot, 0u, recv);
	else
		hipLaunchKernelGGL((k_new_rows<2, SHARDED>), dim3(grid), dim3(256), 0, st, d, (u32x4 *)d_onehot, out_dtype == RK_OH_F16 ? 0x3C00u : 0x3F80u, recv);
}

// records + sort + merge passes + queue insert + end of iteration; returns through the launches only
template <bool SHARDED>
int launch_commit(rk_astar *h, const float *d_values, const uint8_t *recv, hipStream_t st, const AstarDev *geometry = nullptr)
{
	const AstarDev &d = geometry ? *geometry : h->d;                    // (sharded: the sort geometry of this iteration, see shard_push_impl)
	int from = 0;
	if (d.chunk == SMALL_CHUNK) {
		hipLaunchKernelGGL((k_records_sort<SMALL_CHUNK>), dim3(d.Kpad / SMALL_CHUNK), dim3(SMALL_CHUNK / 2), 0, st, d, d_values);
	} else {
		hipLaunchKernelGGL((k_records_sort<SORT_CHUNK>), dim3(d.Kpad / SORT_CHUNK), dim3(SORT_CHUNK / 2), 0, st, d, d_values);
		// up to eight chunks go to the queue insert as they are (its merge is multi-way); more are merged into one run first
		for (int L = SORT_CHUNK; L < d.Kpad && new_chunk_of(d.chunk, d.Kpad) == 0; L <<= 1) {
			hipLaunchKernelGGL(k_merge_pass, dim3(blocks(d.Kpad)), dim3(256), 0, st, d, L, from);
			from ^= 1;
		}
	}
	// The merge's size is decided on the device (it is the new records most of the time and a whole queue level now and
	// then); workgroups beyond it leave after the plan.  With 32 workgroups at N = 100 the occasional level merge (up to
	// the whole open set) ran 32 records per thread, each a chain of dependent binary searches: 310 us spikes in round 2.
	static const unsigned min_grid = [] { const char *e = std::getenv("RK_INSERT_MIN_GRID"); return e ? (unsigned)std::atoi(e) : 512u; }();
	// (512 workgroups, two per CU, also are what the usual merge at N = 1000 -- 30 000 records -- runs fastest on: 256 / 384 / 512 / 768 /
	//  1024 workgroups: 14.3 / 13.5 / 12.2 / 12.8 / 16.8 us, every workgroup pays the plan and the coarse index before it merges)
	const unsigned grid = std::min<unsigned>(1024u, std::max<unsigned>(blocks((size_t)d.Kpad * 4), min_grid));
	hipLaunchKernelGGL((k_queue_insert<SHARDED>), dim3(grid), dim3(256), 0, st, d, from);
	if (!SHARDED) {
		hipLaunchKernelGGL((k_end<false>), dim3(1), dim3(1024), 0, st, d, from, 1, 0);
		if (pop_is_wide_host(d)) hipLaunchKernelGGL(k_pop_wide, dim3(blocks((size_t)d.q.levels * d.N)), dim3(256), 0, st, d);
	}
	(void)recv;
	return from;
}

}  // namespace

extern "C" {

static int astar_create_impl(rk_astar_t **out, size_t capacity, int max_expansions, int rank, int world)
{
	if (!out) return fail(RK_EINVAL, "rk_astar_create: null out pointer");
	if (capacity < 2 || capacity > 0x3FFFFFF0ull) return fail(RK_EINVAL, "rk_astar_create: capacity %zu out of range", capacity);
	if (max_expansions < 1 || max_expansions > (1 << 22)) return fail(RK_EINVAL, "rk_astar_create: max_expansions %d out of range", max_expansions);
	if (world < 1 || world > 64 || rank < 0 || rank >= world) return fail(RK_EINVAL, "rk_astar_create: rank %d / world %d out of range", rank, world);
	rk_astar *h = new rk_astar();
	h->cap = capacity;
	h->max_exp = max_expansions;
	AstarDev &d = h->d;
	d.rank = rank; d.world = world;
	d.N = max_expansions; d.K = 12 * max_expansions;
	d.KI = d.K * world;                                   // a rank can receive every rank's children
	const int kin = world == 1 ? d.K : d.KI;
	// The NEW records of an iteration are at most K = 12 N on any rank, whatever the world size (all ranks together pop N nodes), so
	// the sort / merge / insert geometry follows K.  (Rounds 2-4 sized it by the incoming SLOTS, world * K: at world 8 every
	// iteration launched six merge passes over 33 mostly empty chunks where there are at most five chunks of records.)
	d.chunk = d.K <= SORT_CHUNK ? SMALL_CHUNK : SORT_CHUNK;
	d.Kpad = ((d.K + d.chunk - 1) / d.chunk) * d.chunk;
	d.cap1 = (uint32_t)(capacity + 1);
	uint64_t t = 1024;
	while (t < 2 * (uint64_t)capacity + 2) t <<= 1;
	d.mask = (uint32_t)(t - 1);
	const size_t C1 = capacity + 1, KS = (size_t)kin + 64;
	int e = RK_OK;
	#define A(ptr, cnt) if (!e) e = dev_alloc(h, &d.ptr, (cnt))
	A(states, C1 * 5); A(G, C1); A(parents, C1); A(pact, C1); A(prank, C1); A(table, (size_t)t); A(mark, C1);
	A(ctr, C_COUNT);
	A(exp_idx, (size_t)d.N + 8); A(cand_key, (size_t)d.N + 8); A(cand_level, (size_t)d.N + 8);
	A(children, (size_t)d.K * 5 + 64); A(solved, (size_t)d.K + 64);
	A(seen, KS); A(child_slot, KS); A(flags, KS); A(rank_local, 16); A(newway, KS); A(shortcut, KS); A(val1, KS); A(val2, KS);
	A(rec0, (size_t)d.Kpad + 16); A(rec1, (size_t)d.Kpad + 16);
	const size_t n_scan_blocks = (KS + ASCAN - 1) / ASCAN + 1;
	A(chain0, n_scan_blocks); A(chain1, n_scan_blocks * (size_t)world); A(chain2, n_scan_blocks * (size_t)world);
	A(hit, KS); A(gather_in, (size_t)d.N + 16);
	#undef A
	// queue levels: 4 K, 16 K, 64 K, ... records, the top level holds the whole pool.  K here is what an iteration is expected to PUSH:
	// 12 N on one GPU, 12 N / world on a rank of a sharded search (owner = hash).  Rounds 2-4 sized a rank's levels by its incoming
	// SLOTS, world * 12 N: at 8 ranks level 0 held 64 iterations' worth of records and every push rewrote all of it -- 183 us of a
	// rank's iteration in the weak-scaling run (benchmarks/sharded_sim8.py).  The sizes are tuning, not correctness: a push that does
	// not fit level 0 goes to the first level that holds it (make_plan).
	if (!e) e = dev_alloc(h, &d.q.meta, 4 * QL);
	uint64_t c = std::max<uint64_t>(4ull * (uint64_t)queue_inflow(d), 4096ull);
	int levels = 0;
	for (; levels < QL && !e; levels++) {
		const bool top = c >= C1 || levels == QL - 1;
		const uint64_t cap_l = top ? C1 : c;
		d.q.cap[levels] = (uint32_t)cap_l;
		for (int k = 0; k < 2 && !e; k++) e = dev_alloc(h, &d.q.buf[levels][k], (size_t)cap_l + 16);
		if (top) { levels++; break; }
		c *= 4;
	}
	d.q.levels = levels;
	if (!e) e = dev_alloc(h, &h->root_dev, 8);
	if (!e) e = dev_alloc(h, &h->walk, WALK_MAX + 8);
	if (!e) e = dev_alloc(h, &h->decision, D_COUNT);
	if (!e && hipHostMalloc((void **)&h->ctr_host, C_COUNT * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); h->ctr_host = nullptr; }
	if (!e) {
		hipError_t he = hipMemset(d.chain0, 0, n_scan_blocks * sizeof(unsigned long long));
		if (he == hipSuccess) he = hipMemset(d.chain1, 0, n_scan_blocks * world * sizeof(unsigned long long));
		if (he == hipSuccess) he = hipMemset(d.chain2, 0, n_scan_blocks * world * sizeof(unsigned long long));
		if (he != hipSuccess) e = fail(RK_EHIP, "hipMemset failed: %s", hipGetErrorString(he));
	}
	if (e) { rk_astar_destroy(h); return e; }
	*out = h;
	return RK_OK;
}

int rk_astar_create(rk_astar_t **out, size_t capacity, int max_expansions)
{
	return astar_create_impl(out, capacity, max_expansions, 0, 1);
}

int rk_astar_create_sharded(rk_astar_t **out, size_t capacity, int max_expansions, int rank, int world)
{
	return astar_create_impl(out, capacity, max_expansions, rank, world);
}

int rk_astar_destroy(rk_astar_t *h)
{
	if (!h) return RK_OK;
	for (void *p : h->allocs) (void)hipFree(p);
	if (h->ctr_host != nullptr) (void)hipHostFree(h->ctr_host);
	delete h;
	return RK_OK;
}

static int astar_reset_impl(rk_astar_t *h, const int8_t *h_start_state, double lambda, int insert, hipStream_t st)
{
	AstarDev &d = h->d;
	d.lambda = lambda;
	RK_HIP(hipMemsetAsync(d.table, 0, ((size_t)d.mask + 1) * sizeof(uint32_t), st));
	RK_HIP(hipMemsetAsync(d.mark, 0xFF, (h->cap + 1) * sizeof(uint32_t), st));
	// look-back epochs restart with the search: forget the words of the previous one
	const size_t n_scan_blocks = ((size_t)(d.world == 1 ? d.K : d.KI) + 64 + ASCAN - 1) / ASCAN + 1;
	RK_HIP(hipMemsetAsync(d.chain0, 0, n_scan_blocks * sizeof(unsigned long long), st));
	RK_HIP(hipMemsetAsync(d.chain1, 0, n_scan_blocks * d.world * sizeof(unsigned long long), st));
	RK_HIP(hipMemsetAsync(d.chain2, 0, n_scan_blocks * d.world * sizeof(unsigned long long), st));
	RK_HIP(hipMemcpyAsync(h->root_dev, h_start_state, STATE_BYTES, hipMemcpyHostToDevice, st));
	hipLaunchKernelGGL(k_astar_root, dim3(1), dim3(64), 0, st, d, h->root_dev, insert);
	RK_HIP(hipGetLastError());
	RK_HIP(hipStreamSynchronize(st));       // the host buffer may go away after return
	h->n_exp = d.N;
	h->ready = true;
	h->pending = false;
	h->budget_explicit = false;
	return RK_OK;
}

int rk_astar_reset(rk_astar_t *h, const int8_t *h_start_state, double lambda, void *stream)
{
	if (!h || !h_start_state) return fail(RK_EINVAL, "rk_astar_reset: null argument");
	return astar_reset_impl(h, h_start_state, lambda, 1, (hipStream_t)stream);
}

int rk_astar_set_budget(rk_astar_t *h, long long max_states, void *stream)
{
	if (!h || !h->ready) return fail(RK_ESTATE, "rk_astar_set_budget: reset the engine first");
	if (h->pending) return fail(RK_ESTATE, "rk_astar_set_budget: an iteration is pending");
	long long b = max_states < 0 ? 0 : max_states;
	if (b > (long long)h->cap) b = (long long)h->cap;
	hipLaunchKernelGGL(k_set_budget, dim3(1), dim3(64), 0, (hipStream_t)stream, h->d, (int)b);
	RK_HIP(hipGetLastError());
	h->budget_explicit = true;
	return RK_OK;
}


/* increase_stack_size (agents.py:396-402; called from expand_batch, :273-274): the node pool grows to new_capacity states IN
 * PLACE -- new arrays, device-to-device copies of the old ones, one kernel that rebuilds the (larger) hash table; the open
 * queue's levels stay where they are, only the top level (which holds up to the whole pool) and any level above it get
 * larger buffers.  Everything the search holds survives: states, G, parents, actions, the open queue, the pop list of the
 * next iteration and every counter.  A search that had stopped at its loop guard because the pool was its budget goes on
 * after rk_astar_set_budget.  Call between iterations (nothing pending); synchronises `stream`.  A hipGraph captured from
 * this engine holds the old arrays and must be captured again. */
int rk_astar_grow(rk_astar_t *h, size_t new_capacity, void *stream)
{
	if (!h || !h->ready) return fail(RK_ESTATE, "rk_astar_grow: reset the engine first");
	if (h->pending) return fail(RK_ESTATE, "rk_astar_grow: an iteration is pending");
	if (new_capacity <= h->cap) return new_capacity == h->cap ? RK_OK : fail(RK_EINVAL, "rk_astar_grow: %zu is below the current capacity %zu", new_capacity, h->cap);
	if (new_capacity > 0x3FFFFFF0ull) return fail(RK_EINVAL, "rk_astar_grow: capacity %zu out of range", new_capacity);
	hipStream_t st = (hipStream_t)stream;
	const AstarDev old = h->d;
	AstarDev d = old;
	const size_t C1 = new_capacity + 1, C1_old = h->cap + 1;
	d.cap1 = (uint32_t)C1;
	uint64_t t = 1024;
	while (t < 2 * (uint64_t)new_capacity + 2) t <<= 1;
	d.mask = (uint32_t)(t - 1);
	std::vector<void *> fresh, stale;
	auto get = [&](size_t bytes) -> void * { void *q = nullptr; if (hipMalloc(&q, bytes + 64) != hipSuccess) return nullptr; fresh.push_back(q); return q; };
	bool ok = true;
	#define RK_GROW(ptr, type, cnt) do { d.ptr = (type *)get((cnt) * sizeof(type)); ok = ok && d.ptr != nullptr; stale.push_back(old.ptr); } while (0)
	RK_GROW(states, uint32_t, C1 * 5); RK_GROW(G, int32_t, C1); RK_GROW(parents, int32_t, C1); RK_GROW(pact, uint8_t, C1); RK_GROW(prank, uint8_t, C1);
	RK_GROW(table, uint32_t, (size_t)t); RK_GROW(mark, uint32_t, C1);
	#undef RK_GROW
	// queue levels: capacities 4 K, 16 K, ... as at creation; a level whose capacity is unchanged keeps its buffers
	uint64_t c = std::max<uint64_t>(4ull * (uint64_t)queue_inflow(d), 4096ull);
	int levels = 0;
	for (; levels < QL && ok; levels++) {
		const bool top = c >= C1 || levels == QL - 1;
		const uint64_t cap_l = top ? C1 : c;
		if (levels >= old.q.levels || old.q.cap[levels] != (uint32_t)cap_l) {
			d.q.cap[levels] = (uint32_t)cap_l;
			for (int k = 0; k < 2; k++) {
				d.q.buf[levels][k] = (Rec *)get(((size_t)cap_l + 16) * sizeof(Rec));
				ok = ok && d.q.buf[levels][k] != nullptr;
				if (levels < old.q.levels) stale.push_back(old.q.buf[levels][k]);
			}
		}
		if (top) { levels++; break; }
		c *= 4;
	}
	d.q.levels = levels;
	if (!ok) {
		for (void *q : fresh) (void)hipFree(q);
		(void)hipGetLastError();
		return fail(RK_ECAPACITY, "rk_astar_grow: no device memory for a pool of %zu states", new_capacity);
	}
	// the copies and the rehash; an error in here leaves the engine as it was (the new arrays are given back)
	auto fill = [&]() -> hipError_t {
		#define RK_TRY(call) do { const hipError_t e_ = (call); if (e_ != hipSuccess) return e_; } while (0)
		RK_TRY(hipMemcpyAsync(d.states, old.states, C1_old * STATE_BYTES, hipMemcpyDeviceToDevice, st));
		RK_TRY(hipMemcpyAsync(d.G, old.G, C1_old * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
		RK_TRY(hipMemcpyAsync(d.parents, old.parents, C1_old * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
		RK_TRY(hipMemcpyAsync(d.pact, old.pact, C1_old, hipMemcpyDeviceToDevice, st));
		RK_TRY(hipMemcpyAsync(d.prank, old.prank, C1_old, hipMemcpyDeviceToDevice, st));
		RK_TRY(hipMemsetAsync(d.table, 0, (size_t)t * sizeof(uint32_t), st));
		RK_TRY(hipMemsetAsync(d.mark, 0xFF, C1 * sizeof(uint32_t), st));       // between iterations every mark is NO_MARK
		for (int j = 0; j < old.q.levels; j++)                                 // a level that moved: both halves as they are
			for (int k = 0; k < 2; k++)
				if (d.q.buf[j][k] != old.q.buf[j][k])
					RK_TRY(hipMemcpyAsync(d.q.buf[j][k], old.q.buf[j][k], (size_t)old.q.cap[j] * sizeof(Rec), hipMemcpyDeviceToDevice, st));
		hipLaunchKernelGGL(k_astar_rehash, dim3(std::min<unsigned>(blocks(C1_old), 4096u)), dim3(256), 0, st, d);
		RK_TRY(hipGetLastError());
		RK_TRY(hipStreamSynchronize(st));
		#undef RK_TRY
		return hipSuccess;
	};
	if (const hipError_t e = fill(); e != hipSuccess) {
		(void)hipStreamSynchronize(st);                                        // nothing may still write into what is freed next
		for (void *q : fresh) (void)hipFree(q);
		(void)hipGetLastError();
		return fail(RK_EHIP, "rk_astar_grow: %s", hipGetErrorString(e));
	}
	for (void *q : stale) {
		for (size_t i = 0; i < h->allocs.size(); i++) if (h->allocs[i] == q) { h->allocs.erase(h->allocs.begin() + (long)i); break; }
		(void)hipFree(q);
	}
	for (void *q : fresh) h->allocs.push_back(q);
	h->d = d;
	h->cap = new_capacity;
	return RK_OK;
}

int rk_astar_step_expand(rk_astar_t *h, void *d_onehot, int out_dtype, void *stream)
{
	if (!h || !h->ready) return fail(RK_ESTATE, "rk_astar_step_expand: reset the engine first");
	if (h->d.world != 1) return fail(RK_ESTATE, "rk_astar_step_expand: sharded engines use rk_astar_shard_*");
	if (h->pending) return fail(RK_ESTATE, "rk_astar_step_expand: previous iteration not committed");
	if (d_onehot && (reinterpret_cast<uintptr_t>(d_onehot) & 15)) return fail(RK_EINVAL, "rk_astar_step_expand: one-hot buffer must be 16-byte aligned");
	if (out_dtype < RK_OH_F32 || out_dtype > RK_OH_STATES) return fail(RK_EINVAL, "rk_astar_step_expand: unknown dtype %d", out_dtype);
	hipStream_t st = (hipStream_t)stream;
	const AstarDev &d = h->d;
	hipLaunchKernelGGL(k_expand_lookup, dim3(blocks((size_t)12 * h->n_exp)), dim3(256), 0, st, d);
	launch_append<false>(h, nullptr, d_onehot, out_dtype, st);
	RK_HIP(hipGetLastError());
	h->pending = true;
	return RK_OK;
}

int rk_astar_set_values_dtype(rk_astar_t *h, int dtype)
{
	if (!h) return fail(RK_EINVAL, "rk_astar_set_values_dtype: null handle");
	if (dtype != RK_OH_F32 && dtype != RK_OH_BF16) return fail(RK_EINVAL, "rk_astar_set_values_dtype: float32 or bfloat16");
	h->d.values_bf16 = dtype == RK_OH_BF16 ? 1 : 0;
	return RK_OK;
}

int rk_astar_step_commit(rk_astar_t *h, const float *d_values, void *stream)
{
	if (!h || !h->pending) return fail(RK_ESTATE, "rk_astar_step_commit: no pending iteration");
	if (h->d.world != 1) return fail(RK_ESTATE, "rk_astar_step_commit: sharded engines use rk_astar_shard_push");
	if (!d_values) return fail(RK_EINVAL, "rk_astar_step_commit: null values");
	launch_commit<false>(h, d_values, nullptr, (hipStream_t)stream);
	RK_HIP(hipGetLastError());
	h->pending = false;
	return RK_OK;
}

int rk_astar_status(rk_astar_t *h, long long *h_status, void *stream)
{
	if (!h || !h->ready || !h_status) return fail(RK_EINVAL, "rk_astar_status: bad argument");
	int32_t c[C_COUNT];
	if (int e = read_ctr(h, c, (hipStream_t)stream)) return e;
	h_status[0] = c[C_DONE]; h_status[1] = c[C_WON]; h_status[2] = c[C_NSTATES]; h_status[3] = c[C_ITERS];
	h_status[4] = c[C_OPEN]; h_status[5] = c[C_SOLVED]; h_status[6] = c[C_ERROR]; h_status[7] = c[C_NPOP];
	return RK_OK;
}

/* The three-call form of one iteration (expand synchronises and reports the sizes; the host then feeds exactly the new
 * states to the net).  Same kernels as rk_astar_step_*. */
int rk_astar_expand(rk_astar_t *h, int n_expand, long long *h_info, void *stream)
{
	if (!h || !h_info) return fail(RK_EINVAL, "rk_astar_expand: null argument");
	if (!h->ready) return fail(RK_ESTATE, "rk_astar_expand: reset the engine first");
	if (h->d.world != 1) return fail(RK_ESTATE, "rk_astar_expand: sharded engines use rk_astar_shard_*");
	if (h->pending) return fail(RK_ESTATE, "rk_astar_expand: previous expansion not committed");
	if (n_expand < 1 || n_expand > h->max_exp) return fail(RK_EINVAL, "rk_astar_expand: n_expand %d outside 1..%d", n_expand, h->max_exp);
	hipStream_t st = (hipStream_t)stream;
	int32_t c[C_COUNT];
	if (n_expand != h->n_exp) {
		hipLaunchKernelGGL(k_pop_select_only, dim3(1), dim3(1024), 0, st, h->d, n_expand);
		if (pop_is_wide_host(h->d)) hipLaunchKernelGGL(k_pop_wide, dim3(blocks((size_t)h->d.q.levels * h->d.N)), dim3(256), 0, st, h->d);
		h->n_exp = n_expand;
	}
	if (!h->budget_explicit) {
		// a host that never set a state budget learns about a full pool as an error; with a budget the engine's own loop
		// guard has ended the search (done, nothing popped) before the pool can overflow, and this round trip is skipped
		if (int e = read_ctr(h, c, st)) return e;
		if ((size_t)c[C_NSTATES] + 12 * (size_t)c[C_NCAND] > h->cap)
			return fail(RK_ECAPACITY, "rk_astar_expand: %d states + %d children exceed capacity %zu", c[C_NSTATES], 12 * c[C_NCAND], h->cap);
	}
	if (int e = rk_astar_step_expand(h, nullptr, RK_OH_F32, stream)) return e;
	if (int e = read_ctr(h, c, st)) return e;            // the one synchronisation of the iteration (the kernels above leave NPOP alone)
	h->last_n_new = c[C_NNEW];
	h->last_n_before = c[C_NBEFORE];
	if (c[C_NPOP] == 0) {
		// the engine is done (budget, nothing open): the launches above were no-ops, but they drew their tickets.  Close the empty
		// iteration here so that the caller has nothing to commit and the engine can go on later (rk_astar_set_budget after a
		// growth, rk_astar_grow itself): no cost records, no push, no iteration counted, the counters of the launch sequence reset.
		if (int e = rk_astar_step_commit(h, reinterpret_cast<const float *>(h->d.val1), stream)) return e;
	}
	h_info[0] = c[C_NPOP]; h_info[1] = c[C_NNEW]; h_info[2] = c[C_WON]; h_info[3] = c[C_SOLVED]; h_info[4] = c[C_NSTATES];
	return RK_OK;
}

int rk_astar_new_states_oh(rk_astar_t *h, void *d_out, int out_dtype, void *stream)
{
	if (!h || !h->pending) return fail(RK_ESTATE, "rk_astar_new_states_oh: no pending expansion");
	const size_t n_new = (size_t)h->last_n_new, first = (size_t)h->last_n_before + 1;      // as reported by rk_astar_expand
	if (n_new == 0) return RK_OK;
	if (out_dtype == RK_OH_STATES) {              // first layer fused: the new states themselves
		if (!d_out) return fail(RK_EINVAL, "rk_astar_new_states_oh: null output");
		RK_HIP(hipMemcpyAsync(d_out, h->d.states + first * 5, n_new * STATE_BYTES, hipMemcpyDeviceToDevice, (hipStream_t)stream));
		return RK_OK;
	}
	return rk_as_oh(RK_REPR_2024, (const int8_t *)(h->d.states + first * 5), d_out, out_dtype, n_new, stream);
}

int rk_astar_commit(rk_astar_t *h, const float *d_values, void *stream)
{
	if (!h || !h->pending) return fail(RK_ESTATE, "rk_astar_commit: no pending expansion");
	if (h->d.world != 1) return fail(RK_ESTATE, "rk_astar_commit: sharded engines use rk_astar_shard_push");
	if (!d_values) d_values = reinterpret_cast<const float *>(h->d.val1);       // no new states: never read
	return rk_astar_step_commit(h, d_values, stream);
}

/* The pop list of the NEXT iteration (node indices in pop order): what heappop would return (agents.py:238-239). */
long long rk_astar_next_pops(rk_astar_t *h, long long *h_indices, size_t max_len, void *stream)
{
	if (!h || !h->ready) return fail(RK_ESTATE, "rk_astar_next_pops: reset the engine first");
	hipStream_t st = (hipStream_t)stream;
	int32_t c[C_COUNT];
	if (int e = read_ctr(h, c, st)) return e;
	const size_t n = std::min<size_t>((size_t)c[C_NPOP], max_len);
	std::vector<int32_t> idx(n);
	if (n) {
		RK_HIP(hipMemcpyAsync(idx.data(), h->d.exp_idx, n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
		RK_HIP(hipStreamSynchronize(st));
	}
	for (size_t i = 0; i < n; i++) h_indices[i] = idx[i];
	return (long long)n;
}

long long rk_astar_size(const rk_astar_t *hc)
{
	rk_astar_t *h = const_cast<rk_astar_t *>(hc);
	if (!h || !h->ready) return 0;
	int32_t c[C_COUNT];
	if (read_ctr(h, c, nullptr)) return RK_EHIP;
	return c[C_NSTATES];
}

long long rk_astar_open_size(const rk_astar_t *hc)
{
	rk_astar_t *h = const_cast<rk_astar_t *>(hc);
	if (!h || !h->ready) return 0;
	int32_t c[C_COUNT];
	if (read_ctr(h, c, nullptr)) return RK_EHIP;
	return c[C_OPEN];
}

int rk_astar_export(rk_astar_t *h, size_t first, size_t count, int8_t *h_states, double *h_G, long long *h_parents,
                    long long *h_parent_actions, void *stream)
{
	if (!h) return fail(RK_EINVAL, "rk_astar_export: null handle");
	if (first + count > h->cap + 1) return fail(RK_EINVAL, "rk_astar_export: rows %zu..%zu outside the pool", first, first + count);
	if (count == 0) return RK_OK;
	hipStream_t st = (hipStream_t)stream;
	const AstarDev &d = h->d;
	std::vector<int32_t> g, p;
	std::vector<uint8_t> a;
	if (h_states) RK_HIP(hipMemcpyAsync(h_states, d.states + first * 5, count * STATE_BYTES, hipMemcpyDeviceToHost, st));
	if (h_G) { g.resize(count); RK_HIP(hipMemcpyAsync(g.data(), d.G + first, count * sizeof(int32_t), hipMemcpyDeviceToHost, st)); }
	if (h_parents) { p.resize(count); RK_HIP(hipMemcpyAsync(p.data(), d.parents + first, count * sizeof(int32_t), hipMemcpyDeviceToHost, st)); }
	if (h_parent_actions) { a.resize(count); RK_HIP(hipMemcpyAsync(a.data(), d.pact + first, count, hipMemcpyDeviceToHost, st)); }
	RK_HIP(hipStreamSynchronize(st));
	for (size_t i = 0; i < count; i++) {
		if (h_G) h_G[i] = (double)g[i];
		if (h_parents) h_parents[i] = p[i];
		if (h_parent_actions) h_parent_actions[i] = a[i];
	}
	return RK_OK;
}

long long rk_astar_path(rk_astar_t *h, long long index, long long *h_actions, size_t max_len, void *stream)
{
	if (!h || !h->ready) return fail(RK_EINVAL, "rk_astar_path: null handle");
	if (index < 1 || (size_t)index > h->cap) return fail(RK_EINVAL, "rk_astar_path: index %lld outside 1..%zu", index, h->cap);
	hipStream_t st = (hipStream_t)stream;
	hipLaunchKernelGGL(k_astar_walk, dim3(1), dim3(64), 0, st, h->d, (int)index, h->walk, WALK_MAX);
	RK_HIP(hipGetLastError());
	int32_t len = 0;
	RK_HIP(hipMemcpyAsync(&len, h->walk, sizeof len, hipMemcpyDeviceToHost, st));
	RK_HIP(hipStreamSynchronize(st));
	if (len < 0) return fail(RK_ESTATE, "rk_astar_path: broken parent chain");
	const size_t n = std::min<size_t>(std::min<size_t>((size_t)len, max_len), (size_t)WALK_MAX);
	std::vector<int32_t> acts(n);
	if (n) {
		RK_HIP(hipMemcpyAsync(acts.data(), h->walk + 1, n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
		RK_HIP(hipStreamSynchronize(st));
	}
	for (size_t k = 0; k < n; k++) h_actions[k] = acts[k];
	return (long long)len;
}

long long rk_astar_lookup(rk_astar_t *h, const int8_t *h_state, void *stream)
{
	if (!h || !h_state) return fail(RK_EINVAL, "rk_astar_lookup: null argument");
	hipStream_t st = (hipStream_t)stream;
	int32_t out = 0;
	RK_HIP(hipMemcpyAsync(h->root_dev, h_state, STATE_BYTES, hipMemcpyHostToDevice, st));
	hipLaunchKernelGGL(k_astar_find, dim3(1), dim3(64), 0, st, h->root_dev, h->d.states, h->d.table, h->d.mask, h->walk);
	RK_HIP(hipGetLastError());
	RK_HIP(hipMemcpyAsync(&out, h->walk, sizeof out, hipMemcpyDeviceToHost, st));
	RK_HIP(hipStreamSynchronize(st));
	return out;
}

/* The whole open queue in pop order (inspection only: gathers every level's live records and sorts them on the host). */
long long rk_astar_export_open(rk_astar_t *h, double *h_costs, long long *h_indices, size_t max_len, void *stream)
{
	if (!h || !h->ready) return fail(RK_EINVAL, "rk_astar_export_open: null handle");
	hipStream_t st = (hipStream_t)stream;
	const QueueDev &q = h->d.q;
	int32_t meta[4 * QL];
	RK_HIP(hipMemcpyAsync(meta, q.meta, sizeof meta, hipMemcpyDeviceToHost, st));
	RK_HIP(hipStreamSynchronize(st));
	std::vector<Rec> all;
	for (int j = 0; j < q.levels; j++) {
		const int head = meta[Q_HEAD * QL + j] + meta[Q_TAKE * QL + j], len = meta[Q_LEN * QL + j];
		if (len <= head) continue;
		const size_t at = all.size();
		all.resize(at + (size_t)(len - head));
		RK_HIP(hipMemcpyAsync(all.data() + at, q.buf[j][meta[Q_CUR * QL + j]] + head, (size_t)(len - head) * sizeof(Rec), hipMemcpyDeviceToHost, st));
	}
	RK_HIP(hipStreamSynchronize(st));
	std::sort(all.begin(), all.end(), [](const Rec &a, const Rec &b) { return a.key < b.key || (a.key == b.key && a.idx < b.idx); });
	const size_t n = std::min(all.size(), max_len);
	for (size_t i = 0; i < n; i++) {
		if (h_costs) h_costs[i] = key_to_double(all[i].key);
		if (h_indices) h_indices[i] = (long long)all[i].idx;
	}
	return (long long)n;
}

// ---- hash-sharded mode ---------------------------------------------------------------------------------------------

int rk_shard_owner(const int8_t *h_state, int world)
{
	if (!h_state || world < 1) return fail(RK_EINVAL, "rk_shard_owner: bad argument");
	uint32_t s[5];
	memcpy(s, h_state, STATE_BYTES);
	return (int)owner_of(s, (uint32_t)world);
}

long long rk_astar_shard_block_bytes(const rk_astar_t *h)
{
	return h ? (long long)(32 + (size_t)h->d.K * 48) : 0;
}

long long rk_astar_shard_gather_len(const rk_astar_t *h) { return h ? 8 + h->d.N : 0; }

int rk_astar_shard_reset(rk_astar_t *h, const int8_t *h_start_state, double lambda, void *d_send, void *stream)
{
	if (!h || !h_start_state || !d_send) return fail(RK_EINVAL, "rk_astar_shard_reset: null argument");
	hipStream_t st = (hipStream_t)stream;
	const int mine = rk_shard_owner(h_start_state, h->d.world) == h->d.rank;
	if (int e = astar_reset_impl(h, h_start_state, lambda, mine, st)) return e;
	hipLaunchKernelGGL(k_shard_clear_send, dim3(1), dim3(64), 0, st, h->d, (uint8_t *)d_send, 3);
	// the first all-gather contribution: as k_end<true> would write it
	hipLaunchKernelGGL((k_end<true>), dim3(1), dim3(1024), 0, st, h->d, 0, 0, 0);
	launch_shard_wide_selection(h->d, st);
	RK_HIP(hipGetLastError());
	RK_HIP(hipStreamSynchronize(st));
	return RK_OK;
}

/* This rank's all-gather contribution (device pointer to 8 + N doubles; slot 5 = elapsed seconds, for the host of rank 0). */
void *rk_astar_shard_gather_ptr(rk_astar_t *h) { return h ? h->d.gather_in : nullptr; }

/* Let the engine write its all-gather contribution straight into caller memory (8 + N doubles, e.g. a torch tensor). */
int rk_astar_shard_bind(rk_astar_t *h, void *d_gather)
{
	if (!h || !d_gather || (reinterpret_cast<uintptr_t>(d_gather) & 7)) return fail(RK_EINVAL, "rk_astar_shard_bind: bad argument");
	h->d.gather_in = static_cast<double *>(d_gather);
	return RK_OK;
}

int rk_astar_shard_select(rk_astar_t *h, const void *d_gathered, double time_limit, double max_states, void *d_send, void *stream)
{
	if (!h || !h->ready || !d_gathered || !d_send) return fail(RK_EINVAL, "rk_astar_shard_select: bad argument");
	if (h->pending) return fail(RK_ESTATE, "rk_astar_shard_select: previous iteration not finished");
	hipStream_t st = (hipStream_t)stream;
	const AstarDev &d = h->d;
	hipLaunchKernelGGL(k_shard_decide, dim3(blocks((size_t)d.N, 256 / d.world)), dim3(256), 0, st, d, (const double *)d_gathered, time_limit, max_states, h->decision);
	hipLaunchKernelGGL(k_shard_expand, dim3(blocks((size_t)d.K, ASCAN)), dim3(ASCAN), 0, st, d, (uint8_t *)d_send);
	RK_HIP(hipGetLastError());
	return RK_OK;
}

/* h_out[8] = {stop reason, winner rank, winner index, total states, my pops, iterations, my states, 0}.  Synchronises. */
int rk_astar_shard_decision(rk_astar_t *h, long long *h_out, void *stream)
{
	if (!h || !h_out) return fail(RK_EINVAL, "rk_astar_shard_decision: null argument");
	hipStream_t st = (hipStream_t)stream;
	RK_HIP(hipMemcpyAsync(h_out, h->decision, D_COUNT * sizeof(long long), hipMemcpyDeviceToHost, st));
	RK_HIP(hipStreamSynchronize(st));
	return RK_OK;
}

int rk_astar_shard_insert(rk_astar_t *h, const void *d_recv, void *d_send, void *d_onehot, int out_dtype, void *stream)
{
	if (!h || !h->ready || !d_recv || !d_send) return fail(RK_EINVAL, "rk_astar_shard_insert: null argument");
	if (h->pending) return fail(RK_ESTATE, "rk_astar_shard_insert: previous iteration not finished");
	if (d_onehot && (reinterpret_cast<uintptr_t>(d_onehot) & 15)) return fail(RK_EINVAL, "rk_astar_shard_insert: one-hot buffer must be 16-byte aligned");
	hipStream_t st = (hipStream_t)stream;
	const AstarDev &d = h->d;
	const uint8_t *recv = (const uint8_t *)d_recv;
	const unsigned gK = blocks((size_t)d.K);
	for (int phase = 0; phase < 2; phase++)
		hipLaunchKernelGGL(k_shard_offers_in, dim3(gK), dim3(256), 0, st, d, recv, phase);
	hipLaunchKernelGGL(k_shard_lookup, dim3(gK), dim3(256), 0, st, d, recv);
	launch_append<true>(h, recv, d_onehot, out_dtype, st);
	RK_HIP(hipGetLastError());
	h->pending = true;
	return RK_OK;
}

/* Between insert and push: start an asynchronous copy of this iteration's new-state count (an int) into host memory and
 * return at once.  With pinned host memory the caller can enqueue the net on the first rows, wait for an event recorded
 * behind this call and then size the rest of the net batch exactly -- the count is never larger than 12 N. */
int rk_astar_shard_new_count(rk_astar_t *h, int *h_out, void *stream)
{
	if (!h || !h->pending || !h_out) return fail(RK_ESTATE, "rk_astar_shard_new_count: needs a pending insert and a host pointer");
	RK_HIP(hipMemcpyAsync(h_out, &h->d.ctr[C_NNEW], sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
	return RK_OK;
}

static int shard_push_impl(rk_astar_t *h, const float *d_values, int rows, const void *d_recv, void *d_send, void *stream)
{
	if (!h || !h->pending) return fail(RK_ESTATE, "rk_astar_shard_push: no pending insert");
	if (!d_values || !d_recv || !d_send) return fail(RK_EINVAL, "rk_astar_shard_push: null argument");
	if (rows < 0 || rows > h->d.K) return fail(RK_EINVAL, "rk_astar_shard_push_rows: %d rows outside 0..12 N = %d", rows, h->d.K);
	hipStream_t st = (hipStream_t)stream;
	// The caller promises values for the first `rows` new states and the engine stops the search if there are more (error 3), so this
	// iteration's sort / merge / insert need not be sized for K = 12 N records: `rows` of them at most.  At world 8, N = 700 that is one
	// pass over 1 344 records in 256-record chunks staged in LDS (the K <= 2048 form) instead of five 2 048-record chunks; at
	// N = 5 600 (weak) five chunks handed to the insert as they are instead of 33 chunks and five merge passes.  Every kernel of the
	// sequence receives the same geometry by value; should the promise break, they cover `Kpad` records and no more (clamped), and
	// k_end<true> reports it.
	AstarDev d = h->d;
	if (rows > 0 && rows < d.K) {
		if (rows <= SORT_CHUNK) d.chunk = SMALL_CHUNK;
		d.Kpad = std::min(d.Kpad, ((rows + d.chunk - 1) / d.chunk) * d.chunk);
	}
	const int from = launch_commit<true>(h, d_values, (const uint8_t *)d_recv, st, &d);
	hipLaunchKernelGGL(k_shard_offers, dim3(blocks((size_t)d.K, ASCAN)), dim3(ASCAN), 0, st, d, (const uint8_t *)d_recv, (uint8_t *)d_send);
	hipLaunchKernelGGL((k_end<true>), dim3(1), dim3(1024), 0, st, d, from, 1, rows == d.K ? 0 : rows);
	launch_shard_wide_selection(d, st);
	RK_HIP(hipGetLastError());
	h->pending = false;
	return RK_OK;
}

int rk_astar_shard_push(rk_astar_t *h, const float *d_values, const void *d_recv, void *d_send, void *stream)
{
	return shard_push_impl(h, d_values, 0, d_recv, d_send, stream);
}

int rk_astar_shard_push_rows(rk_astar_t *h, const float *d_values, int rows, const void *d_recv, void *d_send, void *stream)
{
	return shard_push_impl(h, d_values, rows, d_recv, d_send, stream);
}

/* After the search ended without a win: apply the offers that arrived with the last exchange (no records). */
int rk_astar_shard_flush(rk_astar_t *h, const void *d_recv, void *stream)
{
	if (!h || !h->ready || !d_recv) return fail(RK_EINVAL, "rk_astar_shard_flush: bad argument");
	hipStream_t st = (hipStream_t)stream;
	const unsigned gK = blocks((size_t)h->d.K);
	for (int phase = 0; phase < 2; phase++)
		hipLaunchKernelGGL(k_shard_offers_in, dim3(gK), dim3(256), 0, st, h->d, (const uint8_t *)d_recv, phase);
	RK_HIP(hipGetLastError());
	return RK_OK;
}

int rk_astar_shard_clear_send(rk_astar_t *h, void *d_send, int records, int offers, void *stream)
{
	if (!h || !d_send) return fail(RK_EINVAL, "rk_astar_shard_clear_send: bad argument");
	hipLaunchKernelGGL(k_shard_clear_send, dim3(1), dim3(64), 0, (hipStream_t)stream, h->d, (uint8_t *)d_send, (records ? 1 : 0) | (offers ? 2 : 0));
	RK_HIP(hipGetLastError());
	return RK_OK;
}

/* One hop of a cross-rank parent walk: (parent rank, parent index, action) of node `index` on this rank. */
int rk_astar_shard_parent(rk_astar_t *h, long long index, long long *h_out /* [3] */, void *stream)
{
	if (!h || !h_out) return fail(RK_EINVAL, "rk_astar_shard_parent: null argument");
	if (index < 1 || (size_t)index > h->cap) return fail(RK_EINVAL, "rk_astar_shard_parent: index %lld outside 1..%zu", index, h->cap);
	hipStream_t st = (hipStream_t)stream;
	int32_t p = 0;
	uint8_t a = 0, r = 0;
	RK_HIP(hipMemcpyAsync(&p, h->d.parents + index, sizeof p, hipMemcpyDeviceToHost, st));
	RK_HIP(hipMemcpyAsync(&a, h->d.pact + index, 1, hipMemcpyDeviceToHost, st));
	RK_HIP(hipMemcpyAsync(&r, h->d.prank + index, 1, hipMemcpyDeviceToHost, st));
	RK_HIP(hipStreamSynchronize(st));
	h_out[0] = r; h_out[1] = p; h_out[2] = a;
	return RK_OK;
}

/* Rows [first, first + count) of the parents' OWNER RANKS (the fourth node array of a sharded engine, beside rk_astar_export's
 * states / G / parents / actions): what a cross-rank walk of the parent links, or a comparison of whole shards, needs. */
int rk_astar_shard_export_ranks(rk_astar_t *h, size_t first, size_t count, long long *h_parent_ranks, void *stream)
{
	if (!h || !h_parent_ranks) return fail(RK_EINVAL, "rk_astar_shard_export_ranks: null argument");
	if (first + count > h->cap + 1) return fail(RK_EINVAL, "rk_astar_shard_export_ranks: rows %zu..%zu outside the pool", first, first + count);
	if (count == 0) return RK_OK;
	std::vector<uint8_t> r(count);
	RK_HIP(hipMemcpyAsync(r.data(), h->d.prank + first, count, hipMemcpyDeviceToHost, (hipStream_t)stream));
	RK_HIP(hipStreamSynchronize((hipStream_t)stream));
	for (size_t i = 0; i < count; i++) h_parent_ranks[i] = r[i];
	return RK_OK;
}

// ================================================================================================================
// Batched A*: S independent searches in lock-step (rk_astarb_*; AStarBatch in librubiks_amd/solving/agents.py).
// Every search is a complete single-search engine (its own pool, hash table, queue and counter block: an rk_astar);
// the batch keeps the S device descriptors in one array and launches the SAME kernels with a second grid dimension
// (blockIdx.y = search: the kb_* wrappers above), so one iteration of all searches is the six launches of one search,
// around one net forward on the (S * 12 N, 480) batch.  Round 1 had a separate set of batch kernels that re-merged
// every search's whole open queue each iteration; this form inherits the log-structured queue and everything else.
// ================================================================================================================
struct rk_astarb {
	int S = 0;
	std::vector<rk_astar *> eng;
	std::vector<uint8_t> start_solved;
	AstarDev *devs = nullptr;                     // device copy of every engine's descriptor
	int32_t *row_off = nullptr;                   // compact mode: S + 1 row offsets of the pending step
	bool ready = false, pending = false, compact = false;
};

static int astarb_upload(rk_astarb *b, hipStream_t st)
{
	std::vector<AstarDev> host((size_t)b->S);
	for (int s = 0; s < b->S; s++) host[(size_t)s] = b->eng[(size_t)s]->d;
	RK_HIP(hipMemcpyAsync(b->devs, host.data(), host.size() * sizeof(AstarDev), hipMemcpyHostToDevice, st));
	RK_HIP(hipStreamSynchronize(st));
	return RK_OK;
}

int rk_astarb_create(rk_astarb_t **out, int n_searches, size_t capacity_per_search, int max_expansions)
{
	if (!out) return fail(RK_EINVAL, "rk_astarb_create: null out pointer");
	if (n_searches < 1 || n_searches > 65535) return fail(RK_EINVAL, "rk_astarb_create: n_searches %d out of range", n_searches);
	if (max_expansions < 1 || max_expansions > (1 << 20)) return fail(RK_EINVAL, "rk_astarb_create: max_expansions %d out of range", max_expansions);
	if (capacity_per_search < (size_t)12 * max_expansions + 2 || capacity_per_search > 0x3FFFFFF0ull)
		return fail(RK_EINVAL, "rk_astarb_create: capacity %zu out of range (needs at least 12 * expansions + 2)", capacity_per_search);
	rk_astarb *b = new rk_astarb();
	b->S = n_searches;
	b->start_solved.assign((size_t)n_searches, 0);
	for (int s = 0; s < n_searches; s++) {
		rk_astar *e = nullptr;
		const int rc = astar_create_impl(&e, capacity_per_search, max_expansions, 0, 1);
		if (rc) { rk_astarb_destroy(b); return rc; }
		b->eng.push_back(e);
	}
	if (hipMalloc((void **)&b->devs, (size_t)n_searches * sizeof(AstarDev)) != hipSuccess ||
	    hipMalloc((void **)&b->row_off, ((size_t)n_searches + 1) * sizeof(int32_t)) != hipSuccess) {
		rk_astarb_destroy(b);
		return fail(RK_EHIP, "rk_astarb_create: hipMalloc failed");
	}
	*out = b;
	return RK_OK;
}

int rk_astarb_destroy(rk_astarb_t *b)
{
	if (!b) return RK_OK;
	for (rk_astar *e : b->eng) rk_astar_destroy(e);
	(void)hipFree(b->devs);
	(void)hipFree(b->row_off);
	delete b;
	return RK_OK;
}

int rk_astarb_reset(rk_astarb_t *b, const int8_t *h_start_states, const long long *h_max_states, double lambda, void *stream)
{
	if (!b || !h_start_states) return fail(RK_EINVAL, "rk_astarb_reset: null argument");
	hipStream_t st = (hipStream_t)stream;
	uint32_t solved5[5];
	{
		int8_t tmp[STATE_BYTES];
		rk_solved(RK_REPR_2024, tmp);
		memcpy(solved5, tmp, STATE_BYTES);
	}
	for (int s = 0; s < b->S; s++) {
		rk_astar *e = b->eng[(size_t)s];
		const int8_t *start = h_start_states + (size_t)s * STATE_BYTES;
		int rc = astar_reset_impl(e, start, lambda, 1, st);
		if (rc) return rc;
		b->start_solved[(size_t)s] = memcmp(start, solved5, STATE_BYTES) == 0 ? 1 : 0;      // agents.py:225: nothing to search
		long long budget = h_max_states ? h_max_states[s] : (long long)e->cap;
		if (b->start_solved[(size_t)s]) budget = 0;
		rc = rk_astar_set_budget(e, budget, stream);
		if (rc) return rc;
	}
	const int rc = astarb_upload(b, st);
	if (rc) return rc;
	b->ready = true;
	b->pending = false;
	return RK_OK;
}

int rk_astarb_set_values_dtype(rk_astarb_t *b, int dtype, void *stream)
{
	if (!b) return fail(RK_EINVAL, "rk_astarb_set_values_dtype: null handle");
	if (dtype != RK_OH_F32 && dtype != RK_OH_BF16) return fail(RK_EINVAL, "rk_astarb_set_values_dtype: float32 or bfloat16");
	const int flag = dtype == RK_OH_BF16 ? 1 : 0;
	if (b->eng[0]->d.values_bf16 == flag) return RK_OK;
	for (rk_astar *e : b->eng) e->d.values_bf16 = flag;
	return astarb_upload(b, (hipStream_t)stream);
}

static int astarb_step_expand_impl(rk_astarb_t *b, void *d_onehot, int out_dtype, int *h_total, void *stream);

int rk_astarb_step_expand(rk_astarb_t *b, void *d_onehot, int out_dtype, void *stream)
{
	return astarb_step_expand_impl(b, d_onehot, out_dtype, nullptr, stream);
}

/* The same step with the net's rows COMPACTED across the searches: only the new states of every search, one search after
 * the other (each search's first row rounded up to a multiple of 4), and the total row count copied asynchronously into
 * (pinned) host memory.  The caller waits for that count (an event behind this call), runs the net on that many rows and
 * commits with values laid out the same way.  Not capturable in a hipGraph (the batch size varies). */
int rk_astarb_step_expand_compact(rk_astarb_t *b, void *d_rows, int out_dtype, int *h_total, void *stream)
{
	if (!h_total) return fail(RK_EINVAL, "rk_astarb_step_expand_compact: null host pointer for the row count");
	return astarb_step_expand_impl(b, d_rows, out_dtype, h_total, stream);
}

static int astarb_step_expand_impl(rk_astarb_t *b, void *d_onehot, int out_dtype, int *h_total, void *stream)
{
	if (!b || !b->ready) return fail(RK_ESTATE, "rk_astarb_step_expand: reset the engine first");
	if (b->pending) return fail(RK_ESTATE, "rk_astarb_step_expand: previous step not committed");
	if (!d_onehot || (reinterpret_cast<uintptr_t>(d_onehot) & 15)) return fail(RK_EINVAL, "rk_astarb_step_expand: one-hot buffer must be 16-byte aligned");
	if (out_dtype < RK_OH_F32 || out_dtype > RK_OH_STATES) return fail(RK_EINVAL, "rk_astarb_step_expand: unknown dtype %d", out_dtype);
	hipStream_t st = (hipStream_t)stream;
	const AstarDev &d = b->eng[0]->d;                 // shapes are the same for every search
	const unsigned S = (unsigned)b->S;
	hipLaunchKernelGGL(kb_expand_lookup, dim3(blocks((size_t)d.K), S), dim3(256), 0, st, b->devs);
	hipLaunchKernelGGL((kb_append<false>), dim3(blocks((size_t)d.K, ASCAN), S), dim3(ASCAN), 0, st, b->devs, (const uint8_t *)nullptr);
	const int32_t *row_off = nullptr;
	if (h_total != nullptr) {
		hipLaunchKernelGGL(kb_row_offsets, dim3(1), dim3(1024), 0, st, b->devs, b->S, b->row_off);
		RK_HIP(hipMemcpyAsync(h_total, b->row_off + b->S, sizeof(int32_t), hipMemcpyDeviceToHost, st));
		row_off = b->row_off;
	}
	const size_t chunks = (size_t)d.K * (out_dtype == RK_OH_F32 ? 120 : out_dtype == RK_OH_STATES ? 2 : 60);
	const unsigned grid = std::min<unsigned>(blocks(chunks), 8192u);
	if (out_dtype == RK_OH_F32)
		hipLaunchKernelGGL((kb_new_rows<4, false>), dim3(grid, S), dim3(256), 0, st, b->devs, (u32x4 *)d_onehot, 0x3F800000u, (const uint8_t *)nullptr, row_off);
	else if (out_dtype == RK_OH_STATES)
		hipLaunchKernelGGL((kb_new_rows<0, false>), dim3(grid, S), dim3(256), 0, st, b->devs, (u32x4 *)d_onehot, 0u, (const uint8_t *)nullptr, row_off);
	else
		hipLaunchKernelGGL((kb_new_rows<2, false>), dim3(grid, S), dim3(256), 0, st, b->devs, (u32x4 *)d_onehot, out_dtype == RK_OH_F16 ? 0x3C00u : 0x3F80u, (const uint8_t *)nullptr, row_off);
	RK_HIP(hipGetLastError());
	b->pending = true;
	b->compact = h_total != nullptr;
	return RK_OK;
}

int rk_astarb_step_commit(rk_astarb_t *b, const float *d_values, void *stream)
{
	if (!b || !b->pending) return fail(RK_ESTATE, "rk_astarb_step_commit: no pending step");
	if (!d_values) return fail(RK_EINVAL, "rk_astarb_step_commit: null values");
	hipStream_t st = (hipStream_t)stream;
	const AstarDev &d = b->eng[0]->d;
	const unsigned S = (unsigned)b->S;
	int from = 0;
	const int32_t *row_off = b->compact ? b->row_off : nullptr;
	if (d.chunk == SMALL_CHUNK) {
		hipLaunchKernelGGL((kb_records_sort<SMALL_CHUNK>), dim3(d.Kpad / SMALL_CHUNK, S), dim3(SMALL_CHUNK / 2), 0, st, b->devs, d_values, row_off);
	} else {
		hipLaunchKernelGGL((kb_records_sort<SORT_CHUNK>), dim3(d.Kpad / SORT_CHUNK, S), dim3(SORT_CHUNK / 2), 0, st, b->devs, d_values, row_off);
		for (int L = SORT_CHUNK; L < d.Kpad && new_chunk_of(d.chunk, d.Kpad) == 0; L <<= 1) {
			hipLaunchKernelGGL(kb_merge_pass, dim3(blocks(d.Kpad), S), dim3(256), 0, st, b->devs, L, from);
			from ^= 1;
		}
	}
	const unsigned grid = std::min<unsigned>(1024u, std::max<unsigned>(blocks((size_t)d.Kpad * 4), 8u));
	hipLaunchKernelGGL((kb_queue_insert<false>), dim3(grid, S), dim3(256), 0, st, b->devs, from);
	hipLaunchKernelGGL((kb_end<false>), dim3(1, S), dim3(1024), 0, st, b->devs, from, 1);
	if (pop_is_wide_host(d)) hipLaunchKernelGGL(kb_pop_wide, dim3(blocks((size_t)d.q.levels * d.N), S), dim3(256), 0, st, b->devs);
	RK_HIP(hipGetLastError());
	b->pending = false;
	return RK_OK;
}

int rk_astarb_status(rk_astarb_t *b, long long *h_status, void *stream)
{
	if (!b || !b->ready || !h_status) return fail(RK_EINVAL, "rk_astarb_status: bad argument");
	hipStream_t st = (hipStream_t)stream;
	std::vector<int32_t> c((size_t)b->S * C_COUNT);
	for (int s = 0; s < b->S; s++)
		RK_HIP(hipMemcpyAsync(c.data() + (size_t)s * C_COUNT, b->eng[(size_t)s]->d.ctr, C_COUNT * sizeof(int32_t), hipMemcpyDeviceToHost, st));
	RK_HIP(hipStreamSynchronize(st));
	for (int s = 0; s < b->S; s++) {
		const int32_t *k = c.data() + (size_t)s * C_COUNT;
		long long *o = h_status + (size_t)s * 7;
		const bool ss = b->start_solved[(size_t)s] != 0;
		o[0] = ss ? 1 : k[C_DONE]; o[1] = ss ? 2 : k[C_WON]; o[2] = k[C_NSTATES]; o[3] = k[C_ITERS]; o[4] = k[C_OPEN];
		o[5] = k[C_SOLVED]; o[6] = k[C_ERROR];
	}
	return RK_OK;
}

int rk_astarb_export(rk_astarb_t *b, int search, size_t first, size_t count, int8_t *h_states, double *h_G,
                     long long *h_parents, long long *h_parent_actions, void *stream)
{
	if (!b || !b->ready) return fail(RK_ESTATE, "rk_astarb_export: reset the engine first");
	if (search < 0 || search >= b->S) return fail(RK_EINVAL, "rk_astarb_export: search %d out of range", search);
	return rk_astar_export(b->eng[(size_t)search], first, count, h_states, h_G, h_parents, h_parent_actions, stream);
}

long long rk_astarb_path(rk_astarb_t *b, int search, long long index, long long *h_actions, size_t max_len, void *stream)
{
	if (!b || !b->ready) return fail(RK_ESTATE, "rk_astarb_path: reset the engine first");
	if (search < 0 || search >= b->S) return fail(RK_EINVAL, "rk_astarb_path: search %d out of range", search);
	return rk_astar_path(b->eng[(size_t)search], index, h_actions, max_len, stream);
}

}  // extern "C"
