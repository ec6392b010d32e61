// Device-resident batch weighted A* (reference: librubiks/solving/agents.py:171-413).
//
// What lives in HBM (capacity C states, N = max expansions per iteration, K = 12 N children per iteration):
//   states  int8 (C+1, 20)   node pool, index 0 unused, root = 1              (agents.py:202, :390)
//   G       int32 (C+1)      path cost (whole numbers; exported as float64)    (agents.py:203, :393)
//   parents int32, pact uint8                                                  (agents.py:204-205)
//   table   uint32 (T)       open-addressing hash table state -> index, T = pow2 >= 2C   (the `indices` dict)
//   mark    uint32 (C+1)     per-node scratch: batch position of a seen state's first occurrence
//   open    Rec[2][C+1]      the open queue as an array SORTED by (cost, index); ping-pong buffers
//
// The reference pops with heapq from a heap of (cost, idx) tuples and never re-pushes a node, so "the N smallest
// (cost, idx) pairs in ascending order" is exactly what it expands.  A sorted array makes the pop free (take the
// head); new nodes always carry larger indices than old ones, so pushing is: sort the <= K new records by
// (cost, idx), then one rank-merge with the remaining queue (every element finds its output slot by a binary
// search in the other run; all (cost, idx) keys are distinct, so there are no tie cases).
//
// Order-dependent semantics that are reproduced exactly:
//   * children are generated parent-major / action-minor in pop order                      (agents.py:277-282)
//   * np.unique(..., return_index=True) keeps the FIRST occurrence of a state in batch order; only first
//     occurrences are appended (unseen) or relaxed (seen)                                  (agents.py:291-295)
//   * new indices are handed out in batch order                                            (agents.py:300)
//   * relaxation is two vectorised passes, each reading all of G before writing, the second seeing the first's
//     writes; duplicate targets in the second pass resolve to the LAST one in batch order  (agents.py:353-367)
#include <hip/hip_runtime.h>
#include <climits>
#include <cstring>
#include <vector>

#include "../../include/rubiks_hip.h"
#include "rk_device.h"
#include "rk_error.h"
#include "rk_kernels.h"
#include "rk_search_dev.h"

namespace rk {

// ---------------------------------------------------------------------------------------------------------------
__global__ void k_astar_root(uint32_t *states, int32_t *G, int32_t *parents, uint8_t *pact, uint32_t *table, uint32_t mask,
                             Rec *open, const uint32_t *root)
{
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	uint32_t s[5];
	load5(root, s);
	#pragma unroll
	for (int j = 0; j < 5; j++) states[5 + j] = s[j];
	G[1] = 0; parents[1] = 0; pact[1] = 0;
	table[hash_state(s) & mask] = 1u;
	open[0] = Rec{sortable_key(0.0), 1ull};       // heappush(open_queue, (0, 1))   agents.py:234
}

// pop: the head of the sorted queue; gather the parents' states                                 agents.py:238-239
__global__ void k_astar_pop(const Rec *open, int n_pop, const uint32_t *states, int32_t *exp_idx, uint32_t *par_states)
{
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n_pop * 5) return;
	const int i = t / 5, j = t - 5 * i;
	const uint32_t idx = (uint32_t)open[i].idx;
	if (j == 0) exp_idx[i] = (int32_t)idx;
	par_states[t] = states[(size_t)idx * 5 + j];
}

// membership test + in-batch first-occurrence election through the hash table                    agents.py:286-295
// `stride` = dwords between consecutive child states: 5 for a plain (K,20) array, 8 for 32-byte exchange records
__global__ void k_astar_lookup(const uint32_t *children, int stride, int K, const uint32_t *states, uint32_t *table, uint32_t mask,
                               uint32_t *mark, int32_t *seen, uint32_t *child_slot)
{
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= K) return;
	uint32_t s[5];
	load5(children + (size_t)c * stride, s);
	uint32_t slot = hash_state(s) & mask;
	for (;;) {
		uint32_t e = __atomic_load_n(&table[slot], __ATOMIC_RELAXED);
		if (e == 0u) {
			e = atomicCAS(&table[slot], 0u, TENT | (uint32_t)c);
			if (e == 0u) { seen[c] = 0; child_slot[c] = slot; return; }
		}
		if (e & TENT) {
			if (equal5(s, children + (size_t)(e & ~TENT) * stride)) {
				atomicMin(&table[slot], TENT | (uint32_t)c);          // all claimants hold the same state: smallest position wins
				seen[c] = 0; child_slot[c] = slot;
				return;
			}
		} else if (equal5(s, states + (size_t)e * 5)) {
			seen[c] = (int32_t)e;
			atomicMin(&mark[e], (uint32_t)c);
			return;
		}
		slot = (slot + 1) & mask;
	}
}

// ---- order-preserving compaction across many workgroups ---------------------------------------------------------
// Three small launches: (1) per-1024-element workgroup: predicate, rank inside the workgroup, workgroup total;
// (2) one workgroup scans the totals; (3) the consumer adds the workgroup's offset to the local rank.
// first_unseen / first_seen flags (agents.py:291-295) and the rank of every first_unseen child inside its workgroup
__global__ __launch_bounds__(SCAN_BLOCK)
void k_astar_flags(int K, const uint32_t *table, const uint32_t *mark, const int32_t *seen, const uint32_t *child_slot,
                   uint8_t *flags, int32_t *rank, int32_t *block_sums)
{
	__shared__ int s_wave[16];
	const int c = blockIdx.x * SCAN_BLOCK + threadIdx.x;
	int fu = 0, fs = 0;
	if (c < K) {
		const int32_t sidx = seen[c];
		if (sidx == 0) fu = table[child_slot[c]] == (TENT | (uint32_t)c);
		else fs = mark[sidx] == (uint32_t)c;
		flags[c] = (uint8_t)(fu | (fs << 1));
	}
	int total;
	const int r = block_rank(fu != 0, s_wave, &total);
	if (c < K) rank[c] = r;
	if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// exclusive scan of `n` workgroup totals in place (one workgroup); the grand total goes to *out_total
__global__ __launch_bounds__(SCAN_BLOCK)
void k_scan_blocks(int32_t *sums, int n, long long *out_total)
{
	__shared__ int s_wave[16];
	__shared__ int s_base;
	if (threadIdx.x == 0) s_base = 0;
	__syncthreads();
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	for (int i0 = 0; i0 < n; i0 += SCAN_BLOCK) {
		const int i = i0 + threadIdx.x;
		const int v = i < n ? sums[i] : 0;
		int incl = v;                                  // inclusive scan inside the wave
		#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const int o = __shfl_up(incl, d, 64);
			if (lane >= d) incl += o;
		}
		if (lane == 63) s_wave[wv] = incl;
		__syncthreads();
		int before = s_base;
		for (int w = 0; w < wv; w++) before += s_wave[w];
		if (i < n) sums[i] = before + incl - v;
		__syncthreads();
		if (threadIdx.x == 0) {
			int tot = 0;
			for (int w = 0; w < 16; w++) tot += s_wave[w];
			s_base += tot;
		}
		__syncthreads();
	}
	if (threadIdx.x == 0 && out_total != nullptr) *out_total = s_base;
}

// append the new states (agents.py:299-313), finalise their hash slots, goal test of the new states
// (agents.py:321-323) and the read half of relaxation case 1 (agents.py:354).
// SHARDED = false: children is a (K,20) array, the parent of child c is exp_idx[c/12], its action c%12.
// SHARDED = true : children are 32-byte records {state[5], parent_idx, g | action<<16 | parent_rank<<24, pad}
//                  received from the ranks that expanded them; the parent lives on rank parent_rank.
template <bool SHARDED>
__global__ void k_astar_append(const uint32_t *children, const uint8_t *solved, int K, const uint8_t *flags, const int32_t *rank,
                               const int32_t *block_off, const int32_t *seen, const uint32_t *child_slot, const int32_t *exp_idx, uint32_t n_before,
                               uint32_t *states, int32_t *G, int32_t *parents, uint8_t *pact, uint8_t *prank, uint32_t *table,
                               uint8_t *newway, int32_t *val1, long long *counters)
{
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= K) return;
	constexpr int STRIDE = SHARDED ? 8 : 5;
	const uint8_t f = flags[c];
	const uint32_t *cs = children + (size_t)c * STRIDE;
	int32_t p, g;
	uint8_t act, pr = 0;
	if (SHARDED) {
		p = (int32_t)cs[5];
		g = (int32_t)(cs[6] & 0xFFFFu);
		act = (uint8_t)((cs[6] >> 16) & 0xFFu);
		pr = (uint8_t)(cs[6] >> 24);
	} else {
		p = exp_idx[c / 12];
		g = G[p] + 1;
		act = (uint8_t)(c % 12);
	}
	if (f & 1) {
		const uint32_t idx = n_before + 1u + (uint32_t)(rank[c] + block_off[c / SCAN_BLOCK]);
		uint32_t s[5];
		load5(cs, s);
		#pragma unroll
		for (int j = 0; j < 5; j++) states[(size_t)idx * 5 + j] = s[j];
		G[idx] = g;
		parents[idx] = p;
		pact[idx] = act;
		prank[idx] = pr;
		table[child_slot[c]] = idx;
		const bool is_goal = SHARDED ? is_solved5(s) : (solved[c] != 0);
		if (is_goal) { counters[CTR_WON] = 1; counters[CTR_SOLVED_IDX] = idx; }
	}
	uint8_t nw = 0;
	if (f & 2) {
		nw = g < G[seen[c]];
		val1[c] = g;
	}
	newway[c] = nw;
}

// cost = lambda * G + (-value), float64, no fused multiply-add                                      agents.py:380-383
__global__ void k_astar_records(const float *values, int n_new, uint32_t n_before, const int32_t *G, double lambda, Rec *rec)
{
	const int j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n_new) return;
	const uint32_t idx = n_before + 1u + (uint32_t)j;
	const double h = (double)(-values[j]);
	const double lg = lambda * (double)G[idx];
	rec[j] = Rec{sortable_key(lg + h), (uint64_t)idx};
}

// bitonic sort of chunks of 1024 records in LDS
__global__ __launch_bounds__(512)
void k_sort_chunks(Rec *rec, int n)
{
	__shared__ Rec s[1024];
	const int base = blockIdx.x * 1024, tid = threadIdx.x;
	for (int i = tid; i < 1024; i += 512) s[i] = (base + i < n) ? rec[base + i] : Rec{~0ull, ~0ull};
	__syncthreads();
	for (int k = 2; k <= 1024; k <<= 1)
		for (int j = k >> 1; j > 0; j >>= 1) {
			const int i = 2 * tid - (tid & (j - 1));              // lower index of this thread's pair
			const int l = i + j;
			const bool up = (i & k) == 0;
			const Rec a = s[i], b = s[l];
			if (rec_less(b, a) == up) { s[i] = b; s[l] = a; }
			__syncthreads();
		}
	for (int i = tid; i < 1024; i += 512)
		if (base + i < n) rec[base + i] = s[i];
}

// merge neighbouring sorted runs of length L (keys are distinct)
__global__ void k_merge_pass(const Rec *src, Rec *dst, int n, int L)
{
	const int e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n) return;
	const int r = e / L, i = e - r * L;
	const int base = (r & ~1) * L;
	const int pstart = (r ^ 1) * L;
	int plen = n - pstart;
	plen = plen < 0 ? 0 : (plen > L ? L : plen);
	const Rec x = src[e];
	dst[base + i + lower_bound_rec(src + pstart, plen, x)] = x;
}

__global__ void k_merge_two(const Rec *a, int na, const Rec *b, int nb, Rec *out)
{
	const int e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= na + nb) return;
	if (e < na) {
		const Rec x = a[e];
		out[e + lower_bound_rec(b, nb, x)] = x;
	} else {
		const Rec x = b[e - na];
		out[(e - na) + lower_bound_rec(a, na, x)] = x;
	}
}

// relaxation, case 1 write half (agents.py:357-359)
template <bool SHARDED>
__global__ void k_relax_1b(int K, const uint8_t *newway, const int32_t *val1, const int32_t *seen, const int32_t *exp_idx,
                           const uint32_t *recs, int32_t *G, int32_t *parents, uint8_t *pact, uint8_t *prank, const long long *counters)
{
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= K || !newway[c]) return;
	if (SHARDED && counters[CTR_WON]) return;       // the reference returns before relaxing once it has won (agents.py:321-323)
	const int32_t s = seen[c];
	G[s] = val1[c];
	if (SHARDED) {
		const uint32_t *r = recs + (size_t)c * 8;
		pact[s] = (uint8_t)((r[6] >> 16) & 0xFFu);
		parents[s] = (int32_t)r[5];
		prank[s] = (uint8_t)(r[6] >> 24);
	} else {
		pact[s] = (uint8_t)(c % 12);
		parents[s] = exp_idx[c / 12];
	}
}

// case 2 read half (agents.py:362); also clears the marks this batch set
__global__ void k_relax_2a(int K, const uint8_t *flags, const int32_t *seen, const int32_t *exp_idx, const int32_t *G,
                           uint32_t *mark, uint8_t *shortcut, int32_t *val2)
{
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= K) return;
	uint8_t sc = 0;
	if (flags[c] & 2) {
		const int32_t s = seen[c];
		const int32_t g = G[s] + 1;
		sc = g < G[exp_idx[c / 12]];
		val2[c] = g;
		mark[s] = NO_MARK;
	}
	shortcut[c] = sc;
}

// case 2 write half (agents.py:365-367): one thread per expanded parent walks its 12 children in order, so the
// last shortcut child in batch order wins, as NumPy's fancy assignment with repeated indices does
__global__ void k_relax_2b(int n_pop, const uint8_t *shortcut, const int32_t *val2, const int32_t *seen, const int32_t *exp_idx,
                           int32_t *G, int32_t *parents, uint8_t *pact)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_pop) return;
	const int32_t p = exp_idx[i];
	for (int a = 0; a < 12; a++) {
		const int c = 12 * i + a;
		if (shortcut[c]) {
			G[p] = val2[c];
			pact[p] = (uint8_t)(a ^ 1);           // rev_action                                 cube.py:197-200
			parents[p] = seen[c];
		}
	}
}


// ---------------------------------------------------------------------------------------------------------------
// Hash-sharded search (one engine per GPU, owner(state) = owner_of(state, world)).  Per iteration a rank expands its
// share of the globally best nodes, buckets the 12 n children by owner (stable: batch order inside a bucket), the
// host exchanges the buckets (all-to-all), and every owner inserts what it received with exactly the single-GPU
// semantics above.  Relaxation case 2 (a seen child offers its parent a shortcut) becomes a second, small exchange
// of 16-byte records back to the parent's owner.
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_shard_records(const uint32_t *children, int K, const int32_t *exp_idx, const int32_t *G, uint32_t my_rank,
                                uint32_t world, uint32_t *recs, uint8_t *owner)
{
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= K) return;
	uint32_t s[5];
	load5(children + (size_t)c * 5, s);
	const int32_t p = exp_idx[c / 12];
	uint32_t *r = recs + (size_t)c * 8;
	#pragma unroll
	for (int j = 0; j < 5; j++) r[j] = s[j];
	r[5] = (uint32_t)p;
	r[6] = ((uint32_t)(G[p] + 1) & 0xFFFFu) | ((uint32_t)(c % 12) << 16) | (my_rank << 24);
	r[7] = (uint32_t)c;
	owner[c] = (uint8_t)owner_of(s, world);
}

// Stable partition of the K records by owner into `send` (batch order inside every bucket), many workgroups:
// (1) per-workgroup histogram over owners, laid out owner-major [w][block] so that ONE exclusive scan of the whole
//     array yields, for every (owner, workgroup), the first output slot of that workgroup's records for that owner;
// (2) k_scan_blocks; (3) scatter: rank among the same-owner records of the workgroup + that offset.
__global__ __launch_bounds__(SCAN_BLOCK)
void k_shard_hist(int K, uint32_t world, const uint8_t *owner, int32_t *hist /* [world][n_blocks] */, int n_blocks)
{
	__shared__ int s_cnt[256];
	if (threadIdx.x < 256) s_cnt[threadIdx.x] = 0;
	__syncthreads();
	const int c = blockIdx.x * SCAN_BLOCK + threadIdx.x;
	if (c < K) atomicAdd(&s_cnt[owner[c]], 1);
	__syncthreads();
	if (threadIdx.x < world) hist[(size_t)threadIdx.x * n_blocks + blockIdx.x] = s_cnt[threadIdx.x];
}

__global__ __launch_bounds__(SCAN_BLOCK)
void k_shard_scatter(int K, uint32_t world, const uint32_t *recs, const uint8_t *owner, const int32_t *offs /* scanned hist */,
                     int n_blocks, u32x4 *send, long long *counts)
{
	__shared__ int s_wave[16];
	const int c = blockIdx.x * SCAN_BLOCK + threadIdx.x;
	const uint32_t mine = c < K ? owner[c] : 0xFFFFFFFFu;
	int dest = -1;
	for (uint32_t w = 0; w < world; w++) {             // world <= 8 on a node; every round is one ballot per wave
		int total;
		const int r = block_rank(mine == w, s_wave, &total);
		if (mine == w) dest = offs[(size_t)w * n_blocks + blockIdx.x] + r;
	}
	if (dest >= 0) {
		const u32x4 *src = reinterpret_cast<const u32x4 *>(recs);
		send[2 * (size_t)dest] = src[2 * (size_t)c];
		send[2 * (size_t)dest + 1] = src[2 * (size_t)c + 1];
	}
	// per-owner totals: first slot of the next owner minus first slot of this one
	if (blockIdx.x == 0 && threadIdx.x < world) {
		const uint32_t w = threadIdx.x;
		const long long start = offs[(size_t)w * n_blocks];
		const long long end = w + 1 < world ? offs[(size_t)(w + 1) * n_blocks] : (long long)K;
		counts[w] = end - start;
	}
}

// receiver side of relaxation case 2: a first-seen child whose own G is at least two below its would-be parent's
// offers the parent a shortcut.  Offers keep receive order (= grouped by the rank that sent the child).
// Shortcut record (16 B): {parent_idx, new G for the parent, index of the child on this rank, this rank | rev(action) << 8}
__device__ __forceinline__ bool shortcut_offer(int c, int K, const uint8_t *flags, const int32_t *seen, const uint32_t *recs,
                                               const int32_t *G, uint32_t my_rank, u32x4 *rec, uint32_t *dst)
{
	if (c >= K || !(flags[c] & 2)) return false;
	const uint32_t *r = recs + (size_t)c * 8;
	const int32_t s = seen[c];
	const int32_t g_parent = (int32_t)(r[6] & 0xFFFFu) - 1;
	const int32_t g_new = G[s] + 1;
	*dst = r[6] >> 24;
	*rec = u32x4{r[5], (uint32_t)g_new, (uint32_t)s, my_rank | ((((r[6] >> 16) & 0xFFu) ^ 1u) << 8)};
	return g_new < g_parent;
}

__global__ __launch_bounds__(SCAN_BLOCK)
void k_shard_offers_count(int K, const uint8_t *flags, const int32_t *seen, const uint32_t *recs, const int32_t *G, uint32_t my_rank,
                          uint32_t *mark, int32_t *rank, int32_t *block_sums, long long *counts)
{
	__shared__ int s_wave[16];
	const int c = blockIdx.x * SCAN_BLOCK + threadIdx.x;
	u32x4 rec;
	uint32_t dst = 0;
	const bool cand = shortcut_offer(c, K, flags, seen, recs, G, my_rank, &rec, &dst);
	if (c < K && (flags[c] & 2)) mark[seen[c]] = NO_MARK;       // the marks this batch set are no longer needed
	int total;
	const int r = block_rank(cand, s_wave, &total);
	if (c < K) rank[c] = cand ? r : -1;
	if (cand) atomicAdd(reinterpret_cast<unsigned long long *>(&counts[dst]), 1ull);
	if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(SCAN_BLOCK)
void k_shard_offers_write(int K, const uint8_t *flags, const int32_t *seen, const uint32_t *recs, const int32_t *G, uint32_t my_rank,
                          const int32_t *rank, const int32_t *block_off, u32x4 *out)
{
	const int c = blockIdx.x * SCAN_BLOCK + threadIdx.x;
	if (c >= K || rank[c] < 0) return;
	u32x4 rec;
	uint32_t dst;
	shortcut_offer(c, K, flags, seen, recs, G, my_rank, &rec, &dst);
	out[block_off[blockIdx.x] + rank[c]] = rec;
}

// parent side of case 2 (agents.py:362-367): evaluate every offer against G as it stands, then let the LAST hit per
// parent (in arrival order) win -- what NumPy's fancy assignment with repeated indices does.
__global__ void k_shard_shortcut_eval(const u32x4 *recs, int n, const int32_t *G, uint32_t *mark, uint8_t *hit)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const u32x4 r = recs[i];
	const bool h = (int32_t)r.y < G[r.x];
	hit[i] = h;
	if (h) atomicMin(&mark[r.x], ~(uint32_t)i);
}

__global__ void k_shard_shortcut_apply(const u32x4 *recs, int n, const uint8_t *hit, int32_t *G, int32_t *parents, uint8_t *pact,
                                       uint8_t *prank, uint32_t *mark)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n || !hit[i]) return;
	const u32x4 r = recs[i];
	if (mark[r.x] != ~(uint32_t)i) return;
	G[r.x] = (int32_t)r.y;
	parents[r.x] = (int32_t)r.z;
	prank[r.x] = (uint8_t)(r.w & 0xFFu);
	pact[r.x] = (uint8_t)((r.w >> 8) & 0xFFu);
}

__global__ void k_shard_shortcut_clear(const u32x4 *recs, int n, uint32_t *mark)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) mark[recs[i].x] = NO_MARK;
}

__global__ void k_astar_find(const uint32_t *query, const uint32_t *states, const uint32_t *table, uint32_t mask, long long *out)
{
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	uint32_t s[5];
	load5(query, s);
	uint32_t slot = hash_state(s) & mask;
	for (;;) {
		const uint32_t e = table[slot];
		if (e == 0u) { *out = 0; return; }
		if (!(e & TENT) && equal5(s, states + (size_t)e * 5)) { *out = e; return; }
		slot = (slot + 1) & mask;
	}
}

}  // namespace rk

using namespace rk;

struct rk_astar {
	size_t cap = 0;
	int max_exp = 0;
	uint32_t mask = 0;
	double lambda = 0.0;
	// node pool
	uint32_t *states = nullptr; int32_t *G = nullptr, *parents = nullptr; uint8_t *pact = nullptr, *prank = nullptr;
	uint32_t *table = nullptr, *mark = nullptr;
	Rec *open[2] = {nullptr, nullptr};
	int cur = 0;
	// per-iteration scratch, K = 12 * max_exp
	int32_t *exp_idx = nullptr; uint32_t *par_states = nullptr, *children = nullptr; uint8_t *solved = nullptr;
	int32_t *seen = nullptr; uint32_t *child_slot = nullptr; uint8_t *flags = nullptr; int32_t *rank_ = nullptr;
	uint8_t *newway = nullptr, *shortcut = nullptr; int32_t *val1 = nullptr, *val2 = nullptr;
	Rec *newrec[2] = {nullptr, nullptr};
	long long *counters = nullptr;
	uint32_t *root_dev = nullptr;
	// hash-sharded mode
	int rank = 0, world = 1;
	size_t k_in = 0;              // capacity (records) of the per-iteration scratch: 12 * max_exp * world
	uint32_t *recs = nullptr; uint8_t *owner = nullptr, *hit = nullptr;
	long long *dev_counts = nullptr;
	int32_t *blk = nullptr;       // workgroup totals / offsets of the multi-workgroup compactions: (k_in/1024 + 2) * world
	const uint32_t *pending_recs = nullptr;   // received records of the pending insert (caller memory)
	int n_in = 0;
	// host mirrors
	size_t n_states = 0, open_len = 0;
	size_t n_before = 0;          // n_states before the pending expand
	int n_pop = 0, n_new = 0;
	bool pending = false;         // expand done, commit not yet
	std::vector<void *> allocs;
};

namespace {

template <typename T>
int dev_alloc(rk_astar *h, T **p, size_t count)
{
	void *q = nullptr;
	RK_HIP(hipMalloc(&q, count * sizeof(T) + 16));
	h->allocs.push_back(q);
	*p = static_cast<T *>(q);
	return RK_OK;
}

inline unsigned blocks(size_t n, unsigned per = 256) { return (unsigned)((n + per - 1) / per); }

}  // namespace

extern "C" {

static int astar_create_impl(rk_astar_t **out, size_t capacity, int max_expansions, int rank, int world)
{
	if (!out) return fail(RK_EINVAL, "rk_astar_create: null out pointer");
	if (capacity < 2 || capacity > 0x3FFFFFF0ull) return fail(RK_EINVAL, "rk_astar_create: capacity %zu out of range", capacity);
	if (max_expansions < 1 || max_expansions > (1 << 24)) return fail(RK_EINVAL, "rk_astar_create: max_expansions %d out of range", max_expansions);
	if (world < 1 || world > 255 || rank < 0 || rank >= world) return fail(RK_EINVAL, "rk_astar_create: rank %d / world %d out of range", rank, world);
	rk_astar *h = new rk_astar();
	h->cap = capacity;
	h->max_exp = max_expansions;
	h->rank = rank;
	h->world = world;
	uint64_t t = 1024;
	while (t < 2 * (uint64_t)capacity + 2) t <<= 1;
	h->mask = (uint32_t)(t - 1);
	const size_t K = 12 * (size_t)max_expansions, C1 = capacity + 1;
	const size_t KI = K * (size_t)world;          // a rank can receive every rank's children
	h->k_in = KI;
	int e = RK_OK;
	#define A(ptr, cnt) if (!e) e = dev_alloc(h, &h->ptr, (cnt))
	A(states, C1 * 5); A(G, C1); A(parents, C1); A(pact, C1); A(prank, C1); A(table, (size_t)t); A(mark, C1);
	A(open[0], C1); A(open[1], C1);
	A(exp_idx, (size_t)max_expansions); A(par_states, (size_t)max_expansions * 5); A(children, K * 5 + 64); A(solved, K + 64);
	A(seen, KI); A(child_slot, KI); A(flags, KI); A(rank_, KI); A(newway, KI); A(shortcut, KI); A(val1, KI); A(val2, KI);
	A(newrec[0], KI + 1024); A(newrec[1], KI + 1024);
	A(counters, CTR_COUNT); A(root_dev, 8);
	A(recs, K * 8 + 64); A(owner, K + 64); A(hit, KI + 64); A(dev_counts, 256);
	A(blk, (KI / SCAN_BLOCK + 2) * (size_t)world + 64);
	#undef A
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
	delete h;
	return RK_OK;
}

int rk_astar_reset(rk_astar_t *h, const int8_t *h_start_state, double lambda, void *stream)
{
	if (!h || !h_start_state) return fail(RK_EINVAL, "rk_astar_reset: null argument");
	hipStream_t st = (hipStream_t)stream;
	h->lambda = lambda;
	RK_HIP(hipMemsetAsync(h->table, 0, ((size_t)h->mask + 1) * sizeof(uint32_t), st));
	RK_HIP(hipMemsetAsync(h->mark, 0xFF, (h->cap + 1) * sizeof(uint32_t), st));
	RK_HIP(hipMemsetAsync(h->counters, 0, CTR_COUNT * sizeof(long long), st));
	RK_HIP(hipMemcpyAsync(h->root_dev, h_start_state, STATE_BYTES, hipMemcpyHostToDevice, st));
	hipLaunchKernelGGL(k_astar_root, dim3(1), dim3(64), 0, st, h->states, h->G, h->parents, h->pact, h->table, h->mask, h->open[0], h->root_dev);
	RK_HIP(hipGetLastError());
	RK_HIP(hipStreamSynchronize(st));       // the host buffer may go away after return
	h->cur = 0;
	h->n_states = 1;
	h->open_len = 1;
	h->pending = false;
	h->n_pop = h->n_new = 0;
	return RK_OK;
}

int rk_astar_expand(rk_astar_t *h, int n_expand, long long *h_info, void *stream)
{
	if (!h || !h_info) return fail(RK_EINVAL, "rk_astar_expand: null argument");
	if (h->n_states == 0) return fail(RK_ESTATE, "rk_astar_expand: reset the engine first");
	if (h->pending) return fail(RK_ESTATE, "rk_astar_expand: previous expansion not committed");
	if (n_expand < 1 || n_expand > h->max_exp) return fail(RK_EINVAL, "rk_astar_expand: n_expand %d outside 1..%d", n_expand, h->max_exp);
	hipStream_t st = (hipStream_t)stream;
	const int n_pop = (int)(h->open_len < (size_t)n_expand ? h->open_len : (size_t)n_expand);     // agents.py:238
	const int K = 12 * n_pop;
	if (h->n_states + (size_t)K > h->cap) return fail(RK_ECAPACITY, "rk_astar_expand: %zu states + %d children exceed capacity %zu", h->n_states, K, h->cap);
	h->n_before = h->n_states;
	h->n_pop = n_pop;
	h->n_new = 0;
	long long ctr[CTR_COUNT] = {0, 0, 0, 0};
	if (n_pop > 0) {
		RK_HIP(hipMemsetAsync(h->counters, 0, CTR_COUNT * sizeof(long long), st));
		hipLaunchKernelGGL(k_astar_pop, dim3(blocks((size_t)n_pop * 5)), dim3(256), 0, st, h->open[h->cur], n_pop, h->states, h->exp_idx, h->par_states);
		launch_expand12((const int8_t *)h->par_states, (int8_t *)h->children, h->solved, nullptr, (size_t)n_pop, st);
		hipLaunchKernelGGL(k_astar_lookup, dim3(blocks(K)), dim3(256), 0, st, h->children, 5, K, h->states, h->table, h->mask, h->mark, h->seen, h->child_slot);
		const int nb = (int)blocks(K, SCAN_BLOCK);
		hipLaunchKernelGGL(k_astar_flags, dim3(nb), dim3(SCAN_BLOCK), 0, st, K, h->table, h->mark, h->seen, h->child_slot, h->flags, h->rank_, h->blk);
		hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(SCAN_BLOCK), 0, st, h->blk, nb, h->counters + CTR_NEW);
		hipLaunchKernelGGL(k_astar_append<false>, dim3(blocks(K)), dim3(256), 0, st, h->children, h->solved, K, h->flags, h->rank_, h->blk, h->seen, h->child_slot,
		                   h->exp_idx, (uint32_t)h->n_before, h->states, h->G, h->parents, h->pact, h->prank, h->table, h->newway, h->val1, h->counters);
		RK_HIP(hipGetLastError());
		RK_HIP(hipMemcpyAsync(ctr, h->counters, sizeof ctr, hipMemcpyDeviceToHost, st));
		RK_HIP(hipStreamSynchronize(st));
	}
	h->n_new = (int)ctr[CTR_NEW];
	h->n_states = h->n_before + (size_t)h->n_new;
	h->pending = true;
	h_info[0] = n_pop; h_info[1] = h->n_new; h_info[2] = ctr[CTR_WON]; h_info[3] = ctr[CTR_SOLVED_IDX]; h_info[4] = (long long)h->n_states;
	return RK_OK;
}

int rk_astar_new_states_oh(rk_astar_t *h, void *d_out, int out_dtype, void *stream)
{
	if (!h || !h->pending) return fail(RK_ESTATE, "rk_astar_new_states_oh: no pending expansion");
	if (h->n_new == 0) return RK_OK;
	return rk_as_oh(RK_REPR_2024, (const int8_t *)(h->states + (h->n_before + 1) * 5), d_out, out_dtype, (size_t)h->n_new, stream);
}

// cost of the pending new states, sort, merge into the open queue (agents.py:315-317); n_pop entries leave the head
static int astar_push(rk_astar_t *h, const float *d_values, hipStream_t st)
{
	const int n_new = h->n_new, n_pop = h->n_pop;
	Rec *sorted_new = h->newrec[0];
	if (n_new > 0) {
		hipLaunchKernelGGL(k_astar_records, dim3(blocks(n_new)), dim3(256), 0, st, d_values, n_new, (uint32_t)h->n_before, h->G, h->lambda, h->newrec[0]);
		hipLaunchKernelGGL(k_sort_chunks, dim3(blocks(n_new, 1024)), dim3(512), 0, st, h->newrec[0], n_new);
		int src = 0;
		for (int L = 1024; L < n_new; L <<= 1) {
			hipLaunchKernelGGL(k_merge_pass, dim3(blocks(n_new)), dim3(256), 0, st, h->newrec[src], h->newrec[src ^ 1], n_new, L);
			src ^= 1;
		}
		sorted_new = h->newrec[src];
	}
	const int n_left = (int)(h->open_len - (size_t)n_pop);
	if (n_left + n_new > 0)
		hipLaunchKernelGGL(k_merge_two, dim3(blocks((size_t)n_left + n_new)), dim3(256), 0, st, h->open[h->cur] + n_pop, n_left, sorted_new, n_new, h->open[h->cur ^ 1]);
	h->cur ^= 1;
	h->open_len = (size_t)n_left + (size_t)n_new;
	RK_HIP(hipGetLastError());
	return RK_OK;
}

int rk_astar_commit(rk_astar_t *h, const float *d_values, void *stream)
{
	if (!h || !h->pending) return fail(RK_ESTATE, "rk_astar_commit: no pending expansion");
	if (h->world != 1) return fail(RK_ESTATE, "rk_astar_commit: sharded engines use rk_astar_shard_push");
	if (h->n_new > 0 && !d_values) return fail(RK_EINVAL, "rk_astar_commit: null values");
	hipStream_t st = (hipStream_t)stream;
	const int n_pop = h->n_pop, K = 12 * n_pop;
	if (int e = astar_push(h, d_values, st)) return e;
	if (K > 0) {
		hipLaunchKernelGGL(k_relax_1b<false>, dim3(blocks(K)), dim3(256), 0, st, K, h->newway, h->val1, h->seen, h->exp_idx, (const uint32_t *)nullptr,
		                   h->G, h->parents, h->pact, h->prank, h->counters);
		hipLaunchKernelGGL(k_relax_2a, dim3(blocks(K)), dim3(256), 0, st, K, h->flags, h->seen, h->exp_idx, h->G, h->mark, h->shortcut, h->val2);
		hipLaunchKernelGGL(k_relax_2b, dim3(blocks(n_pop)), dim3(256), 0, st, n_pop, h->shortcut, h->val2, h->seen, h->exp_idx, h->G, h->parents, h->pact);
	}
	RK_HIP(hipGetLastError());
	h->pending = false;
	return RK_OK;
}

// ---- hash-sharded mode ---------------------------------------------------------------------------------------------

int rk_shard_owner(const int8_t *h_state, int world)
{
	if (!h_state || world < 1) return fail(RK_EINVAL, "rk_shard_owner: bad argument");
	uint32_t s[5];
	memcpy(s, h_state, STATE_BYTES);
	return (int)owner_of(s, (uint32_t)world);
}

int rk_astar_shard_reset(rk_astar_t *h, const int8_t *h_start_state, double lambda, void *stream)
{
	if (!h || !h_start_state) return fail(RK_EINVAL, "rk_astar_shard_reset: null argument");
	if (int e = rk_astar_reset(h, h_start_state, lambda, stream)) return e;
	if (rk_shard_owner(h_start_state, h->world) != h->rank) {
		// not the root's owner: start empty (the root record written by reset is dropped again)
		hipStream_t st = (hipStream_t)stream;
		RK_HIP(hipMemsetAsync(h->table, 0, ((size_t)h->mask + 1) * sizeof(uint32_t), st));
		RK_HIP(hipStreamSynchronize(st));
		h->n_states = 0;
		h->open_len = 0;
	}
	h->pending = false;
	return RK_OK;
}

int rk_astar_shard_pop(rk_astar_t *h, int n_pop, void *d_send, long long *h_send_counts, void *stream)
{
	if (!h || !h_send_counts) return fail(RK_EINVAL, "rk_astar_shard_pop: null argument");
	if (h->pending) return fail(RK_ESTATE, "rk_astar_shard_pop: previous iteration not finished");
	if (n_pop < 0 || n_pop > h->max_exp || (size_t)n_pop > h->open_len) return fail(RK_EINVAL, "rk_astar_shard_pop: n_pop %d out of range", n_pop);
	hipStream_t st = (hipStream_t)stream;
	for (int w = 0; w < h->world; w++) h_send_counts[w] = 0;
	h->n_pop = n_pop;
	const int K = 12 * n_pop;
	if (K > 0) {
		if (!d_send) return fail(RK_EINVAL, "rk_astar_shard_pop: null send buffer");
		RK_HIP(hipMemsetAsync(h->dev_counts, 0, 256 * sizeof(long long), st));
		hipLaunchKernelGGL(k_astar_pop, dim3(blocks((size_t)n_pop * 5)), dim3(256), 0, st, h->open[h->cur], n_pop, h->states, h->exp_idx, h->par_states);
		launch_expand12((const int8_t *)h->par_states, (int8_t *)h->children, nullptr, nullptr, (size_t)n_pop, st);
		hipLaunchKernelGGL(k_shard_records, dim3(blocks(K)), dim3(256), 0, st, h->children, K, h->exp_idx, h->G, (uint32_t)h->rank, (uint32_t)h->world, h->recs, h->owner);
		const int nb = (int)blocks(K, SCAN_BLOCK);
		hipLaunchKernelGGL(k_shard_hist, dim3(nb), dim3(SCAN_BLOCK), 0, st, K, (uint32_t)h->world, h->owner, h->blk, nb);
		hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(SCAN_BLOCK), 0, st, h->blk, nb * h->world, (long long *)nullptr);
		hipLaunchKernelGGL(k_shard_scatter, dim3(nb), dim3(SCAN_BLOCK), 0, st, K, (uint32_t)h->world, h->recs, h->owner, h->blk, nb, (u32x4 *)d_send, h->dev_counts);
		RK_HIP(hipGetLastError());
		RK_HIP(hipMemcpyAsync(h_send_counts, h->dev_counts, (size_t)h->world * sizeof(long long), hipMemcpyDeviceToHost, st));
		RK_HIP(hipStreamSynchronize(st));
	}
	return RK_OK;
}

int rk_astar_shard_insert(rk_astar_t *h, const void *d_recv, long long n_recv, void *d_shortcuts_out, long long *h_shortcut_counts,
                          long long *h_info, void *stream)
{
	if (!h || !h_info || !h_shortcut_counts) return fail(RK_EINVAL, "rk_astar_shard_insert: null argument");
	if (h->pending) return fail(RK_ESTATE, "rk_astar_shard_insert: previous iteration not finished");
	if (n_recv < 0 || (size_t)n_recv > h->k_in) return fail(RK_ECAPACITY, "rk_astar_shard_insert: %lld records exceed the scratch capacity %zu", n_recv, h->k_in);
	if (h->n_states + (size_t)n_recv > h->cap) return fail(RK_ECAPACITY, "rk_astar_shard_insert: %zu states + %lld records exceed capacity %zu", h->n_states, n_recv, h->cap);
	hipStream_t st = (hipStream_t)stream;
	const int K = (int)n_recv;
	h->n_before = h->n_states;
	h->n_in = K;
	h->pending_recs = (const uint32_t *)d_recv;
	long long ctr[CTR_COUNT] = {0, 0, 0, 0};
	for (int w = 0; w < h->world; w++) h_shortcut_counts[w] = 0;
	if (K > 0) {
		if (!d_recv || !d_shortcuts_out) return fail(RK_EINVAL, "rk_astar_shard_insert: null buffer");
		const uint32_t *recs = (const uint32_t *)d_recv;
		RK_HIP(hipMemsetAsync(h->counters, 0, CTR_COUNT * sizeof(long long), st));
		RK_HIP(hipMemsetAsync(h->dev_counts, 0, 256 * sizeof(long long), st));
		hipLaunchKernelGGL(k_astar_lookup, dim3(blocks(K)), dim3(256), 0, st, recs, 8, K, h->states, h->table, h->mask, h->mark, h->seen, h->child_slot);
		const int nb = (int)blocks(K, SCAN_BLOCK);
		hipLaunchKernelGGL(k_astar_flags, dim3(nb), dim3(SCAN_BLOCK), 0, st, K, h->table, h->mark, h->seen, h->child_slot, h->flags, h->rank_, h->blk);
		hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(SCAN_BLOCK), 0, st, h->blk, nb, h->counters + CTR_NEW);
		hipLaunchKernelGGL(k_astar_append<true>, dim3(blocks(K)), dim3(256), 0, st, recs, (const uint8_t *)nullptr, K, h->flags, h->rank_, h->blk, h->seen, h->child_slot,
		                   (const int32_t *)nullptr, (uint32_t)h->n_before, h->states, h->G, h->parents, h->pact, h->prank, h->table, h->newway, h->val1, h->counters);
		hipLaunchKernelGGL(k_relax_1b<true>, dim3(blocks(K)), dim3(256), 0, st, K, h->newway, h->val1, h->seen, (const int32_t *)nullptr, recs,
		                   h->G, h->parents, h->pact, h->prank, h->counters);
		hipLaunchKernelGGL(k_shard_offers_count, dim3(nb), dim3(SCAN_BLOCK), 0, st, K, h->flags, h->seen, recs, h->G, (uint32_t)h->rank, h->mark,
		                   h->rank_, h->blk, h->dev_counts);
		hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(SCAN_BLOCK), 0, st, h->blk, nb, (long long *)nullptr);
		hipLaunchKernelGGL(k_shard_offers_write, dim3(nb), dim3(SCAN_BLOCK), 0, st, K, h->flags, h->seen, recs, h->G, (uint32_t)h->rank, h->rank_, h->blk,
		                   (u32x4 *)d_shortcuts_out);
		RK_HIP(hipGetLastError());
		RK_HIP(hipMemcpyAsync(ctr, h->counters, sizeof ctr, hipMemcpyDeviceToHost, st));
		RK_HIP(hipMemcpyAsync(h_shortcut_counts, h->dev_counts, (size_t)h->world * sizeof(long long), hipMemcpyDeviceToHost, st));
		RK_HIP(hipStreamSynchronize(st));
	}
	h->n_new = (int)ctr[CTR_NEW];
	h->n_states = h->n_before + (size_t)h->n_new;
	h->pending = true;
	h_info[0] = h->n_pop; h_info[1] = h->n_new; h_info[2] = ctr[CTR_WON]; h_info[3] = ctr[CTR_SOLVED_IDX]; h_info[4] = (long long)h->n_states;
	return RK_OK;
}

int rk_astar_shard_push(rk_astar_t *h, const float *d_values, void *stream)
{
	if (!h || !h->pending) return fail(RK_ESTATE, "rk_astar_shard_push: no pending insert");
	if (h->n_new > 0 && !d_values) return fail(RK_EINVAL, "rk_astar_shard_push: null values");
	if (int e = astar_push(h, d_values, (hipStream_t)stream)) return e;
	h->pending = false;
	h->n_pop = 0;
	return RK_OK;
}

int rk_astar_shard_apply_shortcuts(rk_astar_t *h, const void *d_shortcuts, long long n, void *stream)
{
	if (!h) return fail(RK_EINVAL, "rk_astar_shard_apply_shortcuts: null handle");
	if (n == 0) return RK_OK;
	if (n < 0 || (size_t)n > h->k_in || !d_shortcuts) return fail(RK_EINVAL, "rk_astar_shard_apply_shortcuts: bad record count %lld", n);
	hipStream_t st = (hipStream_t)stream;
	const u32x4 *r = (const u32x4 *)d_shortcuts;
	hipLaunchKernelGGL(k_shard_shortcut_eval, dim3(blocks((size_t)n)), dim3(256), 0, st, r, (int)n, h->G, h->mark, h->hit);
	hipLaunchKernelGGL(k_shard_shortcut_apply, dim3(blocks((size_t)n)), dim3(256), 0, st, r, (int)n, h->hit, h->G, h->parents, h->pact, h->prank, h->mark);
	hipLaunchKernelGGL(k_shard_shortcut_clear, dim3(blocks((size_t)n)), dim3(256), 0, st, r, (int)n, h->mark);
	RK_HIP(hipGetLastError());
	return RK_OK;
}

/* One hop of a cross-rank parent walk: (parent rank, parent index, action) of node `index` on this rank. */
int rk_astar_shard_parent(rk_astar_t *h, long long index, long long *h_out /* [3] */, void *stream)
{
	if (!h || !h_out) return fail(RK_EINVAL, "rk_astar_shard_parent: null argument");
	if (index < 1 || (size_t)index > h->n_states) return fail(RK_EINVAL, "rk_astar_shard_parent: index %lld outside 1..%zu", index, h->n_states);
	hipStream_t st = (hipStream_t)stream;
	int32_t p = 0;
	uint8_t a = 0, r = 0;
	RK_HIP(hipMemcpyAsync(&p, h->parents + index, sizeof p, hipMemcpyDeviceToHost, st));
	RK_HIP(hipMemcpyAsync(&a, h->pact + index, 1, hipMemcpyDeviceToHost, st));
	RK_HIP(hipMemcpyAsync(&r, h->prank + index, 1, hipMemcpyDeviceToHost, st));
	RK_HIP(hipStreamSynchronize(st));
	h_out[0] = r; h_out[1] = p; h_out[2] = a;
	return RK_OK;
}

long long rk_astar_size(const rk_astar_t *h) { return h ? (long long)h->n_states : 0; }

long long rk_astar_open_size(const rk_astar_t *h) { return h ? (long long)h->open_len : 0; }

int rk_astar_export(rk_astar_t *h, size_t first, size_t count, int8_t *h_states, double *h_G, long long *h_parents,
                    long long *h_parent_actions, void *stream)
{
	if (!h) return fail(RK_EINVAL, "rk_astar_export: null handle");
	if (first + count > h->cap + 1) return fail(RK_EINVAL, "rk_astar_export: rows %zu..%zu outside the pool", first, first + count);
	if (count == 0) return RK_OK;
	hipStream_t st = (hipStream_t)stream;
	std::vector<int32_t> g, p;
	std::vector<uint8_t> a;
	if (h_states) RK_HIP(hipMemcpyAsync(h_states, h->states + first * 5, count * STATE_BYTES, hipMemcpyDeviceToHost, st));
	if (h_G) { g.resize(count); RK_HIP(hipMemcpyAsync(g.data(), h->G + first, count * sizeof(int32_t), hipMemcpyDeviceToHost, st)); }
	if (h_parents) { p.resize(count); RK_HIP(hipMemcpyAsync(p.data(), h->parents + first, count * sizeof(int32_t), hipMemcpyDeviceToHost, st)); }
	if (h_parent_actions) { a.resize(count); RK_HIP(hipMemcpyAsync(a.data(), h->pact + first, count, hipMemcpyDeviceToHost, st)); }
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
	if (!h) return fail(RK_EINVAL, "rk_astar_path: null handle");
	if (index < 1 || (size_t)index > h->n_states) return fail(RK_EINVAL, "rk_astar_path: index %lld outside 1..%zu", index, h->n_states);
	hipStream_t st = (hipStream_t)stream;
	const size_t n = h->n_states + 1;
	std::vector<int32_t> p(n);
	std::vector<uint8_t> a(n);
	RK_HIP(hipMemcpyAsync(p.data(), h->parents, n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
	RK_HIP(hipMemcpyAsync(a.data(), h->pact, n, hipMemcpyDeviceToHost, st));
	RK_HIP(hipStreamSynchronize(st));
	std::vector<long long> rev;
	long long i = index;
	while (i != 1) {                                    // agents.py:246-250
		if (rev.size() > n) return fail(RK_ESTATE, "rk_astar_path: parent chain does not reach the root");
		rev.push_back(a[(size_t)i]);
		i = p[(size_t)i];
		if (i < 1 || (size_t)i >= n) return fail(RK_ESTATE, "rk_astar_path: broken parent chain");
	}
	const size_t len = rev.size();
	for (size_t k = 0; k < len && k < max_len; k++) h_actions[k] = rev[len - 1 - k];
	return (long long)len;
}

long long rk_astar_lookup(rk_astar_t *h, const int8_t *h_state, void *stream)
{
	if (!h || !h_state) return fail(RK_EINVAL, "rk_astar_lookup: null argument");
	hipStream_t st = (hipStream_t)stream;
	long long out = 0;
	RK_HIP(hipMemcpyAsync(h->root_dev, h_state, STATE_BYTES, hipMemcpyHostToDevice, st));
	hipLaunchKernelGGL(k_astar_find, dim3(1), dim3(64), 0, st, h->root_dev, h->states, h->table, h->mask, h->counters + 3);
	RK_HIP(hipGetLastError());
	RK_HIP(hipMemcpyAsync(&out, h->counters + 3, sizeof out, hipMemcpyDeviceToHost, st));
	RK_HIP(hipStreamSynchronize(st));
	return out;
}

long long rk_astar_export_open(rk_astar_t *h, double *h_costs, long long *h_indices, size_t max_len, void *stream)
{
	if (!h) return fail(RK_EINVAL, "rk_astar_export_open: null handle");
	hipStream_t st = (hipStream_t)stream;
	const size_t n = h->open_len < max_len ? h->open_len : max_len;
	if (n == 0) return 0;
	std::vector<Rec> r(n);
	RK_HIP(hipMemcpyAsync(r.data(), h->open[h->cur], n * sizeof(Rec), hipMemcpyDeviceToHost, st));
	RK_HIP(hipStreamSynchronize(st));
	for (size_t i = 0; i < n; i++) {
		if (h_costs) h_costs[i] = key_to_double(r[i].key);
		if (h_indices) h_indices[i] = (long long)r[i].idx;
	}
	return (long long)n;
}

}  // extern "C"
