// Batched Monte Carlo tree search on the device (reference: librubiks/solving/agents.py:415-645).
//
// T independent trees; ONE wavefront per tree (workgroup = 64 lanes, grid = T).  Lane a < 12 owns action column a
// of whatever node the tree is looking at, so a node's 12-wide rows (neighbors, P, N, W, L) are read and written as
// single coalesced row accesses, the 12-way reductions (sum of N, first arg-max of U+Q, max of the new values) are
// 16-lane shuffles, and the loops over the visited path use all 64 lanes.
//
// Per tree in HBM (cap = capacity + 1 rows, row 0 unused, root = 1): the reference's arrays (agents.py:419-427) packed
// into ONE 512-byte, 512-byte-aligned record per node -- exactly four 128-byte lines, one page -- so that a level of
// the descent (and a step of the backup) is one contiguous fetch instead of six arrays gigabytes apart:
//   +0 expanded uint32 (leaves = !expanded) | +8 V float64 | +16 N int32[12] | +64 neighbors int32[12] |
//   +112 P float64[12] | +208 W float64[12] | +304 L float64[12] | +400 stamp int32[12] |
//   +448 sumN int32 = sum of N, +456 sqrtN float64 = sqrt(sumN): maintained by the backup (64 path nodes in parallel) so
//   that the descent -- a serial pointer chase -- reads sqrt(sum N) (agents.py:580) instead of reducing and rooting it
// (stamp = simulation number of the last N increment: NumPy's `N[path, actions] += 1` counts a repeated (node, action)
// pair once), plus states int8 (cap,20) kept dense, a hash table state -> index, and the current path.
//
// Statistics are float64 and this file is compiled without fused multiply-add so that U = c*P*sqrt(sum N)/(1+N),
// Q = W - L and the arg-max reproduce NumPy's results bit for bit.
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

namespace rk {

struct MctsDev {
	int T;
	uint32_t cap1;            // rows per tree
	uint32_t tmask;           // hash table size - 1 (per tree)
	uint32_t max_path;
	double c, nu;
	uint32_t *states;
	uint8_t *nodes;           // 512-byte node records
	uint32_t *table;
	int32_t *path_nodes; uint8_t *path_actions;
	int32_t *tree;            // per-tree scalars, one 32-byte record each (TR_*): both kernels fetch them with ONE 32-byte load
	                          // instead of five dependent round trips to five arrays
	// per-simulation hand-off between expand and backup
	uint32_t *children; int32_t *child_idx; uint8_t *child_new;
};

// TR_READY: the 12 children of the pending simulation are in place (set by whoever expanded the path's leaf -- k_mcts_expand or
// the previous backup + select launch expanding ahead --, cleared by the backup that consumes them).  The decision "has this
// leaf been expanded already" therefore lives on the DEVICE: a k_mcts_expand that finds it set leaves at once, a backup that
// finds it clear reports error 3 instead of backing up garbage -- so a captured step replays correctly from any state.
enum { TR_PLEN = 0, TR_NSTATES, TR_MAXSTATES, TR_SIMS, TR_SOLVE_ACTION, TR_SOLVE_LEAF, TR_READY, TR_FLAGS /* done | solved << 8 | error << 16 */, TR_INTS = 8 };
enum { MCTS_ERR_PATH = 1, MCTS_ERR_LINK = 2, MCTS_ERR_NOT_EXPANDED = 3 };
__host__ __device__ inline int flags_done(int f) { return f & 0xFF; }
__host__ __device__ inline int flags_solved(int f) { return (f >> 8) & 0xFF; }
__host__ __device__ inline int flags_error(int f) { return (f >> 16) & 0xFF; }

struct TreeRec { int32_t v[TR_INTS]; };

__device__ __forceinline__ TreeRec load_tree(const int32_t *tree, int t)
{
	const u32x4 a = reinterpret_cast<const u32x4 *>(tree)[2 * t], b = reinterpret_cast<const u32x4 *>(tree)[2 * t + 1];
	return TreeRec{{(int32_t)a.x, (int32_t)a.y, (int32_t)a.z, (int32_t)a.w, (int32_t)b.x, (int32_t)b.y, (int32_t)b.z, (int32_t)b.w}};
}

constexpr int NODE_BYTES = 512;
constexpr int OFF_EXPANDED = 0, OFF_V = 8, OFF_N = 16, OFF_NB = 64, OFF_P = 112, OFF_W = 208, OFF_L = 304, OFF_STAMP = 400, OFF_SUMN = 448, OFF_SQRTN = 456;

struct Node {
	uint8_t *p;
	__device__ __forceinline__ uint32_t &expanded() const { return *reinterpret_cast<uint32_t *>(p + OFF_EXPANDED); }
	__device__ __forceinline__ double &V() const { return *reinterpret_cast<double *>(p + OFF_V); }
	__device__ __forceinline__ int32_t *N() const { return reinterpret_cast<int32_t *>(p + OFF_N); }
	__device__ __forceinline__ int32_t *nb() const { return reinterpret_cast<int32_t *>(p + OFF_NB); }
	__device__ __forceinline__ double *P() const { return reinterpret_cast<double *>(p + OFF_P); }
	__device__ __forceinline__ double *W() const { return reinterpret_cast<double *>(p + OFF_W); }
	__device__ __forceinline__ double *L() const { return reinterpret_cast<double *>(p + OFF_L); }
	__device__ __forceinline__ int32_t *stamp() const { return reinterpret_cast<int32_t *>(p + OFF_STAMP); }
	__device__ __forceinline__ int32_t *sumN() const { return reinterpret_cast<int32_t *>(p + OFF_SUMN); }
	__device__ __forceinline__ double &sqrtN() const { return *reinterpret_cast<double *>(p + OFF_SQRTN); }
};

__device__ __forceinline__ Node node_of(const MctsDev &d, size_t node0, int idx)
{
	return Node{d.nodes + (node0 + (size_t)idx) * NODE_BYTES};
}

__device__ __forceinline__ uint32_t mcts_hash(const uint32_t s[5])
{
	uint64_t h = 0x9E3779B97F4A7C15ull;
	#pragma unroll
	for (int j = 0; j < 5; j++) {
		h ^= s[j];
		h *= 0xFF51AFD7ED558CCDull;
		h ^= h >> 29;
	}
	return (uint32_t)(h ^ (h >> 32));
}

__device__ __forceinline__ bool same5(const uint32_t a[5], const uint32_t *p)
{
	return ((a[0] ^ p[0]) | (a[1] ^ p[1]) | (a[2] ^ p[2]) | (a[3] ^ p[3]) | (a[4] ^ p[4])) == 0;
}

// ---- 16-lane reductions on the VALU (DPP row rotations) instead of through the LDS crossbar ----------------------
// A tree level costs one memory round trip plus this arithmetic; with __shfl_xor (ds_bpermute, ~60 cycles each, a dozen
// dependent ones per level for the 64-bit arg-max) the arithmetic was as long as the memory latency.  row_ror:n rotates
// inside each row of 16 lanes, so after n = 8, 4, 2, 1 every lane of a row holds the row's reduction.
template <int N> __device__ __forceinline__ int dpp_ror_i(int v)
{
	return __builtin_amdgcn_update_dpp(v, v, 0x120 + N, 0xF, 0xF, false);
}
template <int N> __device__ __forceinline__ double dpp_ror_d(double v)
{
	const long long b = __double_as_longlong(v);
	const int lo = dpp_ror_i<N>((int)(b & 0xFFFFFFFFll)), hi = dpp_ror_i<N>((int)(b >> 32));
	return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ int row_sum_i(int v)
{
	v += dpp_ror_i<8>(v); v += dpp_ror_i<4>(v); v += dpp_ror_i<2>(v); v += dpp_ror_i<1>(v);
	return v;
}
__device__ __forceinline__ float row_max_f(float v)
{
	#define RK_STEP(N) v = fmaxf(v, __int_as_float(dpp_ror_i<N>(__float_as_int(v))))
	RK_STEP(8); RK_STEP(4); RK_STEP(2); RK_STEP(1);
	#undef RK_STEP
	return v;
}
// Maximum over a row of 16 lanes, on the descent's critical path once per tree level: the rotated copy comes from
// v_mov_b32_dpp with no "old" operand (every lane is written, so there is nothing to preserve: no register copies in front
// of the DPP pair) and the maximum is the bare v_max_f64 (fmax() first canonicalises a value the compiler cannot prove
// quiet -- one more dependent f64 instruction per step; the scores are never NaN).  3 instead of 7 instructions per step.
template <int N> __device__ __forceinline__ double dpp_ror_d_fresh(double v)
{
	const long long b = __double_as_longlong(v);
	const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xFFFFFFFFll), 0x120 + N, 0xF, 0xF, false);
	const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), 0x120 + N, 0xF, 0xF, false);
	return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double max_f64_raw(double a, double b)
{
	double r;
	asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
__device__ __forceinline__ double row_max_d(double v)
{
	#define RK_STEP(N) v = max_f64_raw(v, dpp_ror_d_fresh<N>(v))
	RK_STEP(8); RK_STEP(4); RK_STEP(2); RK_STEP(1);
	#undef RK_STEP
	return v;
}
__device__ __forceinline__ void fence_wave_to_wave()
{
	// later loads of this wave (other lanes) must see earlier stores of this wave
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64)
void k_mcts_root(MctsDev d, const uint32_t *starts, const int32_t *max_states)
{
	const int t = blockIdx.x, lane = threadIdx.x;
	uint32_t s[5];
	#pragma unroll
	for (int j = 0; j < 5; j++) s[j] = starts[(size_t)t * 5 + j];
	if (lane == 0) {
		uint32_t *st = d.states + ((size_t)t * d.cap1 + 1) * 5;
		#pragma unroll
		for (int j = 0; j < 5; j++) st[j] = s[j];
		d.table[(size_t)t * (d.tmask + 1) + (mcts_hash(s) & d.tmask)] = 1u;
		int32_t *tr = d.tree + (size_t)t * TR_INTS;
		tr[TR_NSTATES] = 1;
		tr[TR_MAXSTATES] = max_states[t];
		d.path_nodes[(size_t)t * d.max_path] = 1;
		tr[TR_PLEN] = 1;
		tr[TR_SIMS] = 0;
		tr[TR_READY] = 0;
		tr[TR_SOLVE_ACTION] = -1;
		tr[TR_SOLVE_LEAF] = -1;
		const bool solved = is_solved5(s);            // agents.py:468: a solved start returns immediately
		tr[TR_FLAGS] = solved ? (1 | (2 << 8)) : 0;
	}
}

__global__ void k_mcts_set_root_pv(MctsDev d, const float *probs, const float *values)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= d.T * 12) return;
	const int t = i / 12, k = i - 12 * t;
	const Node root = node_of(d, (size_t)t * d.cap1, 1);
	root.P()[k] = (double)probs[i];                                     // agents.py:472
	if (k == 0) root.V() = (double)values[t];                           // agents.py:473
}

__global__ void k_mcts_gather_roots(MctsDev d, uint32_t *out)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= d.T * 5) return;
	const int t = i / 5, j = i - 5 * t;
	out[i] = d.states[((size_t)t * d.cap1 + 1) * 5 + j];
}

// expand_leaf, first half (agents.py:505-543): the 12 children of `leaf`, membership, new indices, neighbour links, goal
// test.  One wave, lane a < 12 = action a.  `n` = states of the tree, `flags` = its TR_FLAGS word (not done).
__device__ __forceinline__ void expand_leaf_body(const MctsDev &d, const u32x4 *s_act, int t, int lane, int n, int max_states, int flags, int leaf)
{
	const bool active = lane < 12;
	const size_t cbase = (size_t)t * 12 + lane;
	int32_t *trw = d.tree + (size_t)t * TR_INTS;
	if (n + 12 > max_states) {                        // loop guard of agents.py:476
		if (active) d.child_new[cbase] = 0;
		if (lane == 0) { trw[TR_FLAGS] = flags | 1; trw[TR_READY] = 1; }
		return;
	}
	const size_t node0 = (size_t)t * d.cap1;
	uint32_t s[5];
	#pragma unroll
	for (int j = 0; j < 5; j++) s[j] = d.states[(node0 + leaf) * 5 + j];
	uint32_t tab[12];
	load_action_table(s_act, active ? (uint32_t)lane : 0u, tab);
	move5(s, tab);                                    // child `lane` of the leaf                  agents.py:513

	// membership (agents.py:517-520)
	uint32_t *table = d.table + (size_t)t * (d.tmask + 1);
	uint32_t slot = mcts_hash(s) & d.tmask;
	int idx = 0;
	if (active) {
		for (;;) {
			const uint32_t e = table[slot];
			if (e == 0u) break;
			if (same5(s, d.states + (node0 + e) * 5)) { idx = (int)e; break; }
			slot = (slot + 1) & d.tmask;
		}
	}
	const bool is_new = active && idx == 0;
	const unsigned long long newmask = __ballot(is_new);
	if (is_new) {                                     // new indices in action order              agents.py:523-529
		idx = n + 1 + __popcll(newmask & ((1ull << lane) - 1ull));
		while (atomicCAS(&table[slot], 0u, (uint32_t)idx) != 0u) slot = (slot + 1) & d.tmask;
		#pragma unroll
		for (int j = 0; j < 5; j++) d.states[(node0 + idx) * 5 + j] = s[j];
	}
	if (active) {
		#pragma unroll
		for (int j = 0; j < 5; j++) d.children[cbase * 5 + j] = s[j];
		d.child_idx[cbase] = idx;
		d.child_new[cbase] = is_new ? 1 : 0;
		node_of(d, node0, leaf).nb()[lane] = idx;                          // agents.py:534
		node_of(d, node0, idx).nb()[lane ^ 1] = leaf;                      // agents.py:535
	}
	const unsigned long long solvedmask = __ballot(active && is_solved5(s));   // agents.py:540-543: first solved child
	if (solvedmask != 0ull && lane == __ffsll((long long)solvedmask) - 1) {
		trw[TR_FLAGS] = (flags & 0xFF) | (1 << 8);                         // solved = 1
		trw[TR_SOLVE_ACTION] = lane;
		trw[TR_SOLVE_LEAF] = idx;
	}
	if (lane == 0) {
		node_of(d, node0, leaf).expanded() = 1u;                           // leaves[leaf] = False, agents.py:536
		trw[TR_NSTATES] = n + __popcll(newmask);
		trw[TR_READY] = 1;
	}
}

// the first expansion of a search (the root's); later ones ride at the end of the previous simulation's backup + select
// kernel when the engine expands ahead (rk_mcts_set_expand_ahead) -- this kernel then finds TR_READY set and leaves, so it
// is correct to launch it in front of every simulation (what a step captured in a hipGraph does)
__global__ __launch_bounds__(64)
void k_mcts_expand(MctsDev d)
{
	__shared__ u32x4 s_act[36];
	const int t = blockIdx.x, lane = threadIdx.x;
	stage_action_tables(s_act, lane);
	__syncthreads();
	const TreeRec tr = load_tree(d.tree, t);
	if (tr.v[TR_FLAGS] & 0xFF) {                       // done: the backup must see no new children
		if (lane < 12) d.child_new[(size_t)t * 12 + lane] = 0;
		return;
	}
	if (tr.v[TR_READY]) return;                        // expanded ahead by the previous backup + select launch
	const int leaf = d.path_nodes[(size_t)t * d.max_path + tr.v[TR_PLEN] - 1];
	expand_leaf_body(d, s_act, t, lane, tr.v[TR_NSTATES], tr.v[TR_MAXSTATES], tr.v[TR_FLAGS], leaf);
}

// The net's outputs as the kernel takes them.  IN = 0: float32 probabilities (the caller ran softmax, agents.py:551) and
// float32 values; IN = 1 / 2: the net's raw LOGITS and values in float32 / bfloat16 -- the softmax over a child's 12
// logits (exp(x - max) / sum in float32, summed in torch.softmax's own order, see child_policy) happens here, which takes two conversion kernels, the
// softmax kernel and a copy out of every simulation.
template <int IN>
__device__ __forceinline__ float net_scalar(const void *p, size_t i)
{
	if (IN == 2) return __builtin_bit_cast(float, (uint32_t)reinterpret_cast<const uint16_t *>(p)[i] << 16);
	return reinterpret_cast<const float *>(p)[i];
}

template <int IN>
__device__ __forceinline__ void child_policy(const void *probs, size_t row, int stride, float out[12])
{
	#pragma unroll
	for (int k = 0; k < 12; k++) out[k] = net_scalar<IN>(probs, row * (size_t)stride + k);
	if (IN == 0) return;
	float m = out[0];
	#pragma unroll
	for (int k = 1; k < 12; k++) m = out[k] > m ? out[k] : m;
	#pragma unroll
	for (int k = 0; k < 12; k++) out[k] = expf(out[k] - m);
	// The sum in the order torch.softmax adds on this device: its kernel for rows of <= 1024 elements (softmax_warp_forward)
	// pads the 12 logits to 16 lanes with -inf (exp -> +0) and reduces with an XOR butterfly over offsets 8, 4, 2, 1; float
	// addition commutes exactly, so every lane of that butterfly ends with this tree's value.
	const float a0 = (out[0] + out[8]) + out[4], a1 = (out[1] + out[9]) + out[5];
	const float a2 = (out[2] + out[10]) + out[6], a3 = (out[3] + out[11]) + out[7];
	const float sum = (a0 + a2) + (a1 + a3);
	#pragma unroll
	for (int k = 0; k < 12; k++) out[k] = out[k] / sum;
}

// expand_leaf, second half (agents.py:546-571) + find_leaf (agents.py:575-595)
// ahead_limit: 0 = the kernel ends with the selection (the next expansion is a launch of its own: k_mcts_expand);
// otherwise the wave that has just found its tree's next leaf expands it on the spot (expand_leaf's first half of the NEXT
// simulation: one launch and its dependent start-up less per simulation) -- as long as another simulation will follow,
// i.e. this simulation's number is below ahead_limit (< 0: no limit).  The order of the reference's steps is unchanged:
// select, expand, net, backup.
// first_tree: the launch covers the trees first_tree ... first_tree + gridDim.x - 1 and `probs` / `values` hold THEIR rows only (row 0
// = child 0 of tree first_tree): what a step that advances the batch in two halves on two streams passes (MCTSBatch overlap_halves).
template <int IN>
__global__ __launch_bounds__(64)
void k_mcts_backup_select(MctsDev d, const void *probs, const void *values, int p_stride, int v_stride, int ahead_limit, int first_tree)
{
	__shared__ u32x4 s_act[36];
	const int t = blockIdx.x + first_tree, lane = threadIdx.x;
	if (ahead_limit != 0) stage_action_tables(s_act, lane);            // (one wave: visible to it without a barrier after the fence below)
	const bool active = lane < 12;
	const size_t node0 = (size_t)t * d.cap1;
	const size_t cbase = (size_t)t * 12 + lane;
	const int32_t *pnodes = d.path_nodes + (size_t)t * d.max_path;
	uint8_t *pacts = d.path_actions + (size_t)t * d.max_path;
	// everything that does not depend on another load is requested before the first wait (the early exits used to put a
	// round trip between each of these: six of them before any work)
	const TreeRec tr = load_tree(d.tree, t);
	int32_t *trw = d.tree + (size_t)t * TR_INTS;
	const int is_done = flags_done(tr.v[TR_FLAGS]), tree_solved = flags_solved(tr.v[TR_FLAGS]);
	const int plen = tr.v[TR_PLEN], sims_before = tr.v[TR_SIMS];
	const int idx = active ? d.child_idx[cbase] : 0;
	const bool is_new = active && d.child_new[cbase] != 0;
	const size_t rbase = (size_t)blockIdx.x * 12 + lane;                   // this child's row in the net's outputs of this launch
	const float vf = active ? net_scalar<IN>(values, rbase * (size_t)v_stride) : 0.0f;
	if (is_done) {
		if (ahead_limit != 0 && active) d.child_new[cbase] = 0;            // (k_mcts_expand does this when it runs)
		return;
	}
	if (!tr.v[TR_READY]) {                                                 // nobody expanded this path's leaf: refuse, loudly
		if (lane == 0) trw[TR_FLAGS] = tr.v[TR_FLAGS] | 1 | (MCTS_ERR_NOT_EXPANDED << 16);
		return;
	}
	if (lane == 0) trw[TR_READY] = 0;                                      // consumed here; set again below if this launch expands ahead
	const int leaf = pnodes[plen - 1];
	const double v = (double)vf;
	if (is_new) {
		const Node child = node_of(d, node0, idx);
		child.V() = v;                                                     // agents.py:557
		float pk[12];
		child_policy<IN>(probs, rbase, p_stride, pk);
		#pragma unroll
		for (int k = 0; k < 12; k++) {
			child.P()[k] = (double)pk[k];                                  // agents.py:556
			child.W()[k] = v;                                              // agents.py:561
		}
	}
	float bestf = is_new ? vf : -INFINITY;                                 // v.max() over the NEW children (:559)
	bestf = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(row_max_f(bestf))));
	const bool has_new = __ballot(is_new) != 0ull;
	const double best = (double)bestf;
	if (active) node_of(d, node0, leaf).W()[lane] = is_new ? v : node_of(d, node0, idx).V();     // agents.py:560

	const int sim = sims_before + 1;
	for (int e = lane; e < plen - 1; e += 64) {
		const int act = pacts[e];
		const Node nd = node_of(d, node0, pnodes[e]);
		// the three reads of the record travel together (one round trip instead of three dependent ones); a pair that
		// occurs twice on the path has its second exchange return `sim`, so its stale read of N is never written back
		const double w = nd.W()[act];
		const int cnt = nd.N()[act];
		const int last = atomicExch(&nd.stamp()[act], sim);
		if (has_new && best > w) nd.W()[act] = best;                       // agents.py:562
		if (last != sim) {                                                 // agents.py:568 (a repeated pair counts once)
			nd.N()[act] = cnt + 1;
			// the node may sit on the path more than once (with different actions): the sum is an atomic count, and the
			// root of the LARGEST count must stay (non-negative doubles order like their bit patterns)
			const int total = atomicAdd(nd.sumN(), 1) + 1;
			atomicMax(reinterpret_cast<unsigned long long *>(&nd.sqrtN()), (unsigned long long)__double_as_longlong(sqrt((double)total)));
		}
		nd.L()[act] = 0.0;                                                 // agents.py:569
		node_of(d, node0, pnodes[e + 1]).L()[act ^ 1] = 0.0;               // agents.py:570
	}
	if (lane == 0) trw[TR_SIMS] = sim;
	if (tree_solved) {                                                     // agents.py:482-487
		if (lane == 0) {
			pacts[plen - 1] = (uint8_t)tr.v[TR_SOLVE_ACTION];
			trw[TR_FLAGS] = tr.v[TR_FLAGS] | 1;
		}
		if (ahead_limit != 0 && active) d.child_new[cbase] = 0;
		return;
	}
	fence_wave_to_wave();

	// ---- find_leaf ----
	// One round of loads per tree level: the leaf flag and the node's five rows are requested together, and the virtual
	// loss the previous step owes this node's reverse edge (agents.py:591) is applied on arrival by the lane that owns
	// that column, so there is no second dependent round trip for the read-modify-write.
	// (Tried and dropped: touching all 12 children's records as soon as `neighbors` arrives, to overlap the next level's miss
	// with this level's f64 arithmetic -- round 2, first 256 simulations: 31.7 us against 30.6 us without it; round 3, one wave
	// instruction fetching all four lines of all twelve children, whole 4096-simulation runs: 0.1618 against 0.1594 ms per
	// step (profiles/r03_mcts_variants.json).  And a copy of the previous
	// path's records in LDS (56 KB, two records per wave instruction) from which the descent reads while it follows
	// that path: 54.7 us against 49.4 us per simulation over 4096 simulations -- the copy costs more than the hits save.)
	int cur = 1, len = 1, owed_lane = -1;
	for (;;) {
		const Node nd = node_of(d, node0, cur);                            // one 512-byte record: four adjacent lines
		const int col = lane < 12 ? lane : 11;                             // idle lanes re-read column 11: no branches, no extra lines
		const uint32_t expanded = nd.expanded();
		const double sqrtN = nd.sqrtN();
		const int nA = nd.N()[col];
		const double pA = nd.P()[col], wA = nd.W()[col];
		double lval = nd.L()[col];
		const int nb = nd.nb()[col];
		if (lane == owed_lane) { lval += d.nu; nd.L()[col] = lval; }       // agents.py:591
		if (!expanded) break;
		double x = -INFINITY;
		int best_a;
		{
			double U = d.c * pA;                                           // U = c * P * sqrt(sum N) / (1 + N), left to right
			U = U * sqrtN;
			U = U / (double)(1 + nA);
			const double Q = wA - lval;
			x = active ? U + Q : -INFINITY;
		}
		// first arg-max (np.argmax), wave-uniform: the row's maximum by four DPP steps, then the first lane that holds it
		// (a third of the instructions of carrying (value, index) pairs through the reduction)
		const double mx = row_max_d(x);
		best_a = __ffsll((long long)(__ballot(x == mx) & 0xFFFull)) - 1;
		const int next = __builtin_amdgcn_readlane(nb, best_a);
		if (lane == best_a) nd.L()[col] = lval + d.nu;                         // agents.py:589
		owed_lane = best_a ^ 1;
		if (lane == 0) {
			pacts[len - 1] = (uint8_t)best_a;
			d.path_nodes[(size_t)t * d.max_path + len] = next;
		}
		len++;
		cur = next;
		if (len >= (int)d.max_path || next <= 0) {
			if (lane == 0) trw[TR_FLAGS] = tr.v[TR_FLAGS] | 1 | ((next <= 0 ? MCTS_ERR_LINK : MCTS_ERR_PATH) << 16);
			break;
		}
	}
	if (lane == 0) trw[TR_PLEN] = len;
	// ---- expand_leaf of the next simulation, first half (agents.py:505-543) ----
	const bool broke_on_error = len >= (int)d.max_path || cur <= 0;
	if (ahead_limit != 0 && !broke_on_error && (ahead_limit < 0 || sim < ahead_limit)) {
		fence_wave_to_wave();                                              // the table staged at the top, the path written above
		expand_leaf_body(d, s_act, t, lane, tr.v[TR_NSTATES], tr.v[TR_MAXSTATES], tr.v[TR_FLAGS], cur);
	} else if (ahead_limit != 0 && active) {
		d.child_new[cbase] = 0;                                            // nothing pending for this tree
	}
}


// ---- growing the pools in place (agents.py:450-460 increase_stack_size; called from agents.py:503-504) -----------------------
// After the arrays have been copied into pools of the new size: every stored state back into the (larger, cleared) hash table.
__global__ __launch_bounds__(256)
void k_mcts_rehash(MctsDev d)
{
	const int t = blockIdx.y;
	const int n = d.tree[(size_t)t * TR_INTS + TR_NSTATES];
	const size_t node0 = (size_t)t * d.cap1;
	uint32_t *table = d.table + (size_t)t * (d.tmask + 1);
	for (int idx = 1 + blockIdx.x * blockDim.x + threadIdx.x; idx <= n; idx += gridDim.x * blockDim.x) {
		uint32_t s[5];
		#pragma unroll
		for (int j = 0; j < 5; j++) s[j] = d.states[(node0 + idx) * 5 + j];
		uint32_t slot = mcts_hash(s) & d.tmask;
		while (atomicCAS(&table[slot], 0u, (uint32_t)idx) != 0u) slot = (slot + 1) & d.tmask;
	}
}

// New state budgets after a growth; a tree that had stopped at the loop guard only (agents.py:476: not solved, no error) and
// has room again goes on: its descent is complete, its leaf not expanded, so the next expansion (k_mcts_expand) picks it up.
__global__ void k_mcts_set_budgets(MctsDev d, const int32_t *max_states)
{
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= d.T) return;
	int32_t *tr = d.tree + (size_t)t * TR_INTS;
	tr[TR_MAXSTATES] = max_states[t];
	const int f = tr[TR_FLAGS];
	if (flags_done(f) && !flags_solved(f) && !flags_error(f) && tr[TR_NSTATES] + 12 <= max_states[t]) {
		tr[TR_FLAGS] = f & ~0xFF;
		tr[TR_READY] = 0;
	}
}

// ---- the explored graph of a solved tree (agents.py:597-633) ---------------------------------------------------------------
// _complete_graph (agents.py:597-611): every leaf is linked to those of its 12 children that are already in the graph, both
// ways.  One thread per (node, action) of 16 lanes per node; trees that have not solved are left alone (the reference
// completes the graph only on a win, agents.py:483-486).  A child's reverse link neighbors[child, rev(a)] can only ever
// name this leaf (moves are invertible), so the writes of different threads never disagree.
__global__ __launch_bounds__(256)
void k_mcts_complete(MctsDev d)
{
	__shared__ u32x4 s_act[36];
	stage_action_tables(s_act, threadIdx.x);
	__syncthreads();
	const int t = blockIdx.y;
	const int32_t *tr = d.tree + (size_t)t * TR_INTS;
	if (flags_solved(tr[TR_FLAGS]) != 1) return;
	const int n = tr[TR_NSTATES];
	const size_t node0 = (size_t)t * d.cap1;
	const uint32_t *table = d.table + (size_t)t * (d.tmask + 1);
	const int a = threadIdx.x & 15;
	for (int idx = 1 + (blockIdx.x * blockDim.x + threadIdx.x) / 16; idx <= n; idx += gridDim.x * blockDim.x / 16) {
		const Node leaf = node_of(d, node0, idx);
		if (a >= 12 || leaf.expanded()) continue;                          // np.where(self.leaves[:len(self)+1])[0][1:]
		uint32_t s[5];
		#pragma unroll
		for (int j = 0; j < 5; j++) s[j] = d.states[(node0 + idx) * 5 + j];
		uint32_t tab[12];
		load_action_table(s_act, (uint32_t)a, tab);
		move5(s, tab);
		uint32_t slot = mcts_hash(s) & d.tmask;
		int child = 0;
		for (;;) {
			const uint32_t e = table[slot];
			if (e == 0u) break;
			if (same5(s, d.states + (node0 + e) * 5)) { child = (int)e; break; }
			slot = (slot + 1) & d.tmask;
		}
		leaf.nb()[a] = child;                                              // agents.py:607 (0 = not in the graph)
		if (child) node_of(d, node0, child).nb()[a ^ 1] = idx;             // agents.py:608
	}
}

// _shorten_action_queue (agents.py:613-633): breadth-first search from the root (index 1) through `neighbors` until the
// solved state's index turns up; the path back through the BFS tree replaces the action queue.  The reference's queue is
// sequential: a node's BFS parent is the FIRST (node, action) pair, in queue order, that reaches it, and nodes enter the
// queue in that order.  One workgroup per tree reproduces it level by level: every edge (position p of the frontier, action a)
// has the number e = 12 p + a; pass A lets every edge claim its unvisited target with atomicMin(e); pass B walks the edges in
// order again and appends the winners -- in edge order, by an exclusive scan -- to the next frontier.  The first edge that
// reaches the solved index is therefore exactly the one the reference's loop returns on.
// Reads of words that other waves of the workgroup wrote go around the L1 (agent-scope atomic loads).
struct BfsDev {
	uint32_t *claim;          // [T][cap1]  smallest edge number that reached the node (0xFFFFFFFF: none yet)
	int32_t *from;            // [T][cap1]  BFS parent (0: not visited; the root holds -1)
	uint8_t *act;             // [T][cap1]  action from the BFS parent
	int32_t *front[2];        // [T][cap1]  frontier, current and next
	int32_t *len;             // [T]        length of the path found, -1: none (tree not solved, or the root itself is the goal)
	uint8_t *actions;         // [T][max_path]
};

__device__ __forceinline__ int ld_i32(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_u32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(1024)
void k_mcts_bfs(MctsDev d, BfsDev b)
{
	__shared__ int s_wave[16];
	__shared__ int s_count, s_found;
	const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int32_t *tr = d.tree + (size_t)t * TR_INTS;
	const int goal = tr[TR_SOLVE_LEAF];
	if (flags_solved(tr[TR_FLAGS]) != 1 || goal <= 1) {                   // agents.py:614: the queue stays as it is
		if (tid == 0) b.len[t] = -1;
		return;
	}
	const size_t node0 = (size_t)t * d.cap1;
	uint32_t *claim = b.claim + node0;
	int32_t *from = b.from + node0;
	uint8_t *act = b.act + node0;
	int32_t *cur = b.front[0] + node0, *nxt = b.front[1] + node0;
	if (tid == 0) { cur[0] = 1; from[1] = -1; s_found = 0; }
	__syncthreads();
	int fsize = 1, depth = 0;
	while (fsize > 0) {
		const long long total = 12ll * fsize;
		for (long long e = tid; e < total; e += 1024) {                    // pass A: claims
			const int v = ld_i32(cur + e / 12), a = (int)(e % 12);
			const int w = node_of(d, node0, v).nb()[a];
			if (w > 0 && ld_i32(from + w) == 0) atomicMin(claim + w, (uint32_t)e);
		}
		__syncthreads();
		if (tid == 0) s_count = 0;
		__syncthreads();
		for (long long base = 0; base < total; base += 1024) {             // pass B: winners, in edge order
			const long long e = base + tid;
			int v = 0, a = 0, w = 0;
			bool win = false;
			if (e < total) {
				v = ld_i32(cur + e / 12); a = (int)(e % 12);
				w = node_of(d, node0, v).nb()[a];
				win = w > 0 && ld_i32(from + w) == 0 && ld_u32(claim + w) == (uint32_t)e;
			}
			const unsigned long long m = __ballot(win);
			if (lane == 0) s_wave[wv] = __popcll(m);
			__syncthreads();
			int before = 0, tot = 0;
			#pragma unroll
			for (int k = 0; k < 16; k++) { const int c = s_wave[k]; before += k < wv ? c : 0; tot += c; }
			const int at = s_count + before + __popcll(m & ((1ull << lane) - 1ull));
			if (win) {
				nxt[at] = w;
				__hip_atomic_store(from + w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				act[w] = (uint8_t)a;
				if (w == goal) s_found = 1;
			}
			__syncthreads();
			if (tid == 0) s_count += tot;
			__syncthreads();
		}
		depth++;
		fsize = s_count;
		int32_t *sw = cur; cur = nxt; nxt = sw;
		if (s_found) break;
		__syncthreads();
	}
	if (tid == 0) {
		if (!s_found || depth > (int)d.max_path) { b.len[t] = -1; return; }
		uint8_t *out = b.actions + (size_t)t * d.max_path;
		int v = goal;
		for (int k = depth - 1; k >= 0; k--) {                             // agents.py:624-628
			out[k] = __hip_atomic_load(act + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			v = ld_i32(from + v);
		}
		b.len[t] = depth;
	}
}

}  // namespace rk

using namespace rk;

struct rk_mcts {
	MctsDev d{};
	size_t capacity = 0;
	std::vector<void *> allocs;
	uint32_t *starts_dev = nullptr;
	int32_t *max_states_dev = nullptr;
	bool ready = false;
	int ahead_limit = 0;          // rk_mcts_set_expand_ahead: 0 off, < 0 always, > 0 while the simulation number is below it
	bool ahead = false;           // the last backup + select launch already expanded the leaves it found
	int32_t *tree_host = nullptr; // page-locked landing place of the per-tree records: a status poll is one direct copy, no staging
	BfsDev bfs{};                 // scratch of rk_mcts_search_graph, allocated at its first call (and again after a growth)
	std::vector<void *> bfs_allocs;
	bool bfs_valid = false;       // rk_mcts_search_graph has run since the last reset / growth
};

namespace {

template <typename T>
int mcts_alloc(rk_mcts *h, T **p, size_t count)
{
	void *q = nullptr;
	RK_HIP(hipMalloc(&q, count * sizeof(T) + 16));
	h->allocs.push_back(q);
	*p = static_cast<T *>(q);
	return RK_OK;
}

inline unsigned nblocks(size_t n, unsigned per = 256) { return (unsigned)((n + per - 1) / per); }

// is `st` recording into a hipGraph? (the legacy default stream cannot be)
inline bool capturing(hipStream_t st)
{
	if (st == nullptr) return false;
	hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
	if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); return false; }
	return cap != hipStreamCaptureStatusNone;
}

// `rows` blocks of `width` bytes, `spitch` / `dpitch` bytes apart, device to device: one plain copy per block (a tree's block can
// be gigabytes wide, which is not what the 2-D copy engine path is made for), the 2-D call only for very many trees
hipError_t strided_copy(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t rows, hipStream_t st)
{
	if (rows > 4096) return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, hipMemcpyDeviceToDevice, st);
	for (size_t r = 0; r < rows; r++) {
		const hipError_t e = hipMemcpyAsync((char *)dst + r * dpitch, (const char *)src + r * spitch, width, hipMemcpyDeviceToDevice, st);
		if (e != hipSuccess) return e;
	}
	return hipSuccess;
}

}  // namespace

extern "C" {

int rk_mcts_create(rk_mcts_t **out, int n_trees, size_t capacity_per_tree, size_t max_path)
{
	if (!out) return fail(RK_EINVAL, "rk_mcts_create: null out pointer");
	if (n_trees < 1 || n_trees > (1 << 20)) return fail(RK_EINVAL, "rk_mcts_create: n_trees %d out of range", n_trees);
	if (capacity_per_tree < 13 || capacity_per_tree > 0x3FFFFFF0ull) return fail(RK_EINVAL, "rk_mcts_create: capacity %zu out of range", capacity_per_tree);
	if (max_path < 2 || max_path > (1u << 30)) return fail(RK_EINVAL, "rk_mcts_create: max_path %zu out of range", max_path);
	rk_mcts *h = new rk_mcts();
	h->capacity = capacity_per_tree;
	MctsDev &d = h->d;
	d.T = n_trees;
	d.cap1 = (uint32_t)(capacity_per_tree + 1);
	uint64_t ts = 64;
	while (ts < 2ull * d.cap1) ts <<= 1;
	d.tmask = (uint32_t)(ts - 1);
	d.max_path = (uint32_t)max_path;
	const size_t T = (size_t)n_trees, rows = T * d.cap1;
	int e = RK_OK;
	#define A(ptr, cnt) if (!e) e = mcts_alloc(h, &d.ptr, (cnt))
	A(states, rows * 5); A(nodes, (rows + 1) * NODE_BYTES);
	A(table, T * (size_t)ts);
	A(path_nodes, T * max_path); A(path_actions, T * max_path);
	A(tree, T * TR_INTS + 16);
	A(children, T * 12 * 5 + 64); A(child_idx, T * 12); A(child_new, T * 12);
	#undef A
	if (!e) e = mcts_alloc(h, &h->starts_dev, T * 5);
	if (!e) e = mcts_alloc(h, &h->max_states_dev, T);
	if (!e && hipHostMalloc((void **)&h->tree_host, T * TR_INTS * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); h->tree_host = nullptr; }
	if (e) { rk_mcts_destroy(h); return e; }
	*out = h;
	return RK_OK;
}

int rk_mcts_destroy(rk_mcts_t *h)
{
	if (!h) return RK_OK;
	for (void *p : h->allocs) (void)hipFree(p);
	for (void *p : h->bfs_allocs) (void)hipFree(p);
	if (h->tree_host != nullptr) (void)hipHostFree(h->tree_host);
	delete h;
	return RK_OK;
}

int rk_mcts_reset(rk_mcts_t *h, const int8_t *h_start_states, const long long *h_max_states, double c, double nu, void *stream)
{
	if (!h || !h_start_states) return fail(RK_EINVAL, "rk_mcts_reset: null argument");
	hipStream_t st = (hipStream_t)stream;
	MctsDev &d = h->d;
	const size_t T = (size_t)d.T, rows = T * d.cap1;
	d.c = c;
	d.nu = nu;
	std::vector<int32_t> ms(T);
	for (size_t t = 0; t < T; t++) {
		long long m = h_max_states ? h_max_states[t] : (long long)h->capacity;
		if (m > (long long)h->capacity) m = (long long)h->capacity;       // never index past the pool
		ms[t] = (int32_t)(m < 0 ? 0 : m);
	}
	// neighbors, P, V, N, W, L zero and every node a leaf (expanded = 0)                      agents.py:441-447
	RK_HIP(hipMemsetAsync(d.nodes, 0, rows * NODE_BYTES, st));
	RK_HIP(hipMemsetAsync(d.table, 0, T * ((size_t)d.tmask + 1) * sizeof(uint32_t), st));
	RK_HIP(hipMemsetAsync(d.child_new, 0, T * 12, st));
	RK_HIP(hipMemsetAsync(d.child_idx, 0, T * 12 * sizeof(int32_t), st));
	RK_HIP(hipMemsetAsync(d.children, 0, (T * 12 * 5) * sizeof(uint32_t), st));
	RK_HIP(hipMemcpyAsync(h->max_states_dev, ms.data(), T * sizeof(int32_t), hipMemcpyHostToDevice, st));
	RK_HIP(hipMemcpyAsync(h->starts_dev, h_start_states, T * STATE_BYTES, hipMemcpyHostToDevice, st));
	hipLaunchKernelGGL(k_mcts_root, dim3(d.T), dim3(64), 0, st, d, h->starts_dev, h->max_states_dev);
	RK_HIP(hipGetLastError());
	RK_HIP(hipStreamSynchronize(st));           // host buffers may go away after return
	h->ready = true;
	h->ahead = false;
	return RK_OK;
}


/* Grows every tree's pool to new_capacity states (>= the current one) and the path arrays to new_max_path entries, keeping
 * everything the trees hold (agents.py:450-460): new arrays, device-to-device copies, the hash tables rebuilt by one kernel. */
int rk_mcts_grow(rk_mcts_t *h, size_t new_capacity, size_t new_max_path, const long long *h_max_states, void *stream)
{
	if (!h || !h->ready) return fail(RK_ESTATE, "rk_mcts_grow: reset the engine first");
	if (new_capacity < h->capacity || new_capacity > 0x3FFFFFF0ull) return fail(RK_EINVAL, "rk_mcts_grow: capacity %zu out of range (now %zu)", new_capacity, h->capacity);
	if (new_max_path < h->d.max_path) new_max_path = h->d.max_path;
	if (new_max_path > (1u << 30)) return fail(RK_EINVAL, "rk_mcts_grow: max_path %zu out of range", new_max_path);
	hipStream_t st = (hipStream_t)stream;
	const MctsDev old = h->d;
	MctsDev d = old;
	const size_t T = (size_t)d.T;
	d.cap1 = (uint32_t)(new_capacity + 1);
	uint64_t ts = 64;
	while (ts < 2ull * d.cap1) ts <<= 1;
	d.tmask = (uint32_t)(ts - 1);
	d.max_path = (uint32_t)new_max_path;
	const size_t rows = T * d.cap1;
	std::vector<void *> fresh;
	auto get = [&](size_t bytes) -> void * { void *q = nullptr; if (hipMalloc(&q, bytes + 16) != hipSuccess) return nullptr; fresh.push_back(q); return q; };
	const bool pool = d.cap1 != old.cap1, paths = d.max_path != old.max_path;
	if (pool) {
		d.states = (uint32_t *)get(rows * 5 * sizeof(uint32_t));
		d.nodes = (uint8_t *)get((rows + 1) * NODE_BYTES);
		d.table = (uint32_t *)get(T * (size_t)ts * sizeof(uint32_t));
	}
	if (paths) {
		d.path_nodes = (int32_t *)get(T * new_max_path * sizeof(int32_t));
		d.path_actions = (uint8_t *)get(T * new_max_path);
	}
	if ((pool && (!d.states || !d.nodes || !d.table)) || (paths && (!d.path_nodes || !d.path_actions))) {
		for (void *q : fresh) (void)hipFree(q);
		(void)hipGetLastError();
		return fail(RK_ECAPACITY, "rk_mcts_grow: no device memory for %d trees of %zu states", d.T, new_capacity);
	}
	std::vector<int32_t> ms(T);
	for (size_t t = 0; t < T; t++) {
		long long m = h_max_states ? h_max_states[t] : (long long)new_capacity;
		if (m > (long long)new_capacity) m = (long long)new_capacity;
		ms[t] = (int32_t)(m < 0 ? 0 : m);
	}
	// the copies, the rehash and the budgets; an error in here leaves the engine as it was (the new arrays are given back)
	auto fill = [&]() -> hipError_t {
		#define RK_TRY(call) do { const hipError_t e_ = (call); if (e_ != hipSuccess) return e_; } while (0)
		if (pool) {
			RK_TRY(hipMemsetAsync(d.nodes, 0, rows * NODE_BYTES, st));     // rows beyond the old pool: leaves, all statistics zero
			RK_TRY(hipMemsetAsync(d.table, 0, T * (size_t)ts * sizeof(uint32_t), st));
			RK_TRY(strided_copy(d.states, (size_t)d.cap1 * STATE_BYTES, old.states, (size_t)old.cap1 * STATE_BYTES, (size_t)old.cap1 * STATE_BYTES, T, st));
			RK_TRY(strided_copy(d.nodes, (size_t)d.cap1 * NODE_BYTES, old.nodes, (size_t)old.cap1 * NODE_BYTES, (size_t)old.cap1 * NODE_BYTES, T, st));
		}
		if (paths) {
			RK_TRY(strided_copy(d.path_nodes, new_max_path * 4, old.path_nodes, (size_t)old.max_path * 4, (size_t)old.max_path * 4, T, st));
			RK_TRY(strided_copy(d.path_actions, new_max_path, old.path_actions, (size_t)old.max_path, (size_t)old.max_path, T, st));
		}
		if (pool) {
			hipLaunchKernelGGL(k_mcts_rehash, dim3(std::min<unsigned>(nblocks(old.cap1), 4096u), d.T), dim3(256), 0, st, d);
			RK_TRY(hipGetLastError());
		}
		RK_TRY(hipMemcpyAsync(h->max_states_dev, ms.data(), T * sizeof(int32_t), hipMemcpyHostToDevice, st));
		hipLaunchKernelGGL(k_mcts_set_budgets, dim3(nblocks(T)), dim3(256), 0, st, d, h->max_states_dev);
		RK_TRY(hipGetLastError());
		RK_TRY(hipStreamSynchronize(st));                                  // `ms` and the old arrays go away now
		#undef RK_TRY
		return hipSuccess;
	};
	if (const hipError_t e = fill(); e != hipSuccess) {
		(void)hipStreamSynchronize(st);                                    // nothing may still write into what is freed next
		for (void *q : fresh) (void)hipFree(q);
		(void)hipGetLastError();
		return fail(RK_EHIP, "rk_mcts_grow: %s", hipGetErrorString(e));
	}
	auto drop = [&](void *q) {
		for (size_t i = 0; i < h->allocs.size(); i++) if (h->allocs[i] == q) { h->allocs.erase(h->allocs.begin() + (long)i); break; }
		(void)hipFree(q);
	};
	if (pool) { drop(old.states); drop(old.nodes); drop(old.table); }
	if (paths) { drop(old.path_nodes); drop(old.path_actions); }
	for (void *q : fresh) h->allocs.push_back(q);
	h->d = d;
	h->capacity = new_capacity;
	h->ahead = false;                                                      // what is pending is decided on the device (TR_READY)
	for (void *q : h->bfs_allocs) (void)hipFree(q);                        // sized by the pool: allocated again when next needed
	h->bfs_allocs.clear();
	h->bfs = BfsDev{};
	h->bfs_valid = false;
	return RK_OK;
}

/* agents.py:483-486 for every tree that has solved: _complete_graph (:597-611) and the breadth-first search of
 * _shorten_action_queue (:613-633), both on the device; rk_mcts_graph_path then returns the shortened queue. */
int rk_mcts_search_graph(rk_mcts_t *h, void *stream)
{
	if (!h || !h->ready) return fail(RK_ESTATE, "rk_mcts_search_graph: reset the engine first");
	hipStream_t st = (hipStream_t)stream;
	const MctsDev &d = h->d;
	const size_t T = (size_t)d.T, rows = T * d.cap1;
	if (h->bfs.claim == nullptr) {
		BfsDev b{};
		auto get = [&](size_t bytes) -> void * { void *q = nullptr; if (hipMalloc(&q, bytes + 16) != hipSuccess) return nullptr; h->bfs_allocs.push_back(q); return q; };
		b.claim = (uint32_t *)get(rows * 4); b.from = (int32_t *)get(rows * 4); b.act = (uint8_t *)get(rows);
		b.front[0] = (int32_t *)get(rows * 4); b.front[1] = (int32_t *)get(rows * 4);
		b.len = (int32_t *)get(T * 4); b.actions = (uint8_t *)get(T * (size_t)d.max_path);
		if (!b.claim || !b.from || !b.act || !b.front[0] || !b.front[1] || !b.len || !b.actions) {
			for (void *q : h->bfs_allocs) (void)hipFree(q);
			h->bfs_allocs.clear();
			(void)hipGetLastError();
			return fail(RK_ECAPACITY, "rk_mcts_search_graph: no device memory for the search scratch of %d trees", d.T);
		}
		h->bfs = b;
	}
	RK_HIP(hipMemsetAsync(h->bfs.claim, 0xFF, rows * 4, st));
	RK_HIP(hipMemsetAsync(h->bfs.from, 0, rows * 4, st));
	hipLaunchKernelGGL(k_mcts_complete, dim3(std::min<unsigned>(nblocks((size_t)d.cap1 * 16), 8192u), d.T), dim3(256), 0, st, d);
	hipLaunchKernelGGL(k_mcts_bfs, dim3(d.T), dim3(1024), 0, st, d, h->bfs);
	RK_HIP(hipGetLastError());
	h->bfs_valid = true;
	return RK_OK;
}

/* The action queue rk_mcts_search_graph found for `tree` (agents.py:624-628): returns its length, or -1 when the tree has none
 * (not solved, or nothing to shorten: the queue of rk_mcts_path stands).  Synchronises. */
long long rk_mcts_graph_path(rk_mcts_t *h, int tree, long long *h_actions, size_t max_len, void *stream)
{
	if (!h || !h->ready) return fail(RK_ESTATE, "rk_mcts_graph_path: reset the engine first");
	if (!h->bfs_valid) return fail(RK_ESTATE, "rk_mcts_graph_path: call rk_mcts_search_graph first");
	const MctsDev &d = h->d;
	if (tree < 0 || tree >= d.T) return fail(RK_EINVAL, "rk_mcts_graph_path: tree %d out of range", tree);
	hipStream_t st = (hipStream_t)stream;
	int32_t len = -1;
	RK_HIP(hipMemcpyAsync(&len, h->bfs.len + tree, sizeof len, hipMemcpyDeviceToHost, st));
	RK_HIP(hipStreamSynchronize(st));
	if (len <= 0) return len < 0 ? -1 : 0;
	std::vector<uint8_t> acts((size_t)len);
	RK_HIP(hipMemcpyAsync(acts.data(), h->bfs.actions + (size_t)tree * d.max_path, (size_t)len, hipMemcpyDeviceToHost, st));
	RK_HIP(hipStreamSynchronize(st));
	for (size_t i = 0; i < (size_t)len && i < max_len; i++) if (h_actions) h_actions[i] = acts[i];
	return len;
}

int rk_mcts_set_expand_ahead(rk_mcts_t *h, long long sim_limit)
{
	if (!h) return fail(RK_EINVAL, "rk_mcts_set_expand_ahead: null handle");
	h->ahead_limit = sim_limit < 0 ? -1 : (sim_limit > INT_MAX ? INT_MAX : (int)sim_limit);
	return RK_OK;
}

int rk_mcts_roots_oh(rk_mcts_t *h, void *d_out, int out_dtype, void *stream)
{
	if (!h || !h->ready) return fail(RK_ESTATE, "rk_mcts_roots_oh: reset the engine first");
	MctsDev &d = h->d;
	if (out_dtype == RK_OH_STATES) {
		if (!d_out) return fail(RK_EINVAL, "rk_mcts_roots_oh: null output");
		hipLaunchKernelGGL(k_mcts_gather_roots, dim3(nblocks((size_t)d.T * 5)), dim3(256), 0, (hipStream_t)stream, d, (uint32_t *)d_out);
		RK_HIP(hipGetLastError());
		return RK_OK;
	}
	hipLaunchKernelGGL(k_mcts_gather_roots, dim3(nblocks((size_t)d.T * 5)), dim3(256), 0, (hipStream_t)stream, d, d.children);
	RK_HIP(hipGetLastError());
	return rk_as_oh(RK_REPR_2024, (const int8_t *)d.children, d_out, out_dtype, (size_t)d.T, stream);
}

int rk_mcts_set_root_pv(rk_mcts_t *h, const float *d_probs, const float *d_values, void *stream)
{
	if (!h || !h->ready) return fail(RK_ESTATE, "rk_mcts_set_root_pv: reset the engine first");
	if (!d_probs || !d_values) return fail(RK_EINVAL, "rk_mcts_set_root_pv: null pointer");
	hipLaunchKernelGGL(k_mcts_set_root_pv, dim3(nblocks((size_t)h->d.T * 12)), dim3(256), 0, (hipStream_t)stream, h->d, d_probs, d_values);
	RK_HIP(hipGetLastError());
	return RK_OK;
}

int rk_mcts_expand(rk_mcts_t *h, void *stream)
{
	if (!h || !h->ready) return fail(RK_ESTATE, "rk_mcts_expand: reset the engine first");
	// The previous backup + select launch may have expanded the leaves already (expand ahead).  Eagerly the host knows and skips
	// the launch; a call that is being CAPTURED into a hipGraph always records the kernel, which takes the decision on the device
	// (TR_READY) at every replay -- whatever state the graph was captured in and whatever state it is replayed from.  The host's
	// flag is only ever an optimisation of eager sequences: captured calls execute nothing and leave it alone.
	if (!capturing((hipStream_t)stream)) {
		const bool skip = h->ahead;
		h->ahead = false;
		if (skip) return RK_OK;
	}
	hipLaunchKernelGGL(k_mcts_expand, dim3(h->d.T), dim3(64), 0, (hipStream_t)stream, h->d);
	RK_HIP(hipGetLastError());
	return RK_OK;
}

int rk_mcts_children_oh(rk_mcts_t *h, void *d_out, int out_dtype, void *stream)
{
	if (!h || !h->ready) return fail(RK_ESTATE, "rk_mcts_children_oh: reset the engine first");
	if (out_dtype == RK_OH_STATES) {              // a net whose first layer reads states (rk_ohl_*): hand over the (T*12, 20) int8 rows
		if (!d_out) return fail(RK_EINVAL, "rk_mcts_children_oh: null output");
		RK_HIP(hipMemcpyAsync(d_out, h->d.children, (size_t)h->d.T * 12 * STATE_BYTES, hipMemcpyDeviceToDevice, (hipStream_t)stream));
		return RK_OK;
	}
	return rk_as_oh(RK_REPR_2024, (const int8_t *)h->d.children, d_out, out_dtype, (size_t)h->d.T * 12, stream);
}

int rk_mcts_backup_select(rk_mcts_t *h, const float *d_probs, const float *d_values, void *stream)
{
	if (!h || !h->ready) return fail(RK_ESTATE, "rk_mcts_backup_select: reset the engine first");
	if (!d_probs || !d_values) return fail(RK_EINVAL, "rk_mcts_backup_select: null pointer");
	hipLaunchKernelGGL(k_mcts_backup_select<0>, dim3(h->d.T), dim3(64), 0, (hipStream_t)stream, h->d, (const void *)d_probs, (const void *)d_values, 12, 1, h->ahead_limit, 0);
	RK_HIP(hipGetLastError());
	if (!capturing((hipStream_t)stream)) h->ahead = h->ahead_limit != 0;
	return RK_OK;
}

static int backup_select_logits_impl(rk_mcts_t *h, int first_tree, int n_trees, const void *d_logits, int logits_stride, const void *d_values, int values_stride,
                                     int dtype, void *stream)
{
	if (!h || !h->ready) return fail(RK_ESTATE, "rk_mcts_backup_select_logits: reset the engine first");
	if (!d_logits || !d_values) return fail(RK_EINVAL, "rk_mcts_backup_select_logits: null pointer");
	if (logits_stride < 12 || values_stride < 1) return fail(RK_EINVAL, "rk_mcts_backup_select_logits: strides are in elements, at least 12 and 1");
	if (first_tree < 0 || n_trees < 1 || first_tree + n_trees > h->d.T)
		return fail(RK_EINVAL, "rk_mcts_backup_select_logits_range: trees %d..%d outside 0..%d", first_tree, first_tree + n_trees, h->d.T);
	if (dtype == RK_OH_F32)
		hipLaunchKernelGGL(k_mcts_backup_select<1>, dim3(n_trees), dim3(64), 0, (hipStream_t)stream, h->d, d_logits, d_values, logits_stride, values_stride, h->ahead_limit, first_tree);
	else if (dtype == RK_OH_BF16)
		hipLaunchKernelGGL(k_mcts_backup_select<2>, dim3(n_trees), dim3(64), 0, (hipStream_t)stream, h->d, d_logits, d_values, logits_stride, values_stride, h->ahead_limit, first_tree);
	else
		return fail(RK_EINVAL, "rk_mcts_backup_select_logits: logits and values must be float32 or bfloat16");
	RK_HIP(hipGetLastError());
	if (!capturing((hipStream_t)stream)) h->ahead = h->ahead_limit != 0;
	return RK_OK;
}

int rk_mcts_backup_select_logits(rk_mcts_t *h, const void *d_logits, int logits_stride, const void *d_values, int values_stride, int dtype, void *stream)
{
	return backup_select_logits_impl(h, 0, h ? h->d.T : 1, d_logits, logits_stride, d_values, values_stride, dtype, stream);
}

int rk_mcts_backup_select_logits_range(rk_mcts_t *h, int first_tree, int n_trees, const void *d_logits, int logits_stride, const void *d_values,
                                       int values_stride, int dtype, void *stream)
{
	return backup_select_logits_impl(h, first_tree, n_trees, d_logits, logits_stride, d_values, values_stride, dtype, stream);
}

const int8_t *rk_mcts_children(rk_mcts_t *h)
{
	return h ? reinterpret_cast<const int8_t *>(h->d.children) : nullptr;
}

int rk_mcts_status(rk_mcts_t *h, long long *h_status, void *stream)
{
	if (!h || !h->ready || !h_status) return fail(RK_EINVAL, "rk_mcts_status: bad argument");
	hipStream_t st = (hipStream_t)stream;
	const MctsDev &d = h->d;
	const size_t T = (size_t)d.T;
	std::vector<int32_t> pageable;
	int32_t *rec = h->tree_host;
	if (rec == nullptr) { pageable.resize(T * TR_INTS); rec = pageable.data(); }
	RK_HIP(hipMemcpyAsync(rec, d.tree, T * TR_INTS * sizeof(int32_t), hipMemcpyDeviceToHost, st));
	RK_HIP(hipStreamSynchronize(st));
	for (size_t t = 0; t < T; t++) {
		const int32_t *tr = rec + t * TR_INTS;
		long long *r = h_status + 6 * t;
		r[0] = flags_done(tr[TR_FLAGS]); r[1] = flags_solved(tr[TR_FLAGS]); r[2] = tr[TR_NSTATES]; r[3] = tr[TR_SIMS]; r[4] = tr[TR_PLEN];
		r[5] = flags_error(tr[TR_FLAGS]);
	}
	return RK_OK;
}

int rk_mcts_export(rk_mcts_t *h, int tree, size_t first, size_t count, int8_t *h_states, long long *h_neighbors,
                   uint8_t *h_leaves, double *h_P, double *h_V, long long *h_N, double *h_W, double *h_L, void *stream)
{
	if (!h || !h->ready) return fail(RK_ESTATE, "rk_mcts_export: reset the engine first");
	const MctsDev &d = h->d;
	if (tree < 0 || tree >= d.T) return fail(RK_EINVAL, "rk_mcts_export: tree %d out of range", tree);
	if (first + count > d.cap1) return fail(RK_EINVAL, "rk_mcts_export: rows outside the pool");
	if (count == 0) return RK_OK;
	hipStream_t st = (hipStream_t)stream;
	const size_t r0 = (size_t)tree * d.cap1 + first;
	if (h_states) RK_HIP(hipMemcpyAsync(h_states, d.states + r0 * 5, count * STATE_BYTES, hipMemcpyDeviceToHost, st));
	std::vector<uint8_t> rec;
	const bool want_nodes = h_neighbors || h_leaves || h_P || h_V || h_N || h_W || h_L;
	if (want_nodes) {
		rec.resize(count * NODE_BYTES);
		RK_HIP(hipMemcpyAsync(rec.data(), d.nodes + r0 * NODE_BYTES, count * NODE_BYTES, hipMemcpyDeviceToHost, st));
	}
	RK_HIP(hipStreamSynchronize(st));
	for (size_t i = 0; want_nodes && i < count; i++) {                 // unpack the records into the reference's arrays
		const uint8_t *p = rec.data() + i * NODE_BYTES;
		uint32_t expanded;
		memcpy(&expanded, p + OFF_EXPANDED, 4);
		if (h_leaves) h_leaves[i] = expanded ? 0 : 1;
		if (h_V) memcpy(h_V + i, p + OFF_V, 8);
		if (h_P) memcpy(h_P + i * 12, p + OFF_P, 96);
		if (h_W) memcpy(h_W + i * 12, p + OFF_W, 96);
		if (h_L) memcpy(h_L + i * 12, p + OFF_L, 96);
		for (int k = 0; k < 12; k++) {
			int32_t v;
			if (h_neighbors) { memcpy(&v, p + OFF_NB + 4 * k, 4); h_neighbors[i * 12 + k] = v; }
			if (h_N) { memcpy(&v, p + OFF_N + 4 * k, 4); h_N[i * 12 + k] = v; }
		}
	}
	return RK_OK;
}

long long rk_mcts_path(rk_mcts_t *h, int tree, long long *h_actions, long long *h_nodes, size_t max_len, void *stream)
{
	if (!h || !h->ready) return fail(RK_ESTATE, "rk_mcts_path: reset the engine first");
	const MctsDev &d = h->d;
	if (tree < 0 || tree >= d.T) return fail(RK_EINVAL, "rk_mcts_path: tree %d out of range", tree);
	hipStream_t st = (hipStream_t)stream;
	int32_t tr[TR_INTS];
	RK_HIP(hipMemcpyAsync(tr, d.tree + (size_t)tree * TR_INTS, sizeof tr, hipMemcpyDeviceToHost, st));
	RK_HIP(hipStreamSynchronize(st));
	const int32_t plen = tr[TR_PLEN];
	const int solved = flags_solved(tr[TR_FLAGS]);
	if (plen < 1) return fail(RK_ESTATE, "rk_mcts_path: empty path");
	std::vector<int32_t> nodes((size_t)plen);
	std::vector<uint8_t> acts((size_t)plen);
	RK_HIP(hipMemcpyAsync(nodes.data(), d.path_nodes + (size_t)tree * d.max_path, (size_t)plen * 4, hipMemcpyDeviceToHost, st));
	RK_HIP(hipMemcpyAsync(acts.data(), d.path_actions + (size_t)tree * d.max_path, (size_t)plen, hipMemcpyDeviceToHost, st));
	RK_HIP(hipStreamSynchronize(st));
	// solved during a simulation: the descent's actions plus the solving one; solved at the root: nothing; else the descent
	const size_t n_act = solved == 1 ? (size_t)plen : (solved == 2 ? 0 : (size_t)plen - 1);
	for (size_t i = 0; i < n_act && i < max_len; i++) if (h_actions) h_actions[i] = acts[i];
	for (size_t i = 0; i < (size_t)plen && i < max_len; i++) if (h_nodes) h_nodes[i] = nodes[i];
	return (long long)n_act;
}

}  // extern "C"
