// Batched weighted A*: S independent searches (reference: librubiks/solving/agents.py:171-413, one at a time there)
// advanced in lock-step by ONE set of launches per iteration, with every size that varies -- nodes popped, new states,
// queue length, won / out-of-budget -- kept in device memory.  Nothing in an iteration synchronises with the host and
// every launch has a fixed shape, so an iteration (engine kernels + the net forward on the padded (S * 12 N, 480)
// batch) can be captured in a hipGraph and replayed; the host only polls the per-search status now and then.
//
// Each search follows the reference exactly (same pop order, first-occurrence de-duplication, index numbering,
// relaxation, action queue) -- the semantics and most device functions are those of the single-search engine
// (rk_astar.hip); here every kernel has a second grid dimension (blockIdx.y = search) and reads its sizes from the
// search's counter block.
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

// per-search counters (int32 x 16)
enum {
	B_NSTATES = 0, B_OPEN = 1, B_NPOP = 2, B_NNEW = 3, B_NBEFORE = 4, B_WON = 5, B_SOLVED = 6, B_DONE = 7,
	B_BUDGET = 8, B_ITERS = 9, B_CUR = 10, B_ERROR = 11, B_STRIDE = 16
};

struct BatchDev {
	int S, N, K, Kpad, nb;                 // searches, expansions, 12 N, K rounded up to 1024, Kpad / 1024
	uint32_t cap1, tmask;
	double lambda;
	uint32_t *states; int32_t *G, *parents; uint8_t *pact; uint32_t *table, *mark;
	Rec *open0, *open1;
	int32_t *ctr;
	int32_t *exp_idx; uint32_t *par_states, *children; uint8_t *solved;
	int32_t *seen; uint32_t *child_slot; uint8_t *flags; int32_t *rank, *blk;
	uint8_t *newway, *shortcut; int32_t *val1, *val2;
	Rec *rec0, *rec1;
};

__device__ __forceinline__ int32_t *ctr_of(const BatchDev &d, int s) { return d.ctr + (size_t)s * B_STRIDE; }
__device__ __forceinline__ Rec *open_of(const BatchDev &d, int s, int which) { return (which ? d.open1 : d.open0) + (size_t)s * d.cap1; }

__global__ void kb_root(BatchDev d, const uint32_t *starts, const int32_t *budgets)
{
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= d.S) return;
	uint32_t st[5];
	load5(starts + (size_t)s * 5, st);
	const size_t base = (size_t)s * d.cap1;
	#pragma unroll
	for (int j = 0; j < 5; j++) d.states[(base + 1) * 5 + j] = st[j];
	d.G[base + 1] = 0; d.parents[base + 1] = 0; d.pact[base + 1] = 0;
	d.table[(size_t)s * (d.tmask + 1) + (hash_state(st) & d.tmask)] = 1u;
	d.open0[base] = Rec{sortable_key(0.0), 1ull};                       // heappush(open_queue, (0, 1))   agents.py:234
	int32_t *c = ctr_of(d, s);
	for (int i = 0; i < B_STRIDE; i++) c[i] = 0;
	c[B_NSTATES] = 1; c[B_OPEN] = 1; c[B_BUDGET] = budgets[s];
	c[B_DONE] = is_solved5(st) ? 1 : 0;                                 // agents.py:230
	c[B_WON] = is_solved5(st) ? 2 : 0;
}

// loop guard (agents.py:236) and the number of nodes to pop (agents.py:238)
__global__ void kb_begin(BatchDev d)
{
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= d.S) return;
	int32_t *c = ctr_of(d, s);
	c[B_NNEW] = 0;
	c[B_NBEFORE] = c[B_NSTATES];
	if (c[B_DONE]) { c[B_NPOP] = 0; return; }
	if (c[B_NSTATES] + d.K > c[B_BUDGET] || c[B_OPEN] == 0) { c[B_DONE] = 1; c[B_NPOP] = 0; return; }
	c[B_NPOP] = c[B_OPEN] < d.N ? c[B_OPEN] : d.N;
	c[B_ITERS] += 1;
}

// pop + gather; slots beyond n_pop get the solved state so that the fan-out works on valid codes everywhere
__global__ void kb_pop(BatchDev d)
{
	const int s = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= d.N * 5) return;
	const int32_t *c = ctr_of(d, s);
	const int i = t / 5, j = t - 5 * i;
	uint32_t v = SOLVED_DW[j];
	if (i < c[B_NPOP]) {
		const uint32_t idx = (uint32_t)open_of(d, s, c[B_CUR])[i].idx;
		if (j == 0) d.exp_idx[(size_t)s * d.N + i] = (int32_t)idx;
		v = d.states[((size_t)s * d.cap1 + idx) * 5 + j];
	}
	d.par_states[((size_t)s * d.N) * 5 + t] = v;
}

__global__ void kb_lookup(BatchDev d)
{
	const int s = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
	const int32_t *cs = ctr_of(d, s);
	if (c >= 12 * cs[B_NPOP]) return;
	const uint32_t *children = d.children + (size_t)s * d.K * 5;
	const uint32_t *states = d.states + (size_t)s * d.cap1 * 5;
	uint32_t *table = d.table + (size_t)s * (d.tmask + 1);
	uint32_t *mark = d.mark + (size_t)s * d.cap1;
	const size_t o = (size_t)s * d.K + c;
	uint32_t st[5];
	load5(children + (size_t)c * 5, st);
	uint32_t slot = hash_state(st) & d.tmask;
	for (;;) {
		uint32_t e = __atomic_load_n(&table[slot], __ATOMIC_RELAXED);
		if (e == 0u) {
			e = atomicCAS(&table[slot], 0u, TENT | (uint32_t)c);
			if (e == 0u) { d.seen[o] = 0; d.child_slot[o] = slot; return; }
		}
		if (e & TENT) {
			if (equal5(st, children + (size_t)(e & ~TENT) * 5)) {
				atomicMin(&table[slot], TENT | (uint32_t)c);
				d.seen[o] = 0; d.child_slot[o] = slot;
				return;
			}
		} else if (equal5(st, states + (size_t)e * 5)) {
			d.seen[o] = (int32_t)e;
			atomicMin(&mark[e], (uint32_t)c);
			return;
		}
		slot = (slot + 1) & d.tmask;
	}
}

__global__ __launch_bounds__(SCAN_BLOCK)
void kb_flags(BatchDev d)
{
	__shared__ int s_wave[16];
	const int s = blockIdx.y, c = blockIdx.x * SCAN_BLOCK + threadIdx.x;
	const int K = 12 * ctr_of(d, s)[B_NPOP];
	const size_t o = (size_t)s * d.K + c;
	int fu = 0, fs = 0;
	if (c < K) {
		const int32_t sidx = d.seen[o];
		if (sidx == 0) fu = d.table[(size_t)s * (d.tmask + 1) + d.child_slot[o]] == (TENT | (uint32_t)c);
		else fs = d.mark[(size_t)s * d.cap1 + sidx] == (uint32_t)c;
		d.flags[o] = (uint8_t)(fu | (fs << 1));
	}
	int total;
	const int r = block_rank(fu != 0, s_wave, &total);
	if (c < K) d.rank[o] = r;
	if (threadIdx.x == 0) d.blk[(size_t)s * d.nb + blockIdx.x] = total;
}

// exclusive scan of the workgroup totals of one search (at most a few dozen), n_new and n_states
__global__ void kb_scan(BatchDev d)
{
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= d.S) return;
	int32_t *c = ctr_of(d, s);
	int32_t *b = d.blk + (size_t)s * d.nb;
	const int used = (12 * c[B_NPOP] + SCAN_BLOCK - 1) / SCAN_BLOCK;
	int run = 0;
	for (int i = 0; i < used; i++) { const int v = b[i]; b[i] = run; run += v; }
	c[B_NNEW] = run;
	c[B_NSTATES] = c[B_NBEFORE] + run;
}

__global__ void kb_append(BatchDev d)
{
	const int s = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
	int32_t *cs = ctr_of(d, s);
	if (c >= 12 * cs[B_NPOP]) return;
	const size_t o = (size_t)s * d.K + c, base = (size_t)s * d.cap1;
	const uint8_t f = d.flags[o];
	const int32_t p = d.exp_idx[(size_t)s * d.N + c / 12];
	const int32_t g = d.G[base + p] + 1;
	if (f & 1) {
		const uint32_t idx = (uint32_t)cs[B_NBEFORE] + 1u + (uint32_t)(d.rank[o] + d.blk[(size_t)s * d.nb + c / SCAN_BLOCK]);
		#pragma unroll
		for (int j = 0; j < 5; j++) d.states[(base + idx) * 5 + j] = d.children[o * 5 + j];
		d.G[base + idx] = g;
		d.parents[base + idx] = p;
		d.pact[base + idx] = (uint8_t)(c % 12);
		d.table[(size_t)s * (d.tmask + 1) + d.child_slot[o]] = idx;
		if (d.solved[o]) { cs[B_WON] = 1; cs[B_SOLVED] = (int32_t)idx; }
	}
	uint8_t nw = 0;
	if (f & 2) {
		nw = g < d.G[base + d.seen[o]];
		d.val1[o] = g;
	}
	d.newway[o] = nw;
}

// one-hot of the new states of every search into the padded (S, K, 480) net batch; rows past n_new are zero
template <typename T, int ELEM_BYTES>
__global__ __launch_bounds__(256)
void kb_new_oh(BatchDev d, u32x4 *out, uint32_t one_bits)
{
	constexpr int E = 16 / ELEM_BYTES, CPR = 480 / E, CPC = 24 / E;
	const int s = blockIdx.y;
	const int32_t *cs = ctr_of(d, s);
	const int n_new = cs[B_DONE] && !cs[B_NPOP] ? 0 : cs[B_NNEW];
	const uint8_t *pool = reinterpret_cast<const uint8_t *>(d.states + ((size_t)s * d.cap1 + cs[B_NBEFORE] + 1) * 5);
	const size_t total = (size_t)d.K * CPR;
	for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (size_t)gridDim.x * blockDim.x) {
		const int r = (int)(c / CPR), g = (int)(c - (size_t)r * CPR);
		u32x4 val = {0u, 0u, 0u, 0u};
		if (r < n_new) {
			const int cubie = g / CPC, base = (g - cubie * CPC) * E;
			const int rel = (int)pool[(size_t)r * STATE_BYTES + cubie] - base;
			if (ELEM_BYTES == 4) {
				val.x = rel == 0 ? one_bits : 0u; val.y = rel == 1 ? one_bits : 0u;
				val.z = rel == 2 ? one_bits : 0u; val.w = rel == 3 ? one_bits : 0u;
			} else if (rel >= 0 && rel < 8) {
				const uint32_t one = one_bits << (16 * (rel & 1));
				val.x = (rel >> 1) == 0 ? one : 0u; val.y = (rel >> 1) == 1 ? one : 0u;
				val.z = (rel >> 1) == 2 ? one : 0u; val.w = (rel >> 1) == 3 ? one : 0u;
			}
		}
		out[((size_t)s * d.K) * CPR + c] = val;
	}
}

__global__ void kb_records(BatchDev d, const float *values)
{
	const int s = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
	const int32_t *cs = ctr_of(d, s);
	Rec *rec = d.rec0 + (size_t)s * d.Kpad;
	if (j >= d.Kpad) return;
	if (j >= cs[B_NNEW]) { rec[j] = Rec{~0ull, 0xFFFFFFFF00000000ull + (uint64_t)j}; return; }   // distinct padding, sorts to the end
	const uint32_t idx = (uint32_t)cs[B_NBEFORE] + 1u + (uint32_t)j;
	const double h = (double)(-values[(size_t)s * d.K + j]);
	const double lg = d.lambda * (double)d.G[(size_t)s * d.cap1 + idx];
	rec[j] = Rec{sortable_key(lg + h), (uint64_t)idx};
}

__global__ __launch_bounds__(512)
void kb_sort_chunks(BatchDev d)
{
	__shared__ Rec sh[1024];
	const int s = blockIdx.y, tid = threadIdx.x;
	if (ctr_of(d, s)[B_NNEW] == 0) return;                                   // uniform for the workgroup
	Rec *rec = d.rec0 + (size_t)s * d.Kpad + (size_t)blockIdx.x * 1024;
	for (int i = tid; i < 1024; i += 512) sh[i] = rec[i];
	__syncthreads();
	for (int k = 2; k <= 1024; k <<= 1)
		for (int j = k >> 1; j > 0; j >>= 1) {
			const int i = 2 * tid - (tid & (j - 1));
			const int l = i + j;
			const bool up = (i & k) == 0;
			const Rec a = sh[i], b = sh[l];
			if (rec_less(b, a) == up) { sh[i] = b; sh[l] = a; }
			__syncthreads();
		}
	for (int i = tid; i < 1024; i += 512) rec[i] = sh[i];
}

// merge neighbouring runs of length L over the whole padded array.  Padding records carry distinct maximal keys
// (kb_records), so every record of the array is distinct and finds its slot by one binary search in the partner run.
__global__ void kb_merge_pass(BatchDev d, int L, int from)
{
	const int s = blockIdx.y, e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= d.Kpad || ctr_of(d, s)[B_NNEW] == 0) return;
	const Rec *src = (from ? d.rec1 : d.rec0) + (size_t)s * d.Kpad;
	Rec *dst = (from ? d.rec0 : d.rec1) + (size_t)s * d.Kpad;
	const int r = e / L, i = e - r * L;
	const int base = (r & ~1) * L, pstart = (r ^ 1) * L;
	int plen = d.Kpad - pstart;
	plen = plen < 0 ? 0 : (plen > L ? L : plen);
	const Rec x = src[e];
	dst[base + i + lower_bound_rec(src + pstart, plen, x)] = x;
}

// push: merge what is left of the queue with the sorted new records into the search's other queue buffer
__global__ void kb_merge_queue(BatchDev d, int final_in_rec1)
{
	const int s = blockIdx.y, e = blockIdx.x * blockDim.x + threadIdx.x;
	const int32_t *cs = ctr_of(d, s);
	if (cs[B_NPOP] == 0) return;                                            // finished searches keep their queue as it is
	const int na = cs[B_OPEN] - cs[B_NPOP], nb = cs[B_NNEW];
	if (e >= na + nb) return;
	const Rec *a = open_of(d, s, cs[B_CUR]) + cs[B_NPOP];
	const Rec *b = (final_in_rec1 ? d.rec1 : d.rec0) + (size_t)s * d.Kpad;
	Rec *out = open_of(d, s, cs[B_CUR] ^ 1);
	if (e < na) {
		const Rec x = a[e];
		out[e + lower_bound_rec(b, nb, x)] = x;
	} else {
		const Rec x = b[e - na];
		out[(e - na) + lower_bound_rec(a, na, x)] = x;
	}
}

__global__ void kb_relax_1b(BatchDev d)
{
	const int s = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
	const int32_t *cs = ctr_of(d, s);
	if (c >= 12 * cs[B_NPOP] || cs[B_WON]) return;                          // the reference returns before relaxing once it has won
	const size_t o = (size_t)s * d.K + c, base = (size_t)s * d.cap1;
	if (!d.newway[o]) return;
	const int32_t t = d.seen[o];
	d.G[base + t] = d.val1[o];
	d.pact[base + t] = (uint8_t)(c % 12);
	d.parents[base + t] = d.exp_idx[(size_t)s * d.N + c / 12];
}

__global__ void kb_relax_2a(BatchDev d)
{
	const int s = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
	const int32_t *cs = ctr_of(d, s);
	if (c >= 12 * cs[B_NPOP]) return;
	const size_t o = (size_t)s * d.K + c, base = (size_t)s * d.cap1;
	uint8_t sc = 0;
	if (d.flags[o] & 2) {
		const int32_t t = d.seen[o];
		d.mark[base + t] = NO_MARK;
		if (!cs[B_WON]) {
			const int32_t g = d.G[base + t] + 1;
			sc = g < d.G[base + d.exp_idx[(size_t)s * d.N + c / 12]];
			d.val2[o] = g;
		}
	}
	d.shortcut[o] = sc;
}

__global__ void kb_relax_2b(BatchDev d)
{
	const int s = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
	const int32_t *cs = ctr_of(d, s);
	if (i >= cs[B_NPOP] || cs[B_WON]) return;
	const size_t base = (size_t)s * d.cap1;
	const int32_t p = d.exp_idx[(size_t)s * d.N + i];
	for (int a = 0; a < 12; a++) {
		const size_t o = (size_t)s * d.K + 12 * i + a;
		if (d.shortcut[o]) {
			d.G[base + p] = d.val2[o];
			d.pact[base + p] = (uint8_t)(a ^ 1);
			d.parents[base + p] = d.seen[o];
		}
	}
}

__global__ void kb_end(BatchDev d, int merge_width)
{
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= d.S) return;
	int32_t *c = ctr_of(d, s);
	if (c[B_NPOP] == 0) return;
	c[B_OPEN] = c[B_OPEN] - c[B_NPOP] + c[B_NNEW];
	if (c[B_OPEN] > merge_width) { c[B_ERROR] = 1; c[B_DONE] = 1; }         // the host's queue-length bound was too small: the merge was cut short
	c[B_CUR] ^= 1;
	if (c[B_WON]) c[B_DONE] = 1;
}

// action indices from the root to node `index` of search s, walked on the device (agents.py:244-251)
__global__ void kb_walk(BatchDev d, int s, int index, int32_t *out /* [0] = length or -1, then actions root -> node */, int max_len)
{
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	const size_t base = (size_t)s * d.cap1;
	int len = 0, i = index;
	while (i != 1 && len <= (int)d.cap1) { i = d.parents[base + i]; len++; if (i < 1 || (uint32_t)i >= d.cap1) { out[0] = -1; return; } }
	if (i != 1) { out[0] = -1; return; }
	out[0] = len;
	i = index;
	for (int k = len - 1; k >= 0; k--) {
		if (k < max_len) out[1 + k] = d.pact[base + i];
		i = d.parents[base + i];
	}
}

}  // namespace rk

using namespace rk;

struct rk_astarb {
	BatchDev d{};
	size_t capacity = 0;
	std::vector<void *> allocs;
	uint32_t *starts = nullptr;
	int32_t *budgets = nullptr;
	int32_t *walk = nullptr;      // path walk result: length, then actions
	int merge_bound = 0;          // launch width of the queue merge: an upper bound of any search's queue length
	bool ready = false, pending = false;
};

namespace {

template <typename T>
int b_alloc(rk_astarb *h, T **p, size_t count)
{
	void *q = nullptr;
	RK_HIP(hipMalloc(&q, count * sizeof(T) + 64));
	h->allocs.push_back(q);
	*p = static_cast<T *>(q);
	return RK_OK;
}

inline unsigned nblk(size_t n, unsigned per = 256) { return (unsigned)((n + per - 1) / per); }

constexpr int WALK_MAX = 1 << 16;

}  // namespace

extern "C" {

int rk_astarb_create(rk_astarb_t **out, int n_searches, size_t capacity_per_search, int max_expansions)
{
	if (!out) return fail(RK_EINVAL, "rk_astarb_create: null out pointer");
	if (n_searches < 1 || n_searches > 65535) return fail(RK_EINVAL, "rk_astarb_create: n_searches %d out of range", n_searches);
	if (max_expansions < 1 || max_expansions > (1 << 20)) return fail(RK_EINVAL, "rk_astarb_create: max_expansions %d out of range", max_expansions);
	if (capacity_per_search < 12 * (size_t)max_expansions + 2 || capacity_per_search > 0x3FFFFFF0ull)
		return fail(RK_EINVAL, "rk_astarb_create: capacity %zu out of range (needs at least 12 * expansions + 2)", capacity_per_search);
	rk_astarb *h = new rk_astarb();
	h->capacity = capacity_per_search;
	BatchDev &d = h->d;
	d.S = n_searches; d.N = max_expansions; d.K = 12 * max_expansions;
	d.Kpad = ((d.K + 1023) / 1024) * 1024;
	d.nb = d.Kpad / 1024;
	d.cap1 = (uint32_t)(capacity_per_search + 1);
	uint64_t t = 1024;
	while (t < 2ull * d.cap1 + 2) t <<= 1;
	d.tmask = (uint32_t)(t - 1);
	const size_t S = (size_t)d.S, rows = S * d.cap1, SK = S * d.K;
	int e = RK_OK;
	#define A(ptr, cnt) if (!e) e = b_alloc(h, &d.ptr, (cnt))
	A(states, rows * 5); A(G, rows); A(parents, rows); A(pact, rows); A(table, S * (size_t)t); A(mark, rows);
	A(open0, rows); A(open1, rows); A(ctr, S * B_STRIDE);
	A(exp_idx, S * d.N); A(par_states, S * d.N * 5 + 64); A(children, SK * 5 + 64); A(solved, SK + 64);
	A(seen, SK); A(child_slot, SK); A(flags, SK); A(rank, SK); A(blk, S * d.nb + 16);
	A(newway, SK); A(shortcut, SK); A(val1, SK); A(val2, SK);
	A(rec0, S * (size_t)d.Kpad); A(rec1, S * (size_t)d.Kpad);
	#undef A
	if (!e) e = b_alloc(h, &h->starts, S * 5);
	if (!e) e = b_alloc(h, &h->budgets, S);
	if (!e) e = b_alloc(h, &h->walk, (size_t)WALK_MAX + 8);
	if (e) { rk_astarb_destroy(h); return e; }
	*out = h;
	return RK_OK;
}

int rk_astarb_destroy(rk_astarb_t *h)
{
	if (!h) return RK_OK;
	for (void *p : h->allocs) (void)hipFree(p);
	delete h;
	return RK_OK;
}

int rk_astarb_reset(rk_astarb_t *h, const int8_t *h_start_states, const long long *h_max_states, double lambda, void *stream)
{
	if (!h || !h_start_states) return fail(RK_EINVAL, "rk_astarb_reset: null argument");
	hipStream_t st = (hipStream_t)stream;
	BatchDev &d = h->d;
	const size_t S = (size_t)d.S, rows = S * d.cap1;
	d.lambda = lambda;
	std::vector<int32_t> b(S);
	for (size_t s = 0; s < S; s++) {
		long long m = h_max_states ? h_max_states[s] : (long long)h->capacity;
		if (m > (long long)h->capacity) m = (long long)h->capacity;
		b[s] = (int32_t)(m < 0 ? 0 : m);
	}
	RK_HIP(hipMemsetAsync(d.table, 0, S * ((size_t)d.tmask + 1) * sizeof(uint32_t), st));
	RK_HIP(hipMemsetAsync(d.mark, 0xFF, rows * sizeof(uint32_t), st));
	RK_HIP(hipMemcpyAsync(h->starts, h_start_states, S * STATE_BYTES, hipMemcpyHostToDevice, st));
	RK_HIP(hipMemcpyAsync(h->budgets, b.data(), S * sizeof(int32_t), hipMemcpyHostToDevice, st));
	hipLaunchKernelGGL(kb_root, dim3(nblk(S)), dim3(256), 0, st, d, h->starts, h->budgets);
	RK_HIP(hipGetLastError());
	RK_HIP(hipStreamSynchronize(st));
	h->merge_bound = 1 + d.K;
	h->ready = true;
	h->pending = false;
	return RK_OK;
}

int rk_astarb_set_merge_bound(rk_astarb_t *h, long long bound)
{
	if (!h) return fail(RK_EINVAL, "rk_astarb_set_merge_bound: null handle");
	if (bound < 1) bound = 1;
	if (bound > (long long)h->capacity) bound = (long long)h->capacity;
	h->merge_bound = (int)bound;
	return RK_OK;
}

int rk_astarb_step_expand(rk_astarb_t *h, void *d_onehot, int out_dtype, void *stream)
{
	if (!h || !h->ready) return fail(RK_ESTATE, "rk_astarb_step_expand: reset the engine first");
	if (h->pending) return fail(RK_ESTATE, "rk_astarb_step_expand: previous step not committed");
	if (!d_onehot || (reinterpret_cast<uintptr_t>(d_onehot) & 15)) return fail(RK_EINVAL, "rk_astarb_step_expand: one-hot buffer must be 16-byte aligned");
	if (out_dtype < RK_OH_F32 || out_dtype > RK_OH_BF16) return fail(RK_EINVAL, "rk_astarb_step_expand: unknown dtype %d", out_dtype);
	hipStream_t st = (hipStream_t)stream;
	const BatchDev &d = h->d;
	const dim3 gK(nblk(d.K), d.S), gS(nblk(d.S));
	hipLaunchKernelGGL(kb_begin, gS, dim3(256), 0, st, d);
	hipLaunchKernelGGL(kb_pop, dim3(nblk((size_t)d.N * 5), d.S), dim3(256), 0, st, d);
	launch_expand12((const int8_t *)d.par_states, (int8_t *)d.children, d.solved, nullptr, (size_t)d.S * d.N, st);
	hipLaunchKernelGGL(kb_lookup, gK, dim3(256), 0, st, d);
	hipLaunchKernelGGL(kb_flags, dim3(d.nb, d.S), dim3(SCAN_BLOCK), 0, st, d);
	hipLaunchKernelGGL(kb_scan, gS, dim3(256), 0, st, d);
	hipLaunchKernelGGL(kb_append, gK, dim3(256), 0, st, d);
	const unsigned ohg = nblk((size_t)d.K * (out_dtype == RK_OH_F32 ? 120 : 60), 256);
	if (out_dtype == RK_OH_F32) hipLaunchKernelGGL((kb_new_oh<float, 4>), dim3(ohg, d.S), dim3(256), 0, st, d, (u32x4 *)d_onehot, 0x3F800000u);
	else if (out_dtype == RK_OH_F16) hipLaunchKernelGGL((kb_new_oh<_Float16, 2>), dim3(ohg, d.S), dim3(256), 0, st, d, (u32x4 *)d_onehot, 0x3C00u);
	else hipLaunchKernelGGL((kb_new_oh<_Float16, 2>), dim3(ohg, d.S), dim3(256), 0, st, d, (u32x4 *)d_onehot, 0x3F80u);
	RK_HIP(hipGetLastError());
	h->pending = true;
	return RK_OK;
}

int rk_astarb_step_commit(rk_astarb_t *h, const float *d_values, void *stream)
{
	if (!h || !h->pending) return fail(RK_ESTATE, "rk_astarb_step_commit: no pending step");
	if (!d_values) return fail(RK_EINVAL, "rk_astarb_step_commit: null values");
	hipStream_t st = (hipStream_t)stream;
	const BatchDev &d = h->d;
	const dim3 gK(nblk(d.K), d.S), gS(nblk(d.S));
	hipLaunchKernelGGL(kb_records, dim3(nblk(d.Kpad), d.S), dim3(256), 0, st, d, d_values);
	hipLaunchKernelGGL(kb_sort_chunks, dim3(d.nb, d.S), dim3(512), 0, st, d);
	int from = 0;
	for (int L = 1024; L < d.Kpad; L <<= 1) {
		hipLaunchKernelGGL(kb_merge_pass, dim3(nblk(d.Kpad), d.S), dim3(256), 0, st, d, L, from);
		from ^= 1;
	}
	hipLaunchKernelGGL(kb_merge_queue, dim3(nblk((size_t)h->merge_bound + d.K), d.S), dim3(256), 0, st, d, from);
	hipLaunchKernelGGL(kb_relax_1b, gK, dim3(256), 0, st, d);
	hipLaunchKernelGGL(kb_relax_2a, gK, dim3(256), 0, st, d);
	hipLaunchKernelGGL(kb_relax_2b, dim3(nblk(d.N), d.S), dim3(256), 0, st, d);
	hipLaunchKernelGGL(kb_end, gS, dim3(256), 0, st, d, h->merge_bound + d.K);
	RK_HIP(hipGetLastError());
	h->pending = false;
	return RK_OK;
}

int rk_astarb_status(rk_astarb_t *h, long long *h_status, void *stream)
{
	if (!h || !h->ready || !h_status) return fail(RK_EINVAL, "rk_astarb_status: bad argument");
	hipStream_t st = (hipStream_t)stream;
	const size_t S = (size_t)h->d.S;
	std::vector<int32_t> c(S * B_STRIDE);
	RK_HIP(hipMemcpyAsync(c.data(), h->d.ctr, S * B_STRIDE * sizeof(int32_t), hipMemcpyDeviceToHost, st));
	RK_HIP(hipStreamSynchronize(st));
	for (size_t s = 0; s < S; s++) {
		const int32_t *r = c.data() + s * B_STRIDE;
		long long *o = h_status + 7 * s;
		o[0] = r[B_DONE]; o[1] = r[B_WON]; o[2] = r[B_NSTATES]; o[3] = r[B_ITERS]; o[4] = r[B_OPEN]; o[5] = r[B_SOLVED]; o[6] = r[B_ERROR];
	}
	return RK_OK;
}

int rk_astarb_export(rk_astarb_t *h, int search, size_t first, size_t count, int8_t *h_states, double *h_G, long long *h_parents,
                     long long *h_parent_actions, void *stream)
{
	if (!h || !h->ready) return fail(RK_ESTATE, "rk_astarb_export: reset the engine first");
	const BatchDev &d = h->d;
	if (search < 0 || search >= d.S) return fail(RK_EINVAL, "rk_astarb_export: search %d out of range", search);
	if (first + count > d.cap1) return fail(RK_EINVAL, "rk_astarb_export: rows outside the pool");
	if (count == 0) return RK_OK;
	hipStream_t st = (hipStream_t)stream;
	const size_t r0 = (size_t)search * d.cap1 + first;
	std::vector<int32_t> g, p;
	std::vector<uint8_t> a;
	if (h_states) RK_HIP(hipMemcpyAsync(h_states, d.states + r0 * 5, count * STATE_BYTES, hipMemcpyDeviceToHost, st));
	if (h_G) { g.resize(count); RK_HIP(hipMemcpyAsync(g.data(), d.G + r0, count * 4, hipMemcpyDeviceToHost, st)); }
	if (h_parents) { p.resize(count); RK_HIP(hipMemcpyAsync(p.data(), d.parents + r0, count * 4, hipMemcpyDeviceToHost, st)); }
	if (h_parent_actions) { a.resize(count); RK_HIP(hipMemcpyAsync(a.data(), d.pact + r0, count, hipMemcpyDeviceToHost, st)); }
	RK_HIP(hipStreamSynchronize(st));
	for (size_t i = 0; i < count; i++) {
		if (h_G) h_G[i] = (double)g[i];
		if (h_parents) h_parents[i] = p[i];
		if (h_parent_actions) h_parent_actions[i] = a[i];
	}
	return RK_OK;
}

long long rk_astarb_path(rk_astarb_t *h, int search, long long index, long long *h_actions, size_t max_len, void *stream)
{
	if (!h || !h->ready) return fail(RK_ESTATE, "rk_astarb_path: reset the engine first");
	const BatchDev &d = h->d;
	if (search < 0 || search >= d.S) return fail(RK_EINVAL, "rk_astarb_path: search %d out of range", search);
	if (index < 1 || (size_t)index >= d.cap1) return fail(RK_EINVAL, "rk_astarb_path: index %lld out of range", index);
	hipStream_t st = (hipStream_t)stream;
	hipLaunchKernelGGL(kb_walk, dim3(1), dim3(64), 0, st, d, search, (int)index, h->walk, WALK_MAX);
	RK_HIP(hipGetLastError());
	int32_t len = 0;
	RK_HIP(hipMemcpyAsync(&len, h->walk, sizeof len, hipMemcpyDeviceToHost, st));
	RK_HIP(hipStreamSynchronize(st));
	if (len < 0) return fail(RK_ESTATE, "rk_astarb_path: broken parent chain");
	size_t n = (size_t)len < max_len ? (size_t)len : max_len;
	if (n > (size_t)WALK_MAX) n = WALK_MAX;
	std::vector<int32_t> acts(n);
	if (n) {
		RK_HIP(hipMemcpyAsync(acts.data(), h->walk + 1, n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
		RK_HIP(hipStreamSynchronize(st));
	}
	for (size_t k = 0; k < n; k++) h_actions[k] = acts[k];
	return (long long)len;
}

}  // extern "C"
