// Device-side pieces shared by the search engines (single A*, sharded A*, batched A*): queue records and their order,
// the state hash, order-preserving compaction inside a workgroup, binary search in a sorted run.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "rk_device.h"

namespace rk {

struct Rec { uint64_t key; uint64_t idx; };

constexpr uint32_t TENT = 0x80000000u;          // hash slot holds a batch position, not yet an index
constexpr uint32_t NO_MARK = 0xFFFFFFFFu;


__device__ __forceinline__ bool rec_less(const Rec &a, const Rec &b)
{
	return a.key < b.key || (a.key == b.key && a.idx < b.idx);
}

// float64 -> uint64 whose unsigned order is the float order (no NaNs expected)
__device__ __forceinline__ uint64_t sortable_key(double c)
{
	c = c + 0.0;                                  // -0.0 -> +0.0: Python compares them equal
	const uint64_t u = (uint64_t)__double_as_longlong(c);
	return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

__host__ __device__ inline double key_to_double(uint64_t k)
{
	const uint64_t u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
	double d;
	memcpy(&d, &u, sizeof d);
	return d;
}

__host__ __device__ inline uint32_t hash_state(const uint32_t s[5])
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

// owner rank of a state in a hash-sharded search: a remix of the hash, so that it is independent of the table slot
__host__ __device__ inline uint32_t owner_of(const uint32_t s[5], uint32_t world)
{
	const uint32_t h = hash_state(s) * 0x9E3779B1u;
	return (uint32_t)(((uint64_t)h * world) >> 32);
}

__device__ __forceinline__ void load5(const uint32_t *p, uint32_t s[5])
{
	#pragma unroll
	for (int j = 0; j < 5; j++) s[j] = p[j];
}

__device__ __forceinline__ bool equal5(const uint32_t a[5], const uint32_t *p)
{
	return ((a[0] ^ p[0]) | (a[1] ^ p[1]) | (a[2] ^ p[2]) | (a[3] ^ p[3]) | (a[4] ^ p[4])) == 0;
}

// ---- order-preserving compaction across workgroups in ONE launch: tickets + look-back --------------------------------
// Every workgroup draws a ticket (so that "predecessor" means "started earlier": no deadlock whatever the dispatch
// order), publishes its local total at once and then sums the totals of ALL its predecessors, 64 at a time with one wave
// (a decoupled look-back without the prefix hand-over: nobody waits for a chain, only for words that are published
// before any waiting starts).  A word is {epoch : 32 | total : 32} written with ONE 64-bit agent-scope store and polled
// with agent-scope loads; the epoch makes the words of earlier launches invisible, so nothing has to be cleared except
// the ticket counter (the engine's end-of-iteration kernel does that).  The word carries its payload itself, so no
// fence is needed (MI355X_MICROARCH.md: a naturally aligned 8-byte granule written by one store).
constexpr int ASCAN = 256;                        // threads (= items) per workgroup of these compactions

// exclusive prefix of a 0/1 predicate inside a 256-thread workgroup; *total = number of set predicates
__device__ __forceinline__ int block_rank256(bool pred, int *s_wave /* [4] */, int *total)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const unsigned long long b = __ballot(pred);
	const int in_wave = __popcll(b & ((1ull << lane) - 1ull));
	__syncthreads();                                  // s_wave may still be read from a previous call
	if (lane == 0) s_wave[wv] = __popcll(b);
	__syncthreads();
	int before = 0, tot = 0;
	#pragma unroll
	for (int w = 0; w < 4; w++) {
		const int v = s_wave[w];
		before += w < wv ? v : 0;
		tot += v;
	}
	*total = tot;
	return before + in_wave;
}

__device__ __forceinline__ int scan_ticket(int32_t *ticket_ctr, int *s_ticket)
{
	if (threadIdx.x == 0) *s_ticket = atomicAdd(ticket_ctr, 1);
	__syncthreads();
	return *s_ticket;
}

// exclusive prefix of `total` over tickets 0..b-1.  Call from ALL threads of the workgroup (it contains a barrier);
// s_base is one int of LDS.
__device__ __forceinline__ int scan_lookback(unsigned long long *words, int b, int total, uint32_t epoch, int *s_base)
{
	if (threadIdx.x == 0)
		__hip_atomic_store(&words[b], ((unsigned long long)epoch << 32) | (uint32_t)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	if (threadIdx.x < 64) {                           // wave 0 sums the predecessors' totals, 64 per round
		int sum = 0;
		for (int j0 = 0; j0 < b; j0 += 64) {
			const int j = j0 + (int)threadIdx.x;
			if (j < b) {
				unsigned long long w;
				for (;;) {
					w = __hip_atomic_load(&words[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					if ((uint32_t)(w >> 32) == epoch) break;
					__builtin_amdgcn_s_sleep(1);
				}
				sum += (int)(uint32_t)w;
			}
		}
		#pragma unroll
		for (int m = 32; m > 0; m >>= 1) sum += __shfl_xor(sum, m, 64);
		if (threadIdx.x == 0) *s_base = sum;
	}
	__syncthreads();
	return *s_base;
}

// The same for W classes at once (the sharded engine buckets records by owner rank): every thread belongs to class `cls` (none: cls >= W).
// Returns the thread's position in its class's output -- the class's total over all workgroups with smaller tickets plus the thread's rank
// inside this workgroup; afterwards s_tot[w] / s_base[w] (W ints each, LDS) hold this workgroup's total and its predecessors' sum per class.
// ALL totals are published before anybody waits, so the chain costs ONE wait on the predecessors instead of W of them back to back
// (W = 8: the sharded expand kernel spent 41 us of a 14-us job in eight dependent look-backs, benchmarks/sharded_sim8.py).
// words: W rows of `nblocks` look-back words.  Call from ALL threads of a 256-thread workgroup.
__device__ __forceinline__ int scan_lookback_classes(unsigned long long *words, int nblocks, int b, int W, uint32_t cls, uint32_t epoch,
                                                    int *s_wave /* [4] */, int *s_tot, int *s_base)
{
	// ranks inside this workgroup for all classes with ONE barrier: per wave a ballot per class (no barrier), the waves' counts in LDS
	__shared__ int s_cnt[4 * 64];
	const int lane0 = threadIdx.x & 63, wv0 = threadIdx.x >> 6;
	int my = 0;
	for (int w = 0; w < W; w++) {
		const unsigned long long bal = __ballot(cls == (uint32_t)w);
		if (lane0 == 0) s_cnt[wv0 * 64 + w] = __popcll(bal);
		if (cls == (uint32_t)w) my = __popcll(bal & ((1ull << lane0) - 1ull));
	}
	__syncthreads();
	if (cls < (uint32_t)W)
		for (int v = 0; v < wv0; v++) my += s_cnt[v * 64 + cls];
	if ((int)threadIdx.x < W) s_tot[threadIdx.x] = s_cnt[threadIdx.x] + s_cnt[64 + threadIdx.x] + s_cnt[128 + threadIdx.x] + s_cnt[192 + threadIdx.x];
	(void)s_wave;
	__syncthreads();
	if ((int)threadIdx.x < W)
		__hip_atomic_store(&words[(size_t)threadIdx.x * nblocks + b], ((unsigned long long)epoch << 32) | (uint32_t)s_tot[threadIdx.x],
		                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
	for (int w = wv; w < W; w += nw) {                // a wave per class, 64 predecessors per round
		const unsigned long long *row = words + (size_t)w * nblocks;
		int sum = 0;
		for (int j0 = 0; j0 < b; j0 += 64) {
			const int j = j0 + lane;
			if (j < b) {
				unsigned long long x;
				for (;;) {
					x = __hip_atomic_load(&row[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					if ((uint32_t)(x >> 32) == epoch) break;
					__builtin_amdgcn_s_sleep(1);
				}
				sum += (int)(uint32_t)x;
			}
		}
		#pragma unroll
		for (int m = 32; m > 0; m >>= 1) sum += __shfl_xor(sum, m, 64);
		if (lane == 0) s_base[w] = sum;
	}
	__syncthreads();
	return (cls < (uint32_t)W ? s_base[cls] : 0) + my;
}

__device__ __forceinline__ int lower_bound_rec(const Rec *a, int n, const Rec &x)
{
	int lo = 0, hi = n;
	while (lo < hi) {
		const int mid = (lo + hi) >> 1;
		if (rec_less(a[mid], x)) lo = mid + 1; else hi = mid;
	}
	return lo;
}


}  // namespace rk
