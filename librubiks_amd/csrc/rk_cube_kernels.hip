// Cube hot-path kernels for MI355X (gfx950): 12-child fan-out with fused goal test, per-state moves, goal test,
// move sequences (scramblers), one-hot encoding; 20-byte and 6x8x6 representations.
//
// All of this is HBM-bound byte work (no contraction, so no MFMA).  The shape every kernel follows:
//   global --16 B/lane coalesced--> LDS --transposed, conflict-free--> registers (one state per lane)
//   registers --v_perm_b32 table look-ups / byte transposes--> LDS --16 B/lane coalesced--> global
// The move tables (576..768 B) and the solved state live in the constant segment and are staged in LDS once per
// workgroup.  LDS staging areas are private to a wave, so the streaming loops contain no s_barrier.
//
// The large launches of the write-heavy kernels (fan-out, as_oh, 6x8x6 fan-out, multi_rotate) run in a PACED form (round 3,
// DESIGN 3 steps 3-4; "expand12, paced form" below has the reasons): the inputs of a phase are read first, into the Infinity
// Cache, and every tile's stores are then held until the tile's slot on a fixed-rate schedule of the constant 100 MHz clock,
// because HBM takes an ordered, rate-limited store stream 25 % faster than the same bytes from thousands of independent waves
// and reads mixed into it cost 2.5 x their share.  Only WHEN a finished tile is stored depends on any of that (RK_PACE=0: never).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "rk_device.h"
#include "rk_kernels.h"

namespace rk {

// ================================================================================================================
// expand12: n parents -> 12 n children (+ solved flags)                          reference: agents.py:277-281,
//                                                                                cube.py:256-263, cube.py:88-89
// Algorithmic HBM bytes per parent: 20 read + 240 + 12 written = 272.
//
// A wave owns a tile of 64 parents (one per lane): coalesced loads bring the tile's 1 280 B in, the wave re-reads them
// one state per lane through LDS (stride 5 dwords: conflict-free).  Then each lane
//   1. fetches, for each of its 20 cubies, the 16-byte row rows[kind][code] = that cubie's code in all 12
//      children (20 ds_read_b128; the table is 768 B, codes differing by 16 share a bank: at most 2-way),
//   2. turns the 20x12 byte matrix into 12 children x 5 dwords with 15 4x4 byte transposes (120 v_perm_b32),
//   3. tests every child against the solved dwords,
//   4. writes its 240 B to the wave's staging area (15 ds_write_b128, lane stride 60 dwords: conflict-free);
//      the wave then streams the 15 360 B out as fifteen 1 KiB global_store_dwordx4.
// ================================================================================================================
constexpr int EXP_ROUND = 64;                  // parents per round (one per lane)
constexpr int EXP_WAVES = 4;                   // waves per workgroup of the shipping shape
constexpr unsigned PACE_TAU_PS = 2100;                  // store schedule of the paced fan-out: 2.10 ns per 64-parent tile (16 128 B: 7.68 TB/s, 96 % of the
                                                        // HBM peak; at 2.05 ns no box keeps the schedule: 0.79 of peak instead of 0.83-0.84, profiles/r03_paced_tau.json)
constexpr unsigned PACE_ROT_TAU_PS = 1200;              // multi_rotate from 2 Mi states on: 1.2 ns per 256-state tile and non-temporal stores, 0.71 -> 0.76 of peak
                                                        // at 12 M rows (1.0-1.3 ns alike, 0.75 at 1.4-1.5, 0.70 at 1.7; non-temporal stores unpaced: 0.61), profiles/r03_rows_pace.json
constexpr unsigned PACE_SOLVED_TAU_PS = 0;              // multi_is_solved: unpaced -- with its flag stream 0.72-0.73 at every tau (a pure read stream gains 9 %)
constexpr unsigned PACE_LEAD_TICKS = 50;                // tile 0's slot: 0.5 us after the read phase has ended
constexpr unsigned PACE_PULL_WGS = 128;                 // workgroups of a phase that read its parents (a quarter of the resident workgroups)
constexpr unsigned PACE_PULL_WGS_FIRST = 256;           // ... of the first phase
constexpr unsigned PACE_PHASE_TILES = 16384;            // tiles per phase: 1 Mi parents, 20 MiB of parents in the Infinity Cache at a time
constexpr size_t PACE_MIN_TILES = 3072;                 // launches below 196 608 parents keep the unpaced forms (equal at 100 k, +3 % at 250 k)
constexpr unsigned PACE686_MIN_PARENTS = 65536;          // 6x8x6 fan-out: paced from 65 536 parents on
constexpr unsigned long long PACE_MAX_WAIT_TICKS = 2000; // 20 us
constexpr unsigned long long PACE_STALE_TICKS = 1500;   // a slot more than 15 us before the wave's own start belongs to an older launch
constexpr size_t PACE_FIRST_TILES = 16384;              // tiles whose waves may start before the time base is this launch's (2 x the most waves a launch has resident)
constexpr int EXP_GRID_PERSISTENT = 3072;      // workgroups of the persistent (input-pipelined) launch

template <int HALVES>
struct ExpandWaveLdsT {
	u32x4    stage[EXP_ROUND * 15 / HALVES];   // 15 360 B (or half): children of one round; the head doubles as input staging
	uint32_t flags[EXP_ROUND * 3];             // 768 B: 12 solved bytes per parent
};


// ================================================================================================================
// expand12, ring form (round 3).  Same tile work as k_expand12<., 1, true, 4>, restructured around what the ISA of that
// kernel showed: (i) its predicated loads and stores compile to one exec-branch per access and `s_waitcnt vmcnt(0)` in
// front of every use -- a wave drains all sixteen stores of a tile before it may touch the parents of the next one --;
// (ii) only ONE tile's parents are ever in flight per wave.  Here the main loop runs over FULL tiles only (every load
// and store unconditional, so the compiler counts vmcnt exactly and a wave never waits for its own stores), the parents
// of the next DEPTH tiles of the wave are in flight in registers (a ring, the loop unrolled DEPTH times so that the
// ring index is static), and the ragged last tile takes a separate predicated path after the loop.
// ================================================================================================================
// 16-byte store with a cache policy.  SP 0 plain, 1 non-temporal (`nt`; what ships); the others exist for the tuning build's
// A/B (benchmarks/tune_expand.py): 2 `sc1`, 3 `sc0 sc1` (write-through: the line leaves the XCD's L2 at once), 4 `sc1 nt`.
// They go through a raw buffer store so that the compiler still counts them in vmcnt (an asm store would not be counted).
template <int SP>
__device__ __forceinline__ void store16(u32x4 *base /* the same in every lane */, int idx /* 16-byte words, per lane */, const u32x4 v)
{
	if (SP == 0) base[idx] = v;
	else if (SP == 1) __builtin_nontemporal_store(v, base + idx);
	else {
		const uint64_t b = reinterpret_cast<uint64_t>(base);               // into SGPRs: a buffer descriptor is scalar
		const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
		const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uint64_t)hi << 32) | lo), 0, 1 << 20, 0x00020000);
		constexpr int aux = SP == 2 ? 16 : SP == 3 ? 17 : 18;              // cache-policy bits: sc0 = 1, nt = 2, sc1 = 16
		__builtin_amdgcn_raw_buffer_store_b128(v, rsrc, idx * 16, 0, aux);
	}
}

struct ExpandCtx {
	const u32x4 *rows;            // LDS: 48 rows of the per-cubie table
	u32x4       *stage;           // LDS: this wave's 15 360-byte staging area (head doubles as input staging)
	uint32_t    *flags;           // LDS: this wave's 768 flag bytes
	int          lane;
};

// the twelve children of the state held in par[5] (one state per lane) -> out[a*5 + j], solved bytes -> fl[3]
template <bool WITH_FLAGS>
__device__ __forceinline__ void expand_lane(const ExpandCtx &c, const uint32_t par[5], uint32_t out[60], uint32_t fl[3])
{
	#pragma unroll
	for (int j = 0; j < 5; j++) {
		const uint32_t x = par[j];
		const int kind_base = (j < 2) ? 0 : 24;
		const u32x4 r0 = c.rows[kind_base + (x & 0xFF)];
		const u32x4 r1 = c.rows[kind_base + ((x >> 8) & 0xFF)];
		const u32x4 r2 = c.rows[kind_base + ((x >> 16) & 0xFF)];
		const u32x4 r3 = c.rows[kind_base + (x >> 24)];
		transpose4x4(r0.x, r1.x, r2.x, r3.x, out[0 * 5 + j], out[1 * 5 + j], out[2 * 5 + j], out[3 * 5 + j]);
		transpose4x4(r0.y, r1.y, r2.y, r3.y, out[4 * 5 + j], out[5 * 5 + j], out[6 * 5 + j], out[7 * 5 + j]);
		transpose4x4(r0.z, r1.z, r2.z, r3.z, out[8 * 5 + j], out[9 * 5 + j], out[10 * 5 + j], out[11 * 5 + j]);
	}
	fl[0] = fl[1] = fl[2] = 0u;
	if (WITH_FLAGS) {
		#pragma unroll
		for (int a = 0; a < 12; a++)
			if (is_solved5(&out[a * 5])) fl[a >> 2] |= 1u << (8 * (a & 3));
	}
}

template <bool WITH_FLAGS>
__device__ __forceinline__ void report_solved(const uint32_t fl[3], bool lane_valid, size_t first_child, long long *stats)
{
	// solved children are rare: one ballot decides whether anybody reports
	const bool any = (fl[0] | fl[1] | fl[2]) != 0u && lane_valid;
	if (WITH_FLAGS && stats != nullptr && __ballot(any) != 0ull && any) {
		const int cnt = __popc(fl[0]) + __popc(fl[1]) + __popc(fl[2]);
		int first = 0;
		#pragma unroll
		for (int a = 11; a >= 0; a--)
			if (fl[a >> 2] & (1u << (8 * (a & 3)))) first = a;
		atomicAdd(reinterpret_cast<unsigned long long *>(&stats[0]), (unsigned long long)cnt);
		atomicMin(&stats[1], (long long)(first_child + first));
	}
}

// One FULL tile: raw[k] = dword k*64+lane of the tile's 1 280 bytes (already in registers).  No predicate anywhere.
struct NoHold { __device__ __forceinline__ void operator()() const {} };

template <bool WITH_FLAGS, int NT, class Hold = NoHold>
__device__ __forceinline__ void expand_full_tile(const ExpandCtx &c, const uint32_t raw[5], size_t p0, u32x4 *__restrict__ children,
                                                 uint32_t *__restrict__ solved, long long *__restrict__ stats, Hold hold = Hold())
{
	uint32_t *stage_dw = reinterpret_cast<uint32_t *>(c.stage);
	#pragma unroll
	for (int k = 0; k < 5; k++) stage_dw[k * 64 + c.lane] = raw[k];
	wave_lds_fence();
	uint32_t par[5];
	#pragma unroll
	for (int j = 0; j < 5; j++) par[j] = stage_dw[c.lane * 5 + j];
	wave_lds_fence();
	uint32_t out[60], fl[3];
	expand_lane<WITH_FLAGS>(c, par, out, fl);
	if (WITH_FLAGS) {
		c.flags[c.lane * 3 + 0] = fl[0];
		c.flags[c.lane * 3 + 1] = fl[1];
		c.flags[c.lane * 3 + 2] = fl[2];
	}
	#pragma unroll
	for (int v = 0; v < 15; v++)
		c.stage[c.lane * 15 + v] = u32x4{out[4 * v], out[4 * v + 1], out[4 * v + 2], out[4 * v + 3]};
	wave_lds_fence();
	hold();                                                // the paced form waits here for its slot; the tile is staged, nothing is stored yet
	u32x4 *dst = children + p0 * 15;
	#pragma unroll
	for (int v = 0; v < 15; v++) {
		const u32x4 val = c.stage[v * 64 + c.lane];
		store16<NT>(dst, v * 64 + c.lane, val);
	}
	if (WITH_FLAGS) {
		uint32_t *fdst = solved + p0 * 3;                  // p0 is a multiple of 64 -> 768-byte blocks, 16-byte aligned if the base is
		if ((reinterpret_cast<uintptr_t>(fdst) & 15) == 0) {
			if (c.lane < 48) {
				const u32x4 val = reinterpret_cast<const u32x4 *>(c.flags)[c.lane];
				store16<NT>(reinterpret_cast<u32x4 *>(fdst), c.lane, val);
			}
		} else {
			#pragma unroll
			for (int k = 0; k < 3; k++) fdst[k * 64 + c.lane] = c.flags[k * 64 + c.lane];
		}
		report_solved<WITH_FLAGS>(fl, true, (p0 + c.lane) * 12, stats);
	}
	wave_lds_fence();
}

// The ragged last tile (np < 64 parents): predicated, not pipelined.
template <bool WITH_FLAGS, int NT>
__device__ __forceinline__ void expand_ragged_tile(const ExpandCtx &c, const uint32_t *__restrict__ parents, size_t p0, int np,
                                                   u32x4 *__restrict__ children, uint32_t *__restrict__ solved, long long *__restrict__ stats)
{
	uint32_t *stage_dw = reinterpret_cast<uint32_t *>(c.stage);
	const uint32_t *src = parents + p0 * STATE_DWORDS;
	const int ndw = np * STATE_DWORDS;
	#pragma unroll
	for (int k = 0; k < 5; k++) {
		const int idx = k * 64 + c.lane;
		stage_dw[idx] = src[idx < ndw ? idx : ndw - 1];    // clamped: always a valid address, lanes past np are never stored
	}
	wave_lds_fence();
	uint32_t par[5];
	#pragma unroll
	for (int j = 0; j < 5; j++) par[j] = stage_dw[c.lane * 5 + j];
	wave_lds_fence();
	uint32_t out[60], fl[3];
	expand_lane<WITH_FLAGS>(c, par, out, fl);
	if (WITH_FLAGS) {
		c.flags[c.lane * 3 + 0] = fl[0];
		c.flags[c.lane * 3 + 1] = fl[1];
		c.flags[c.lane * 3 + 2] = fl[2];
	}
	#pragma unroll
	for (int v = 0; v < 15; v++)
		c.stage[c.lane * 15 + v] = u32x4{out[4 * v], out[4 * v + 1], out[4 * v + 2], out[4 * v + 3]};
	wave_lds_fence();
	u32x4 *dst = children + p0 * 15;
	const int nvec = np * 15;
	#pragma unroll
	for (int v = 0; v < 15; v++) {
		const int idx = v * 64 + c.lane;
		if (idx < nvec) {
			const u32x4 val = c.stage[idx];
			store16<NT>(dst, idx, val);
		}
	}
	if (WITH_FLAGS) {
		uint32_t *fdst = solved + p0 * 3;
		#pragma unroll
		for (int k = 0; k < 3; k++) {
			const int idx = k * 64 + c.lane;
			if (idx < np * 3) fdst[idx] = c.flags[idx];
		}
		report_solved<WITH_FLAGS>(fl, c.lane < np, (p0 + c.lane) * 12, stats);
	}
	wave_lds_fence();
}

// DEPTH = tiles of parents a wave keeps in flight (0: load, wait, expand -- for batches with one tile per wave).
// NTL = non-temporal parent loads.
// PULL > 0: the first PULL workgroups of the grid do not expand anything: they read the parent array once, front to back, as a
// dense read-only stream (16 x 1 KiB in flight per wave) and throw the words away.  They are dispatched first, so the stream
// runs in the first microseconds of the launch -- while the expanding workgroups are still waiting for their first tile and
// no child has been written, i.e. while the HBM bus would be idle -- and it leaves the parents in the Infinity Cache: the
// tile loads of every later workgroup are cache hits instead of HBM reads mixed into the write stream (the regime the
// counters of DESIGN 3 show to cost 4-9 % at 1 M parents).  At most PULL_MAX_BYTES are pulled (what the cache holds with room
// to spare); the expanding workgroups are blockIdx.x - PULL of gridDim.x - PULL.
constexpr size_t PULL_MAX_BYTES = (size_t)96 << 20;

__device__ __forceinline__ void pull_parents(const uint32_t *__restrict__ parents, size_t n, size_t wave, size_t n_waves, int lane)
{
	size_t bytes = n * STATE_BYTES;
	bytes = bytes < PULL_MAX_BYTES ? bytes : PULL_MAX_BYTES;
	if ((reinterpret_cast<uintptr_t>(parents) & 15) != 0 || bytes < 1024) return;
	const u32x4 *src16 = reinterpret_cast<const u32x4 *>(parents);
	const size_t last16 = bytes / 16 - 1, n_kib = (bytes + 1023) / 1024;
	size_t c = wave;
	u32x4 a[8], b[8];
	#define RK_ISSUE(r) _Pragma("unroll") for (int j = 0; j < 8; j++) { const size_t i = c * 64 + lane; r[j] = src16[i < last16 ? i : last16]; c += n_waves; }
	#define RK_EAT(r) _Pragma("unroll") for (int j = 0; j < 8; j++) asm volatile("" :: "v"(r[j].x), "v"(r[j].y), "v"(r[j].z), "v"(r[j].w) : "memory");
	RK_ISSUE(a)
	while (c < n_kib) {
		RK_ISSUE(b)
		RK_EAT(a)
		RK_ISSUE(a)
		RK_EAT(b)
	}
	RK_EAT(a)
	#undef RK_ISSUE
	#undef RK_EAT
}

// The read phase of the paced form: `n_waves` waves read the parents once, 1 KiB pieces, wave w takes w, w + n_waves, ...; up to
// twenty-four pieces in flight per wave.  A piece past the end stands in as the wave's own first piece (a line it already holds;
// clamping them all to the array's last word would send every wave to one L2 channel).
__device__ __forceinline__ void pull_front_bytes(const void *__restrict__ parents, size_t bytes, size_t wave, size_t n_waves, int lane)
{
	bytes = bytes < PULL_MAX_BYTES ? bytes : PULL_MAX_BYTES;
	if ((reinterpret_cast<uintptr_t>(parents) & 15) != 0 || bytes < 1024) return;
	const u32x4 *src16 = reinterpret_cast<const u32x4 *>(parents);
	const size_t last16 = bytes / 16 - 1, n_kib = (bytes + 1023) / 1024;
	const size_t own = wave * 64 + lane < last16 ? wave * 64 + lane : last16;
	// 24 pieces in flight per wave: with 256 reading workgroups a phase's 20 MiB are requested in ONE round (20 pieces per wave)
	// instead of three rounds of 16 + 16 + 6 with 128 of them, each a full memory latency
	for (size_t c = wave; c < n_kib; c += 24 * n_waves) {
		u32x4 a[8], b[8], d[8];
		#define RK_ISSUE(r, o) _Pragma("unroll") for (int j = 0; j < 8; j++) { const size_t pc = c + ((o) + j) * n_waves, i = pc * 64 + lane; r[j] = src16[pc < n_kib ? (i < last16 ? i : last16) : own]; }
		#define RK_EAT(r) _Pragma("unroll") for (int j = 0; j < 8; j++) asm volatile("" :: "v"(r[j].x), "v"(r[j].y), "v"(r[j].z), "v"(r[j].w) : "memory");
		RK_ISSUE(a, 0) RK_ISSUE(b, 8) RK_ISSUE(d, 16)
		RK_EAT(a) RK_EAT(b) RK_EAT(d)
		#undef RK_ISSUE
		#undef RK_EAT
	}
}
__device__ __forceinline__ void pull_front(const uint32_t *__restrict__ parents, size_t n, size_t wave, size_t n_waves, int lane)
{
	pull_front_bytes(parents, n * STATE_BYTES, wave, n_waves, lane);
}

template <bool WITH_FLAGS, int DEPTH, int NT = 1, bool NTL = false, int PULL = 0>
__global__ __launch_bounds__(EXP_WAVES * WAVE)
void k_expand12r(const uint32_t *__restrict__ parents, u32x4 *__restrict__ children, uint32_t *__restrict__ solved,
                 long long *__restrict__ stats, size_t n)
{
	__shared__ u32x4 s_rows[48];
	__shared__ ExpandWaveLdsT<1> s_wave[EXP_WAVES];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	if (PULL > 0 && blockIdx.x < PULL) {                                 // the whole workgroup: no barrier has been reached yet
		pull_parents(parents, n, (size_t)blockIdx.x * EXP_WAVES + wv, (size_t)PULL * EXP_WAVES, lane);
		return;
	}
	const size_t n_full = n / EXP_ROUND;                               // tiles without a predicate
	const size_t stride = (size_t)(gridDim.x - PULL) * EXP_WAVES;
	const size_t first = (size_t)(blockIdx.x - PULL) * EXP_WAVES + wv;
	constexpr int D = DEPTH > 0 ? DEPTH : 1;
	uint32_t ring[D][5];

	auto load_tile = [&](uint32_t (&dst)[5], size_t t) {
		const uint32_t *src = parents + t * (EXP_ROUND * STATE_DWORDS) + lane;
		#pragma unroll
		for (int k = 0; k < 5; k++) dst[k] = NTL ? __builtin_nontemporal_load(src + k * 64) : src[k * 64];
	};

	// the first DEPTH tiles' parents are requested before the table is staged
	#pragma unroll
	for (int d = 0; d < DEPTH; d++)
		if (first + d * stride < n_full) load_tile(ring[d], first + d * stride);

	if (tid < 48) {
		const uint32_t *src = reinterpret_cast<const uint32_t *>(D_TAB.rows) + 4 * tid;
		s_rows[tid] = u32x4{src[0], src[1], src[2], src[3]};
	}
	__syncthreads();
	const ExpandCtx c{s_rows, s_wave[wv].stage, s_wave[wv].flags, lane};

	if (DEPTH == 0) {
		for (size_t t = first; t < n_full; t += stride) {
			load_tile(ring[0], t);
			expand_full_tile<WITH_FLAGS, NT>(c, ring[0], t * EXP_ROUND, children, solved, stats);
		}
	} else {
		for (size_t base = first; base < n_full; base += (size_t)DEPTH * stride) {
			#pragma unroll
			for (int d = 0; d < DEPTH; d++) {
				const size_t t = base + d * stride;
				if (t >= n_full) break;                                   // wave-uniform
				uint32_t raw[5];
				#pragma unroll
				for (int k = 0; k < 5; k++) raw[k] = ring[d][k];
				const size_t nx = t + (size_t)DEPTH * stride;
				if (nx < n_full) load_tile(ring[d], nx);                  // refill the slot: lands DEPTH tiles from now
				expand_full_tile<WITH_FLAGS, NT>(c, raw, t * EXP_ROUND, children, solved, stats);
			}
		}
	}
	// ragged tail: the wave that would own tile n_full
	if ((n % EXP_ROUND) != 0 && (n_full % stride) == first)
		expand_ragged_tile<WITH_FLAGS, NT>(c, parents, n_full * EXP_ROUND, (int)(n % EXP_ROUND), children, solved, stats);
}

// ================================================================================================================
// expand12, paced form (round 3, for launches of many tiles).  What the store-stream diagnostics of benchmarks/tune_expand.py
// showed (profiles/r03_store_stream.json): the same 240 MB of children written as 15 KiB per wave reach 5.6 TB/s when every
// wave stores the moment it can and 7.1 TB/s when the waves release their tiles ONE AFTER THE OTHER AT A FIXED RATE just
// below what HBM absorbs -- the stores in flight then always cover one narrow, advancing address window instead of thousands
// of scattered 15 KiB pieces, and the queues in front of the memory channels never fill.  And: 20 MB of parent reads mixed
// into that stream cost 10 us of 45, however the stores are ordered; read FIRST (into the Infinity Cache, which non-temporal
// stores leave alone) they cost 4.
//   * the first `pull_wgs` workgroups of a phase read its parents once, front to back, and leave; the last of them move the
//     phase's time base (the launch's cell of g_pace_cells) up to "now" (atomic max by one wave in 256: the base ends where the read phase ended);
//   * every other wave owns one tile: it loads its parents (cache hits now), expands and stages them exactly as k_expand12r
//     does, then HOLDS the sixteen stores until base + lead + tile x tau on the constant-rate clock (s_memrealtime, 10 ns).
//     A wave behind its slot stores at once; nothing moves the schedule (a first version let late waves push the base: waits
//     then add up, and 2 048 atomics on one address held the whole launch for 50 us).  A schedule nobody can keep therefore
//     degenerates to the unpaced kernel plus the lead.
// Results never depend on any of this -- only the moment at which a finished tile is stored does.
// ================================================================================================================
// Time bases of the paced launches.  Every paced launch gets ITS OWN cell (a kernel argument: the host hands out the cells of this
// ring in turn, next_pace_cell), so two paced launches in flight on two streams -- fan-out beside as_oh, a search's fan-out
// beside an exchange -- never move each other's schedule (round 3 had one global word for all kernels on all streams: results
// never depended on it, but each phase's readers moved the base under the other launch's waves, up to the 20 us cap per
// residency).  A cell is 128 bytes apart from the next, so the atomics of two launches do not share a line.  A hipGraph replays
// a launch with the cell it was captured with: it then finds the base of its own previous replay, far in the past, which the
// hold's stale test handles exactly as it handled round 3's global word.
constexpr int PACE_CELLS = 64, PACE_CELL_STRIDE = 16;
__device__ unsigned long long g_pace_cells[PACE_CELLS * PACE_CELL_STRIDE];

#ifdef RK_TUNING
__device__ unsigned long long *g_pace_dbg;         // tuning build: per tile {base read, time at the hold, due time, time after the hold}
#endif

struct PaceHold {
	unsigned long long base, start; size_t tile; unsigned tau_ps, lead; int lane; const unsigned long long *cell;
	__device__ __forceinline__ void operator()() const
	{
		if (tau_ps == 0) return;
		const unsigned long long slot = lead + (unsigned long long)tile * tau_ps / 10000ull;
		unsigned long long b = base;
		asm volatile("" ::: "memory");
		// A wave of the first residency may have started before this launch's read phase ended: the base it read is the previous
		// launch's then, its slot lies far in the past.  It re-reads until the base is this launch's -- at most eight times
		// (~15 us; a read phase lasts 4-8).  The test cannot tell "another phase's base" from "this launch is more than 15 us
		// behind its schedule" (two processes sharing the GPU): every wave then polls before it goes, which is why the bound is
		// tight.  Three sharper tests were tried -- a phase tag riding in the low bits of the base, first-residency waves only,
		// the first reader publishing an estimate of the read phase's end -- and each lost 1-4 % at 2-32 M parents or more in the
		// 6x8x6 fan-out on the boxes they were measured on; this one is what the records of profiles/ were taken with.
		if (tile < PACE_FIRST_TILES)
			for (int i = 0; i < 8 && b + slot + PACE_STALE_TICKS < start; i++) {
				__builtin_amdgcn_s_sleep(16);
				b = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		const unsigned long long due = b + slot;
		const unsigned long long now = __builtin_amdgcn_s_memrealtime();
#ifdef RK_TUNING
		struct Dbg { unsigned long long *p; size_t t; int l; __device__ ~Dbg() { if (p && l == 0) p[t * 4 + 3] = __builtin_amdgcn_s_memrealtime(); } } dbg{g_pace_dbg, tile, lane};
		if (g_pace_dbg && lane == 0) { g_pace_dbg[tile * 4] = b; g_pace_dbg[tile * 4 + 1] = now; g_pace_dbg[tile * 4 + 2] = due; }
#endif
		// behind its slot a wave simply goes (the schedule is never moved: a wave that waits is the only cost pacing can have,
		// and it is bounded by the lead); ahead of it, it waits -- never more than PACE_MAX_WAIT_TICKS, whatever the base says
		// (a wave is dispatched at most one residency, ~5 us, before its slot; a base moved by a concurrent launch on another
		//  stream can therefore cost a launch 20 us, not more)
		if (now < due && due - now < PACE_MAX_WAIT_TICKS)
			while (__builtin_amdgcn_s_memrealtime() < due) __builtin_amdgcn_s_sleep(1);
		asm volatile("" ::: "memory");
	}
};

// The grid is a sequence of PHASES of `phase_tiles` tiles (a multiple of EXP_WAVES; what the Infinity Cache holds of parents with
// room to spare): `pull_wgs` reading workgroups, then the phase's expanding workgroups, then the next phase's readers, ...
// Workgroups are dispatched in that order, so a phase's readers start while the tail of the previous phase is still being
// written and its expanders follow them.  Every phase has its own schedule: tile index and time base start again.
template <bool WITH_FLAGS>
__global__ __launch_bounds__(EXP_WAVES * WAVE)
void k_expand12p(const uint32_t *__restrict__ parents, u32x4 *__restrict__ children, uint32_t *__restrict__ solved,
                 long long *__restrict__ stats, size_t n, unsigned pull_wgs, unsigned pull_extra, unsigned phase_tiles, unsigned tau_ps, unsigned lead,
                 unsigned long long *pace)
{
	__shared__ u32x4 s_rows[48];
	__shared__ ExpandWaveLdsT<1> s_wave[EXP_WAVES];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	// the first phase has `pull_extra` more reading workgroups than the others (nothing else runs yet: 256 of them request the
	// phase's 20 MiB in one round; a later phase's readers share the chip with the tail of the phase before and 128 do better)
	const unsigned wgs_per_phase = pull_wgs + phase_tiles / EXP_WAVES;
	unsigned phase = 0, r = blockIdx.x, pullers = pull_wgs + pull_extra;             // r: index inside the phase, readers first
	if (blockIdx.x >= pull_extra + wgs_per_phase) {
		const unsigned b = blockIdx.x - pull_extra;
		phase = b / wgs_per_phase; r = b - phase * wgs_per_phase; pullers = pull_wgs;
	}
	const size_t p_first = (size_t)phase * phase_tiles * EXP_ROUND;                  // first parent of the phase
	if (r < pullers) {                                                  // read phase; the whole workgroup leaves before any barrier
		const size_t p_count = n - p_first < (size_t)phase_tiles * EXP_ROUND ? n - p_first : (size_t)phase_tiles * EXP_ROUND;
		pull_front(parents + p_first * STATE_DWORDS, p_count, (size_t)r * EXP_WAVES + wv, (size_t)pullers * EXP_WAVES, lane);
		// the read phase ends when its last workgroups end; one wave in 256 publishes the time (atomics on one address complete
		// one every ~25 ns chip-wide and hold the memory pipeline of the waves behind them)
		if (lane == 0 && wv == 0 && ((r & 63) == 63 || r + 1 == pullers))
			__hip_atomic_fetch_max(pace, (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		return;
	}
	const unsigned long long start = __builtin_amdgcn_s_memrealtime();
	const size_t n_full = n / EXP_ROUND;
	const size_t t_in_phase = (size_t)(r - pullers) * EXP_WAVES + wv;
	const size_t t = (size_t)phase * phase_tiles + t_in_phase;
	if (pullers == 0 && t_in_phase == 0 && lane == 0)                   // no read phase: the time base is the start of the phase's first wave
		__hip_atomic_fetch_max(pace, start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	// requested now, used after the tile is staged: the base of the schedule and the tile's parents
	const unsigned long long base = __hip_atomic_load(pace, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	uint32_t raw[5] = {0, 0, 0, 0, 0};
	if (t < n_full) {
		const uint32_t *src = parents + t * (EXP_ROUND * STATE_DWORDS) + lane;
		#pragma unroll
		for (int k = 0; k < 5; k++) raw[k] = src[k * 64];
	}
	if (tid < 48) {
		const uint32_t *src = reinterpret_cast<const uint32_t *>(D_TAB.rows) + 4 * tid;
		s_rows[tid] = u32x4{src[0], src[1], src[2], src[3]};
	}
	__syncthreads();
	const ExpandCtx c{s_rows, s_wave[wv].stage, s_wave[wv].flags, lane};
	if (t < n_full)
		expand_full_tile<WITH_FLAGS, 1>(c, raw, t * EXP_ROUND, children, solved, stats, PaceHold{base, start, t_in_phase, tau_ps, lead, lane, pace});
	else if (t == n_full && (n % EXP_ROUND) != 0)
		expand_ragged_tile<WITH_FLAGS, 1>(c, parents, n_full * EXP_ROUND, (int)(n % EXP_ROUND), children, solved, stats);
}

// ================================================================================================================
// expand12, structure-of-arrays form.  Parents are five dword planes P[j][n] (plane j = bytes 4j..4j+3 of every state),
// children sixty planes C[a][j][n] (action-major) and twelve flag planes F[a][n].  Lane i of a wave handles parent
// p0+i, so every load and every one of the 60 stores of a wave is one contiguous 256-byte access: no LDS transpose
// at all (LDS only holds the 768-byte row table).  Same arithmetic as k_expand12; the layout differs from the
// reference's (12 n, 20) array, so this form is for device-resident pipelines, not for the drop-in surface.
// ================================================================================================================
template <bool WITH_FLAGS>
__global__ __launch_bounds__(256)
void k_expand12_soa(const uint32_t *__restrict__ parents, uint32_t *__restrict__ children, uint8_t *__restrict__ solved,
                    long long *__restrict__ stats, size_t n)
{
	__shared__ u32x4 s_rows[48];
	const int tid = threadIdx.x;
	if (tid < 48) {
		const uint32_t *src = reinterpret_cast<const uint32_t *>(D_TAB.rows) + 4 * tid;
		s_rows[tid] = u32x4{src[0], src[1], src[2], src[3]};
	}
	__syncthreads();
	for (size_t p = (size_t)blockIdx.x * blockDim.x + tid; p < n; p += (size_t)gridDim.x * blockDim.x) {
		uint32_t out[60];
		#pragma unroll
		for (int j = 0; j < 5; j++) {
			const uint32_t x = parents[(size_t)j * n + p];
			const int kind_base = (j < 2) ? 0 : 24;
			const u32x4 r0 = s_rows[kind_base + (x & 0xFF)];
			const u32x4 r1 = s_rows[kind_base + ((x >> 8) & 0xFF)];
			const u32x4 r2 = s_rows[kind_base + ((x >> 16) & 0xFF)];
			const u32x4 r3 = s_rows[kind_base + (x >> 24)];
			transpose4x4(r0.x, r1.x, r2.x, r3.x, out[0 * 5 + j], out[1 * 5 + j], out[2 * 5 + j], out[3 * 5 + j]);
			transpose4x4(r0.y, r1.y, r2.y, r3.y, out[4 * 5 + j], out[5 * 5 + j], out[6 * 5 + j], out[7 * 5 + j]);
			transpose4x4(r0.z, r1.z, r2.z, r3.z, out[8 * 5 + j], out[9 * 5 + j], out[10 * 5 + j], out[11 * 5 + j]);
		}
		#pragma unroll
		for (int d = 0; d < 60; d++) __builtin_nontemporal_store(out[d], children + (size_t)d * n + p);
		if (WITH_FLAGS) {
			int first = -1, cnt = 0;
			#pragma unroll
			for (int a = 11; a >= 0; a--) {
				const bool ok = is_solved5(&out[a * 5]);
				solved[(size_t)a * n + p] = ok ? 1 : 0;
				if (ok) { first = a; cnt++; }
			}
			if (stats != nullptr && __ballot(cnt != 0) != 0ull && cnt != 0) {
				atomicAdd(reinterpret_cast<unsigned long long *>(&stats[0]), (unsigned long long)cnt);
				atomicMin(&stats[1], (long long)(p * 12 + first));
			}
		}
	}
}

// AoS (n, 20) <-> five dword planes
__global__ __launch_bounds__(256)
void k_states_to_soa(const uint32_t *__restrict__ states, uint32_t *__restrict__ planes, size_t n, int to_soa)
{
	// one thread per dword; consecutive threads walk the AoS array, so the AoS side is coalesced and the plane side is
	// a stride-5 gather/scatter inside a 1 280-byte window (served by L1/L2)
	const size_t total = n * 5;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		const size_t p = i / 5;
		const int j = (int)(i - p * 5);
		if (to_soa) planes[(size_t)j * n + p] = states[i];
		else const_cast<uint32_t *>(states)[i] = planes[(size_t)j * n + p];
	}
}

// ================================================================================================================
// multi_rotate: out[i] = move actions[i] on states[i]                                       cube.py:256-263
// Algorithmic bytes per state: 20 + 1 read, 20 written.
// A wave owns 256 states (5 KiB in, 5 KiB out, all as 1 KiB dwordx4 accesses); lane l handles states 4l..4l+3 of
// the tile, i.e. 80 contiguous LDS bytes = five conflict-free ds_read_b128.  The per-lane action selects a 48-byte
// table (three ds_read_b128 from the 576-byte LDS copy), and each dword of the state is re-coded by lut4().
// ================================================================================================================
constexpr int ROW_TILE = 256;     // states per wave tile
constexpr int ROW_WAVES = 4;

// Device-pointer entries take action codes as they are.  A code >= 12 is treated as action 0 (the kernels never index past the
// move table) AND leaves a mark: g_bad_actions becomes non-zero and stays so until rk_bad_actions_seen() reads and clears it, so
// that a caller who passes garbage can find out without paying a reduction and a synchronisation on every call.
__device__ unsigned g_bad_actions;
__device__ __forceinline__ void note_bad_action(bool bad)
{
	if (__ballot(bad) != 0ull && bad) atomicOr(&g_bad_actions, 1u);      // never taken on valid input
}

// Reads, too, go faster in order and at a fixed rate (profiles/r03_store_stream.json, ids 310-312: a pure read stream of 5 KiB per
// wave runs at 6.27 TB/s when every wave loads the moment it starts and at 6.85 TB/s when wave w loads at t0 + w x 0.70 ns; at
// 0.66 ns it is back at 6.3).  A paced per-row launch has one tile per wave; the wave of tile 0 sets the time base (there is no read
// phase here), and every wave waits for its slot BEFORE it requests its states.
__device__ __forceinline__ void row_pace_hold(size_t tile, unsigned tau_ps, unsigned lead, int lane, unsigned long long *pace)
{
	if (tau_ps == 0) return;
	const unsigned long long start = __builtin_amdgcn_s_memrealtime();
	if (tile == 0 && lane == 0) __hip_atomic_fetch_max(pace, start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	const unsigned long long base = __hip_atomic_load(pace, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	PaceHold{base, start, tile, tau_ps, lead, lane, pace}();
}

// WITH_FLAGS: the goal test of the MOVED states in the same launch (the pair every per-row caller runs back to back:
// agents.py:157-159, :696-703; train.py:277-281) -- flags (one byte per state, nullable) and the count / first index in
// `stats` as k_multi_is_solved reports them; 21 B read + 21 B written per state instead of 41 + 21 in two launches.
template <bool SPLIT_FD, bool WITH_FLAGS = false>
__global__ __launch_bounds__(ROW_WAVES * WAVE)
void k_multi_rotate(const uint32_t *__restrict__ states, const uint8_t *__restrict__ act_or_faces,
                    const uint8_t *__restrict__ dirs, uint32_t *__restrict__ out, size_t n, size_t n_tiles,
                    unsigned tau_ps, unsigned lead, unsigned nt_stores, unsigned long long *pace, uint8_t *__restrict__ flags = nullptr,
                    long long *__restrict__ stats = nullptr)
{
	__shared__ u32x4 s_act[36];
	__shared__ u32x4 s_buf[ROW_WAVES][320];       // 5 120 B per wave

	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const size_t first_tile = (size_t)blockIdx.x * ROW_WAVES + wv;
	row_pace_hold(first_tile, tau_ps, lead, lane, pace);    // paced launches (one tile per wave): the wave's LOADS wait for its slot
	// the first tile's states are requested before the move tables are staged (a wave of the usual one-tile grid would
	// otherwise wait for the table's round trip and only then start its own) ...
	// ... and on a persistent grid every further tile's states are requested while the previous tile is moved and stored.
	u32x4 pre[5];
	const size_t tile_stride = (size_t)gridDim.x * ROW_WAVES;
	auto request = [&](size_t t) {                                       // whole, 16-byte aligned tiles only; wave-uniform answer
		const bool ok = t < n_tiles && n - t * ROW_TILE >= (size_t)ROW_TILE && ((reinterpret_cast<uintptr_t>(states + t * ROW_TILE * STATE_DWORDS) & 15) == 0);
		if (ok) {
			const u32x4 *src4 = reinterpret_cast<const u32x4 *>(states + t * ROW_TILE * STATE_DWORDS);
			#pragma unroll
			for (int k = 0; k < 5; k++) pre[k] = src4[k * 64 + lane];
		}
		return ok;
	};
	bool have_pre = request(first_tile);
	stage_action_tables(s_act, tid);
	__syncthreads();

	u32x4 *buf = s_buf[wv];
	uint32_t *buf_dw = reinterpret_cast<uint32_t *>(buf);

	for (size_t tile = first_tile; tile < n_tiles; tile += tile_stride) {
		const size_t p0 = tile * ROW_TILE;
		const int np = (int)((n - p0 < (size_t)ROW_TILE) ? (n - p0) : (size_t)ROW_TILE);
		const uint32_t *src = states + p0 * STATE_DWORDS;
		uint32_t *dst = out + p0 * STATE_DWORDS;
		const bool full = np == ROW_TILE && (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0);

		if (have_pre) {
			#pragma unroll
			for (int k = 0; k < 5; k++) buf[k * 64 + lane] = pre[k];
			have_pre = request(tile + tile_stride);
		} else if (full) {
			const u32x4 *src4 = reinterpret_cast<const u32x4 *>(src);
			#pragma unroll
			for (int k = 0; k < 5; k++) buf[k * 64 + lane] = src4[k * 64 + lane];
		} else {
			// ragged or misaligned tile: dword loads from CLAMPED addresses -- always valid, so all twenty are issued back to back and
			// waited for once.  As predicated loads (`idx < ndw ? src[idx] : 0`) they compiled to one exec branch and one
			// `s_waitcnt vmcnt(0)` each: twenty dependent round trips, 3.5 us on top of every launch whose row count is not a
			// multiple of 256 (multi_is_solved on 250 000 rows: 6.8 us against 3.4 us on 261 120, profiles/r04_rows_ragged.json)
			const int ndw = np * STATE_DWORDS;
			uint32_t got[20];
			#pragma unroll
			for (int k = 0; k < 20; k++) {
				const int idx = k * 64 + lane;
				got[k] = src[idx < ndw ? idx : ndw - 1];
			}
			#pragma unroll
			for (int k = 0; k < 20; k++) {
				const int idx = k * 64 + lane;
				buf_dw[idx] = idx < ndw ? got[k] : 0u;
			}
		}
		// the four actions of this lane's states: one dword per lane when the tile is whole and the codes are aligned
		uint32_t act[4];
		if (!SPLIT_FD && np == ROW_TILE && ((reinterpret_cast<uintptr_t>(act_or_faces + p0) & 3) == 0)) {
			const uint32_t w = reinterpret_cast<const uint32_t *>(act_or_faces + p0)[lane];
			#pragma unroll
			for (int q = 0; q < 4; q++) {
				const uint32_t a = (w >> (8 * q)) & 0xFFu;
				act[q] = a < 12u ? a : 0u;
			}
			note_bad_action(((w & (w << 1)) & 0x08080808u) != 0u || (w & 0xF0F0F0F0u) != 0u);                // any byte >= 12: bits 3 and 2, or a high nibble
		} else {
			bool bad_any = false;
			uint32_t raw[4];
			#pragma unroll
			for (int q = 0; q < 4; q++) {                 // clamped addresses: four loads in flight, not four round trips (see above)
				const int r = 4 * lane + q;
				const size_t i = p0 + (size_t)(r < np ? r : np - 1);
				raw[q] = SPLIT_FD ? (2u * act_or_faces[i] + (1u - dirs[i])) : act_or_faces[i];
			}
			#pragma unroll
			for (int q = 0; q < 4; q++) {
				const uint32_t a = 4 * lane + q < np ? raw[q] : 0u;
				bad_any |= a >= 12u;
				act[q] = a < 12u ? a : 0u;            // out-of-range actions are rejected on the host; never index past the table
			}
			note_bad_action(bad_any);
		}
		wave_lds_fence();

		u32x4 v[5];
		#pragma unroll
		for (int k = 0; k < 5; k++) v[k] = buf[lane * 5 + k];
		uint32_t s[20] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w, v[2].x, v[2].y, v[2].z, v[2].w,
		                  v[3].x, v[3].y, v[3].z, v[3].w, v[4].x, v[4].y, v[4].z, v[4].w};
		#pragma unroll
		for (int q = 0; q < 4; q++) {
			uint32_t tab[12];
			load_action_table(s_act, act[q], tab);
			move5(&s[5 * q], tab);
		}
		if (WITH_FLAGS) {                                                  // cube.py:88-89 on the moved states
			uint32_t fl = 0;
			#pragma unroll
			for (int q = 0; q < 4; q++)
				if (4 * lane + q < np && is_solved5(&s[5 * q])) fl |= 1u << (8 * q);
			if (flags != nullptr) {
				uint8_t *fdst = flags + p0;
				if (np == ROW_TILE && ((reinterpret_cast<uintptr_t>(fdst) & 3) == 0)) {
					reinterpret_cast<uint32_t *>(fdst)[lane] = fl;
				} else {
					#pragma unroll
					for (int q = 0; q < 4; q++)
						if (4 * lane + q < np) fdst[4 * lane + q] = (uint8_t)((fl >> (8 * q)) & 1u);
				}
			}
			if (stats != nullptr && __ballot(fl != 0u) != 0ull && fl != 0u) {
				atomicAdd(reinterpret_cast<unsigned long long *>(&stats[0]), (unsigned long long)__popc(fl));
				atomicMin(&stats[1], (long long)(p0 + 4 * lane + ((__ffs(fl) - 1) >> 3)));
			}
		}
		#pragma unroll
		for (int k = 0; k < 5; k++) buf[lane * 5 + k] = u32x4{s[4 * k], s[4 * k + 1], s[4 * k + 2], s[4 * k + 3]};
		wave_lds_fence();

		if (full) {
			u32x4 *dst4 = reinterpret_cast<u32x4 *>(dst);
			if (nt_stores) {
				#pragma unroll
				for (int k = 0; k < 5; k++) __builtin_nontemporal_store(buf[k * 64 + lane], dst4 + k * 64 + lane);
			} else {
				#pragma unroll
				for (int k = 0; k < 5; k++) dst4[k * 64 + lane] = buf[k * 64 + lane];
			}
		} else {
			const int ndw = np * STATE_DWORDS;
			#pragma unroll
			for (int k = 0; k < 20; k++) {
				const int idx = k * 64 + lane;
				if (idx < ndw) dst[idx] = buf_dw[idx];
			}
		}
		wave_lds_fence();
	}
}

// ================================================================================================================
// multi_is_solved: flags[i] = states[i] == solved                                               cube.py:88-89
// Algorithmic bytes per state: 20 read + 1 written.  Same tile shape as multi_rotate; the four flag bytes of a lane
// form one dword, so flags leave as one coalesced 256-byte store per tile.  Counting and "first solved index" use a
// wave ballot so that only waves that saw a solved state touch the atomics.
// ================================================================================================================
__global__ __launch_bounds__(ROW_WAVES * WAVE)
void k_multi_is_solved(const uint32_t *__restrict__ states, uint8_t *__restrict__ flags, long long *__restrict__ stats,
                       size_t n, size_t n_tiles, unsigned tau_ps, unsigned lead, unsigned long long *pace)
{
	__shared__ u32x4 s_buf[ROW_WAVES][320];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	row_pace_hold((size_t)blockIdx.x * ROW_WAVES + wv, tau_ps, lead, lane, pace);
	u32x4 *buf = s_buf[wv];
	uint32_t *buf_dw = reinterpret_cast<uint32_t *>(buf);
	// on a persistent grid the next tile's states are requested while the current tile is tested (whole, aligned tiles)
	u32x4 pre[5];
	const size_t first_tile = (size_t)blockIdx.x * ROW_WAVES + wv, tile_stride = (size_t)gridDim.x * ROW_WAVES;
	auto request = [&](size_t t) {
		const bool ok = t < n_tiles && n - t * ROW_TILE >= (size_t)ROW_TILE && ((reinterpret_cast<uintptr_t>(states + t * ROW_TILE * STATE_DWORDS) & 15) == 0);
		if (ok) {
			const u32x4 *src4 = reinterpret_cast<const u32x4 *>(states + t * ROW_TILE * STATE_DWORDS);
			#pragma unroll
			for (int k = 0; k < 5; k++) pre[k] = src4[k * 64 + lane];
		}
		return ok;
	};
	bool have_pre = request(first_tile);

	for (size_t tile = first_tile; tile < n_tiles; tile += tile_stride) {
		const size_t p0 = tile * ROW_TILE;
		const int np = (int)((n - p0 < (size_t)ROW_TILE) ? (n - p0) : (size_t)ROW_TILE);
		const uint32_t *src = states + p0 * STATE_DWORDS;
		if (have_pre) {
			#pragma unroll
			for (int k = 0; k < 5; k++) buf[k * 64 + lane] = pre[k];
			have_pre = request(tile + tile_stride);
		} else if (np == ROW_TILE && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
			const u32x4 *src4 = reinterpret_cast<const u32x4 *>(src);
			#pragma unroll
			for (int k = 0; k < 5; k++) buf[k * 64 + lane] = src4[k * 64 + lane];
		} else {
			const int ndw = np * STATE_DWORDS;                               // clamped addresses, as in k_multi_rotate's ragged path
			uint32_t got[20];
			#pragma unroll
			for (int k = 0; k < 20; k++) {
				const int idx = k * 64 + lane;
				got[k] = src[idx < ndw ? idx : ndw - 1];
			}
			#pragma unroll
			for (int k = 0; k < 20; k++) {
				const int idx = k * 64 + lane;
				buf_dw[idx] = idx < ndw ? got[k] : 0xFFFFFFFFu;
			}
		}
		wave_lds_fence();
		u32x4 v[5];
		#pragma unroll
		for (int k = 0; k < 5; k++) v[k] = buf[lane * 5 + k];
		const uint32_t s[20] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w, v[2].x, v[2].y, v[2].z, v[2].w,
		                        v[3].x, v[3].y, v[3].z, v[3].w, v[4].x, v[4].y, v[4].z, v[4].w};
		uint32_t fl = 0;
		#pragma unroll
		for (int q = 0; q < 4; q++)
			if (4 * lane + q < np && is_solved5(&s[5 * q])) fl |= 1u << (8 * q);
		if (flags != nullptr) {
			uint8_t *fdst = flags + p0;
			if (np == ROW_TILE && ((reinterpret_cast<uintptr_t>(fdst) & 3) == 0)) {
				reinterpret_cast<uint32_t *>(fdst)[lane] = fl;
			} else {
				#pragma unroll
				for (int q = 0; q < 4; q++)
					if (4 * lane + q < np) fdst[4 * lane + q] = (uint8_t)((fl >> (8 * q)) & 1u);
			}
		}
		if (stats != nullptr && __ballot(fl != 0u) != 0ull && fl != 0u) {
			atomicAdd(reinterpret_cast<unsigned long long *>(&stats[0]), (unsigned long long)__popc(fl));
			const int firstq = (__ffs(fl) - 1) >> 3;
			atomicMin(&stats[1], (long long)(p0 + 4 * lane + firstq));
		}
		wave_lds_fence();
	}
}

// ================================================================================================================
// apply_sequences: every game starts solved and applies its column of a (depth, games) action matrix
//                                                                              cube.py:206-216, cube.py:218-232
// One lane per game; the state never leaves registers.  Reads of the action matrix are coalesced across games.
// ================================================================================================================
// The moves d0 ... d0 + CH - 1 of game g (those below `todo`), action bytes first: the CH loads do not depend on each other, so they
// are issued back to back and cost ONE memory latency -- as `a = actions[d]; move; a = actions[d + 1]; ...` every move waited for
// its own byte (rocprofv3, 7 500 games x 30 moves: k_apply_sequences 19.1 us, 0.6 us per move; the walk inside k_rollout_fanout 20 us).
template <int CH>
__device__ __forceinline__ void fetch_actions(const uint8_t *__restrict__ actions, size_t games, size_t g, int d0, int todo, uint32_t (&a)[CH])
{
	#pragma unroll
	for (int k = 0; k < CH; k++) a[k] = d0 + k < todo ? (uint32_t)actions[(size_t)(d0 + k) * games + g] : 0xFFu;
}

__global__ __launch_bounds__(256)
void k_apply_sequences(const uint8_t *__restrict__ actions, int moves, int games, int with_solved, int only_last,
                       uint32_t *__restrict__ out)
{
	__shared__ u32x4 s_act[36];
	stage_action_tables(s_act, threadIdx.x);
	__syncthreads();
	const int g = blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= games) return;
	uint32_t s[5] = {SOLVED_DW[0], SOLVED_DW[1], SOLVED_DW[2], SOLVED_DW[3], SOLVED_DW[4]};
	const int rows = moves + (with_solved ? 1 : 0);
	uint32_t *o = out + (size_t)g * (only_last ? 1 : rows) * STATE_DWORDS;
	if (!only_last && with_solved) {
		#pragma unroll
		for (int j = 0; j < 5; j++) o[j] = s[j];
		o += STATE_DWORDS;
	}
	uint32_t worst = 0;
	constexpr int CH = 16;
	for (int d0 = 0; d0 < moves; d0 += CH) {
		uint32_t acts[CH];
		fetch_actions<CH>(actions, (size_t)games, (size_t)g, d0, moves, acts);
		#pragma unroll
		for (int k = 0; k < CH; k++) {
			if (d0 + k < moves) {                                          // (no break: the loop unrolls, the loads above stay batched)
				uint32_t a = acts[k];
				worst = a > worst ? a : worst;
				a = a < 12u ? a : 0u;
				uint32_t tab[12];
				load_action_table(s_act, a, tab);
				move5(s, tab);
				if (!only_last) {
					#pragma unroll
					for (int j = 0; j < 5; j++) o[j] = s[j];
					o += STATE_DWORDS;
				}
			}
		}
	}
	if (only_last) {
		#pragma unroll
		for (int j = 0; j < 5; j++) o[j] = s[j];
	}
	if (worst >= 12u) atomicOr(&g_bad_actions, 1u);                      // never taken on valid input
}

// A move is a permutation of the 24 corner codes and of the 24 edge codes: a 48-byte table, twelve dwords.  Moves COMPOSE --
// (B after A)[v] = B[A[v]], four codes per lut4 -- so the composition of consecutive moves is an inclusive prefix scan over lanes that hold
// one move each: after the scan lane l holds the composition of the moves of lanes l0 ... l, l0 = the first lane of its segment (`seg` =
// the lane's position inside its segment of length `len`: a game's rows, or a 64-move chunk of one long game).  log2(len) steps of twelve
// cross-lane dwords and twelve lut4 each, all in registers.  All 64 lanes call it (ds_bpermute).
__device__ __forceinline__ void scan_moves(uint32_t (&X)[12], int lane, int seg, int len)
{
	for (int off = 1; off < len; off <<= 1) {
		uint32_t Y[12];
		const int src = (lane - off) & 63;
		#pragma unroll
		for (int j = 0; j < 12; j++) Y[j] = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)X[j]);
		if (seg >= off) {                                                // the lane `off` back belongs to the SAME segment
			uint32_t Z[12];
			#pragma unroll
			for (int j = 0; j < 6; j++) { Z[j] = lut4(Y[j], X); Z[6 + j] = lut4(Y[6 + j], X + 6); }   // the earlier moves first, then mine
			#pragma unroll
			for (int j = 0; j < 12; j++) X[j] = Z[j];
		}
	}
}
__device__ __forceinline__ void identity_moves(uint32_t (&X)[12])
{
	#pragma unroll
	for (int j = 0; j < 6; j++) X[j] = X[6 + j] = 0x03020100u + 0x04040404u * (uint32_t)j;
}

// The scramblers for FEW games (one `scramble(depth)` of the evaluation loop, depth 100-999; `sequence_scrambler` of a few thousand
// games): a WAVE per game, lane l holds move d0 + l of a 64-move chunk, the chunk's states come out of one scan_moves, the last one
// carries into the next chunk.  A game of 999 moves is 16 chunks of six scan steps instead of 999 dependent moves of one lane
// (k_apply_sequences: that loop is right when there are enough games to fill the chip with lanes -- bench.py's 1 M walks).
__global__ __launch_bounds__(256)
void k_apply_sequences_scan(const uint8_t *__restrict__ actions, int moves, int games, int with_solved, int only_last, uint32_t *__restrict__ out)
{
	__shared__ u32x4 s_act[36];
	stage_action_tables(s_act, threadIdx.x);
	__syncthreads();
	const int lane = threadIdx.x & 63, g = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (g >= games) return;                                                 // (whole waves)
	uint32_t s[5] = {SOLVED_DW[0], SOLVED_DW[1], SOLVED_DW[2], SOLVED_DW[3], SOLVED_DW[4]};
	const int rows = moves + (with_solved ? 1 : 0);
	uint32_t *o = out + (size_t)g * (only_last ? 1 : rows) * STATE_DWORDS;
	if (!only_last && with_solved) {
		if (lane < 5) o[lane] = SOLVED_DW[lane];
		o += STATE_DWORDS;
	}
	uint32_t worst = 0;
	uint32_t a_next = lane < moves ? (uint32_t)actions[(size_t)lane * games + g] : 0xFFu;
	for (int d0 = 0; d0 < moves; d0 += 64) {
		const int n = moves - d0 < 64 ? moves - d0 : 64;                     // moves of this chunk
		uint32_t a = a_next;
		// the next chunk's action byte is requested before this chunk is scanned: its latency hides behind six scan steps
		a_next = d0 + 64 + lane < moves ? (uint32_t)actions[(size_t)(d0 + 64 + lane) * games + g] : 0xFFu;
		uint32_t X[12];
		if (lane < n) {
			worst = a > worst ? a : worst;
			a = a < 12u ? a : 0u;
			load_action_table(s_act, a, X);
		} else {
			identity_moves(X);
		}
		scan_moves(X, lane, lane, n);
		uint32_t st[5] = {s[0], s[1], s[2], s[3], s[4]};
		move5(st, X);                                                        // the state after move d0 + lane
		if (!only_last && lane < n) {
			#pragma unroll
			for (int j = 0; j < 5; j++) o[(size_t)(d0 + lane) * STATE_DWORDS + j] = st[j];
		}
		#pragma unroll
		for (int j = 0; j < 5; j++) s[j] = (uint32_t)__builtin_amdgcn_readlane((int)st[j], n - 1);   // carries into the next chunk
	}
	if (only_last && lane < 5) o[lane] = s[lane];
	if (worst >= 12u) atomicOr(&g_bad_actions, 1u);                          // never taken on valid input
}

// ================================================================================================================
// rollout_fanout: the cube part of one ADI rollout in ONE launch                                   train.py:277-292
//   states of `games` random walks along their `rows` steps (cube.py:218-232, game-major)  +  is each of them solved (train.py:281)
//   +  their 12 children, parent-major (train.py:285)  +  is each child solved (train.py:292).
// Round 4 ran this as three launches (k_apply_sequences, k_multi_is_solved, k_expand12r) on 225 k states -- 13 MB of states and
// 61 MB of children, every launch in the band where the 3-5 us a launch costs before its first byte is a quarter of it, and the
// states written by the first were read back by the other two.  Here a lane owns ONE state (game g, row r): the lanes of a game
// compute their states TOGETHER (a prefix scan over the moves' permutation tables, see below; games of more than 64 rows: every
// lane walks for itself) and hand them to the fan-out's own expand_lane.  Nothing is read back: 1 action byte in,
// 20 + 1 + 240 + 12 bytes out.
// ================================================================================================================
__global__ __launch_bounds__(EXP_WAVES * WAVE)
void k_rollout_fanout(const uint8_t *__restrict__ actions, int moves, int games, int with_solved, uint32_t *__restrict__ states,
                      uint8_t *__restrict__ state_flags, u32x4 *__restrict__ children, uint32_t *__restrict__ solved, long long *__restrict__ stats)
{
	__shared__ u32x4 s_act[36];
	__shared__ u32x4 s_rows[48];
	__shared__ ExpandWaveLdsT<1> s_wave[EXP_WAVES];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	stage_action_tables(s_act, tid);
	if (tid < 48) {
		const uint32_t *src = reinterpret_cast<const uint32_t *>(D_TAB.rows) + 4 * tid;
		s_rows[tid] = u32x4{src[0], src[1], src[2], src[3]};
	}
	__syncthreads();
	const int rows = moves + (with_solved ? 1 : 0);
	const size_t n = (size_t)games * rows;
	// rows <= 64: a wave's tile is a whole number of games (2 games x 30 rows = 60 lanes for the reference's rollout) and the walks
	// are a parallel prefix over the lanes of a game (below); longer games: 64 consecutive states, every lane walks for itself
	const bool scan = rows <= EXP_ROUND;
	const int tile_states = scan ? (EXP_ROUND / rows) * rows : EXP_ROUND;
	const size_t p0 = ((size_t)blockIdx.x * EXP_WAVES + wv) * tile_states;         // first state of this wave's tile
	if (p0 >= n) return;                                                       // (after the only barrier)
	const int np = n - p0 < (size_t)tile_states ? (int)(n - p0) : tile_states;
	const size_t me = p0 + (lane < np ? lane : np - 1);                            // lanes past the end redo the last state; never stored
	const int g = (int)(me / rows), r = (int)(me - (size_t)g * rows);
	uint32_t par[5] = {SOLVED_DW[0], SOLVED_DW[1], SOLVED_DW[2], SOLVED_DW[3], SOLVED_DW[4]};
	uint32_t worst = 0;
	if (scan) {
		// The state of row r is the composition of the game's first moves applied to the solved state, and the compositions of all rows
		// of a game are ONE scan_moves over its lanes, instead of up to `rows` dependent moves per wave (thirty LDS round trips + v_perm
		// chains: the walk was most of this kernel's 28 us; rocprofv3, profiles/r05_adi_cube_kernels.csv).
		uint32_t X[12];
		const int mv = with_solved ? r - 1 : r;                                // the move that leads to row r (none for the solved row)
		if (mv >= 0) {
			uint32_t a = actions[(size_t)mv * games + g];
			worst = a;
			a = a < 12u ? a : 0u;
			load_action_table(s_act, a, X);
		} else {
			identity_moves(X);
		}
		scan_moves(X, lane, r, rows);
		move5(par, X);
	} else {
		const int todo = with_solved ? r : r + 1;
		constexpr int CH = 16;
		for (int d0 = 0; d0 < todo; d0 += CH) {
			uint32_t acts[CH];
			fetch_actions<CH>(actions, (size_t)games, (size_t)g, d0, todo, acts);
			#pragma unroll
			for (int k = 0; k < CH; k++) {
				if (d0 + k < todo) {
					uint32_t a = acts[k];
					worst = a > worst ? a : worst;
					a = a < 12u ? a : 0u;
					uint32_t tab[12];
					load_action_table(s_act, a, tab);
					move5(par, tab);
				}
			}
		}
	}
	if (worst >= 12u) atomicOr(&g_bad_actions, 1u);                                // never taken on valid input
	const ExpandCtx c{s_rows, s_wave[wv].stage, s_wave[wv].flags, lane};
	// the tile's states, coalesced through the head of the staging area
	uint32_t *stage_dw = reinterpret_cast<uint32_t *>(c.stage);
	#pragma unroll
	for (int j = 0; j < 5; j++) stage_dw[lane * 5 + j] = par[j];
	wave_lds_fence();
	#pragma unroll
	for (int k = 0; k < 5; k++) {
		const int idx = k * 64 + lane;
		if (idx < np * STATE_DWORDS) states[p0 * STATE_DWORDS + idx] = stage_dw[idx];
	}
	if (state_flags != nullptr && lane < np) state_flags[p0 + lane] = is_solved5(par) ? 1 : 0;
	wave_lds_fence();
	// ... and their children, as the fan-out's ragged tile writes them
	uint32_t out[60], fl[3];
	expand_lane<true>(c, par, out, fl);
	c.flags[lane * 3 + 0] = fl[0];
	c.flags[lane * 3 + 1] = fl[1];
	c.flags[lane * 3 + 2] = fl[2];
	#pragma unroll
	for (int v = 0; v < 15; v++)
		c.stage[lane * 15 + v] = u32x4{out[4 * v], out[4 * v + 1], out[4 * v + 2], out[4 * v + 3]};
	wave_lds_fence();
	u32x4 *dst = children + p0 * 15;
	const int nvec = np * 15;
	#pragma unroll
	for (int v = 0; v < 15; v++) {
		const int idx = v * 64 + lane;
		if (idx < nvec) store16<1>(dst, idx, c.stage[idx]);
	}
	uint32_t *fdst = solved + p0 * 3;
	#pragma unroll
	for (int k = 0; k < 3; k++) {
		const int idx = k * 64 + lane;
		if (idx < np * 3) fdst[idx] = c.flags[idx];
	}
	report_solved<true>(fl, lane < np, (p0 + lane) * 12, stats);
}

// ================================================================================================================
// as_oh: (n, 20) int8 -> (n, 480) one-hot, oh[r][24 i + s[r][i]] = 1                            cube.py:265-277
// Algorithmic bytes per state: 20 read + 480 * sizeof(T) written (1 920 for f32): a pure store stream.  A workgroup
// encodes TILE states per step; each thread emits 16-byte chunks (4 f32 or 8 half/bf16 columns of one cubie).
// ================================================================================================================
template <typename T> struct OhOne;
template <> struct OhOne<float>    { static constexpr uint32_t bits = 0x3F800000u; };
template <> struct OhOne<_Float16> { static constexpr uint32_t bits = 0x3C00u; };
struct bf16_tag {};
template <> struct OhOne<bf16_tag> { static constexpr uint32_t bits = 0x3F80u; };

// tau_ps > 0: the paced form of the fan-out (k_expand12p): phases of `phase_tiles` tiles, each opened by `pull_wgs` workgroups that read
// the phase's states into the Infinity Cache and publish the time base; every other workgroup owns one tile and holds its stores
// until base + lead + slot x tau.  (The states are 1-2 % of the traffic, but a tile that waits for them in HBM behind the store
// stream misses its slot: without the read phase the paced form gains 7 % for f32 and loses 7 % for bf16.)  NT: non-temporal stores.
template <typename T, int ELEM_BYTES, int TILE = 64, bool NT = false, int THREADS = 256>
__global__ __launch_bounds__(THREADS)
void k_as_oh(const uint32_t *__restrict__ states, u32x4 *__restrict__ out, size_t n, size_t n_tiles, unsigned tau_ps = 0, unsigned lead = 0,
             unsigned pull_wgs = 0, unsigned phase_tiles = 0, unsigned long long *pace = nullptr)
{
	constexpr int E = 16 / ELEM_BYTES;            // columns per 16-byte chunk: 4 or 8
	constexpr int CHUNKS_PER_ROW = 480 / E;       // 120 or 60
	constexpr int CHUNKS_PER_CUBIE = 24 / E;      // 6 or 3
	__shared__ uint32_t s_st[TILE * STATE_DWORDS];
	const int tid = threadIdx.x;
	const uint8_t *s_bytes = reinterpret_cast<const uint8_t *>(s_st);
	unsigned long long start = 0, base = 0;
	size_t tile_first = blockIdx.x, tile_stride = gridDim.x, slot = 0;
	if (tau_ps > 0) {                                                   // paced: phases of pull_wgs readers + phase_tiles one-tile workgroups
		const unsigned wgs_per_phase = pull_wgs + phase_tiles;
		const unsigned phase = blockIdx.x / wgs_per_phase, r = blockIdx.x - phase * wgs_per_phase;
		if (r < pull_wgs) {                                               // leaves before any barrier
			const size_t p_first = (size_t)phase * phase_tiles * TILE;
			const size_t p_count = n - p_first < (size_t)phase_tiles * TILE ? n - p_first : (size_t)phase_tiles * TILE;
			pull_front(states + p_first * STATE_DWORDS, p_count, (size_t)r * (THREADS / 64) + (tid >> 6), (size_t)pull_wgs * (THREADS / 64), tid & 63);
			if (tid == 0 && ((r & 63) == 63 || r + 1 == pull_wgs))
				__hip_atomic_fetch_max(pace, (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			return;
		}
		start = __builtin_amdgcn_s_memrealtime();
		slot = r - pull_wgs;
		tile_first = (size_t)phase * phase_tiles + slot;
		tile_stride = n_tiles;                                             // one tile per workgroup
		if (pull_wgs == 0 && slot == 0 && tid == 0) __hip_atomic_fetch_max(pace, start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		base = __hip_atomic_load(pace, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}

	for (size_t tile = tile_first; tile < n_tiles; tile += tile_stride) {
		const size_t p0 = tile * TILE;
		const int np = (int)((n - p0 < (size_t)TILE) ? (n - p0) : (size_t)TILE);
		const int ndw = np * STATE_DWORDS;
		for (int idx = tid; idx < TILE * STATE_DWORDS; idx += THREADS) s_st[idx] = idx < ndw ? states[p0 * STATE_DWORDS + idx] : 0u;
		__syncthreads();
		PaceHold{base, start, slot, tau_ps, lead, tid & 63, pace}();
		const int nchunks = np * CHUNKS_PER_ROW;
		u32x4 *dst = out + p0 * CHUNKS_PER_ROW;
		for (int c = tid; c < nchunks; c += THREADS) {
			const int r = c / CHUNKS_PER_ROW, g = c - r * CHUNKS_PER_ROW;
			const int cubie = g / CHUNKS_PER_CUBIE;
			const int base = (g - cubie * CHUNKS_PER_CUBIE) * E;
			const int rel = (int)s_bytes[r * STATE_BYTES + cubie] - base;     // position of the 1 inside this chunk, if any
			u32x4 val = {0u, 0u, 0u, 0u};
			if (ELEM_BYTES == 4) {
				val.x = rel == 0 ? OhOne<T>::bits : 0u;
				val.y = rel == 1 ? OhOne<T>::bits : 0u;
				val.z = rel == 2 ? OhOne<T>::bits : 0u;
				val.w = rel == 3 ? OhOne<T>::bits : 0u;
			} else {
				const uint32_t one = OhOne<T>::bits << (16 * (rel & 1));
				val.x = (rel >> 1) == 0 && rel >= 0 ? one : 0u;
				val.y = (rel >> 1) == 1 ? one : 0u;
				val.z = (rel >> 1) == 2 ? one : 0u;
				val.w = (rel >> 1) == 3 ? one : 0u;
			}
			if (NT) __builtin_nontemporal_store(val, dst + c);
			else dst[c] = val;
		}
		__syncthreads();
	}
}

// ================================================================================================================
// 6x8x6 representation (cube.py:311-388).  A state is 48 sticker slots x 6 one-hot int8 = 144 ushorts; a move is a
// permutation of the slots, so every kernel is a gather of 3-ushort groups through the 576-byte slot table in LDS.
// These are "next" rows of the scope table: correct and coalesced on the store side, not yet tuned.
// ================================================================================================================
__device__ __forceinline__ void stage_perm686(uint8_t *lds, int tid, int nthreads)
{
	const uint8_t *src = &D_TAB.perm686[0][0];
	for (int i = tid; i < N_ACTIONS * S686_SLOTS; i += nthreads) lds[i] = src[i];
}

// FANOUT = false: out[r] = move actions[r] of states[r].  FANOUT = true: out[12 r + a] = move a of states[r].
// A state is 144 ushorts (48 slots x 3).  A workgroup stages GROUP source states in LDS with 16 B/lane loads; each
// thread then produces 16-byte chunks (8 ushorts) of the output, gathering its 8 source ushorts from LDS through the
// per-action source-offset table src[a][u] = perm686[a][u/3]*3 + u%3 (also LDS).  Stores are 16 B/lane, contiguous
// across lanes and non-temporal: a store stream.
// FLAGS (fan-out only): the 12 solved flags of every parent come from the LDS-resident PARENT in the same launch
// (cube.py:88-89 on the children without reading them back): child a of p is solved <=> p == move rev(a) of solved.
// The staged parent is reduced to its 48 slot colours (a slot that is not an exact one-hot gets 255, so that garbage
// never compares equal) and each (parent, action) pair compares 12 dwords of colours with the "one move from solved"
// colour pattern near[a][s] = face(perm686[a^1][s]).
template <bool FANOUT, bool FLAGS = false>
__global__ __launch_bounds__(256)
void k_rotate686(const uint16_t *__restrict__ states, const uint8_t *__restrict__ actions, u32x4 *__restrict__ out, size_t n_in,
                 uint8_t *__restrict__ flags = nullptr, long long *__restrict__ stats = nullptr)
{
	constexpr int GROUP = FANOUT ? 4 : 64;                // source states per workgroup step (about four stores per thread)
	constexpr int OUT_PER_IN = FANOUT ? 12 : 1;
	__shared__ __attribute__((aligned(16))) uint8_t s_src[N_ACTIONS * 144];
	__shared__ __attribute__((aligned(16))) uint16_t s_in[GROUP * 144];
	__shared__ __attribute__((aligned(16))) uint8_t s_near[FLAGS ? N_ACTIONS * S686_SLOTS : 16];
	__shared__ __attribute__((aligned(16))) uint8_t s_col[FLAGS ? GROUP * S686_SLOTS : 16];
	if (threadIdx.x < N_ACTIONS * 144 / 16)               // 1 728 B of source offsets: 108 16-byte loads from the constant segment
		reinterpret_cast<u32x4 *>(s_src)[threadIdx.x] = reinterpret_cast<const u32x4 *>(&D_TAB.src686[0][0])[threadIdx.x];
	if (FLAGS && threadIdx.x >= 128 && threadIdx.x < 128 + N_ACTIONS * S686_SLOTS / 16)
		reinterpret_cast<u32x4 *>(s_near)[threadIdx.x - 128] = reinterpret_cast<const u32x4 *>(&D_TAB.near686[0][0])[threadIdx.x - 128];
	const size_t n_groups = (n_in + GROUP - 1) / GROUP;
	// Software pipeline of the input (round 3, as in the 20-byte fan-out): a workgroup walks several groups and requests the
	// NEXT group's states (GROUP x 18 16-byte words, PW per thread, clamped so that every load is in range)
	// before it gathers and stores the current one -- with the states coming from HBM a workgroup that only ever sees one
	// group waits a full memory latency for it first (cache-neutral: 0.61 of peak for the fan-out, 0.76 with the input
	// cache-resident; profiles/r03_kernels686.json).
	constexpr int PW = (GROUP * 18 + 255) / 256;
	u32x4 pre[PW];
	const u32x4 *all4 = reinterpret_cast<const u32x4 *>(states);
	const size_t last_word = n_in * 18 - 1;
	auto request = [&](size_t group) {
		#pragma unroll
		for (int k = 0; k < PW; k++) {
			const size_t w = group * (GROUP * 18) + (size_t)(k * 256 + threadIdx.x);
			if (k * 256 + (int)threadIdx.x < GROUP * 18) pre[k] = all4[w < last_word ? w : last_word];
		}
	};
	if (blockIdx.x < n_groups) request(blockIdx.x);
	for (size_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
		const size_t first = g * GROUP;
		const int ng = (int)((n_in - first < (size_t)GROUP) ? (n_in - first) : (size_t)GROUP);
		__syncthreads();                                  // previous step's gathers are done (and the tables are ready)
		#pragma unroll
		for (int k = 0; k < PW; k++) {
			const int i = k * 256 + threadIdx.x;
			if (i < GROUP * 18) reinterpret_cast<u32x4 *>(s_in)[i] = pre[k];
		}
		if (g + gridDim.x < n_groups) request(g + gridDim.x);
		__syncthreads();
		if (FLAGS && (int)threadIdx.x < ng * S686_SLOTS) {   // slot colours of the staged parents (threads 0..191)
			const uint16_t *h = s_in + threadIdx.x * 3;       // (parent, slot) -> 3 ushorts = 6 one-hot bytes
			const uint32_t lo = (uint32_t)h[0] | ((uint32_t)h[1] << 16), hi = h[2];
			uint32_t col = 255u;
			if (hi == 0u) {
				if (lo == 0x00000001u) col = 0; else if (lo == 0x00000100u) col = 1;
				else if (lo == 0x00010000u) col = 2; else if (lo == 0x01000000u) col = 3;
			} else if (lo == 0u) {
				if (hi == 0x0001u) col = 4; else if (hi == 0x0100u) col = 5;
			}
			s_col[threadIdx.x] = (uint8_t)col;
		}
		u32x4 *dst = out + first * OUT_PER_IN * 18;
		const int n_chunks = ng * OUT_PER_IN * 18;
		for (int q = threadIdx.x; q < n_chunks; q += 256) {
			const int r = q / 18, k = q - r * 18;          // output row inside the group, chunk inside the row
			int local;
			uint32_t a;
			if (FANOUT) { local = r / 12; a = (uint32_t)(r - local * 12); }
			else        { local = r; a = actions[first + r]; if (a >= 12u) atomicOr(&g_bad_actions, 1u); a = a < 12u ? a : 0u; }
			const uint16_t *row = s_in + local * 144;
			const u32x2 offs = *reinterpret_cast<const u32x2 *>(&s_src[a * 144 + k * 8]);
			uint32_t h[8];
			#pragma unroll
			for (int i = 0; i < 8; i++) h[i] = row[((i < 4 ? offs.x : offs.y) >> (8 * (i & 3))) & 0xFFu];
			__builtin_nontemporal_store(u32x4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)}, dst + q);
		}
		if (FLAGS) {
			__syncthreads();                              // s_col complete (the gathers above hid its latency)
			if ((int)threadIdx.x < ng * N_ACTIONS) {      // one thread per (parent, action): 12 dword compares
				const int local = threadIdx.x / N_ACTIONS, a = threadIdx.x - local * N_ACTIONS;
				const uint32_t *c = reinterpret_cast<const uint32_t *>(s_col + local * S686_SLOTS);
				const uint32_t *w = reinterpret_cast<const uint32_t *>(s_near + a * S686_SLOTS);
				uint32_t diff = 0;
				#pragma unroll
				for (int k = 0; k < S686_SLOTS / 4; k++) diff |= c[k] ^ w[k];
				const size_t o = first * N_ACTIONS + threadIdx.x;
				if (flags != nullptr) flags[o] = diff == 0u ? 1 : 0;
				if (diff == 0u && stats != nullptr) {
					atomicAdd(reinterpret_cast<unsigned long long *>(&stats[0]), 1ull);
					atomicMin(&stats[1], (long long)o);
				}
			}
		}
	}
}

// goal test of 288-byte states: a workgroup takes 128 states per step, every thread compares nine 16-byte chunks
// (coalesced) with the solved pattern's chunk and leaves a mismatch byte in LDS; 128 threads then fold 18 bytes each.
__global__ __launch_bounds__(256)
void k_is_solved686(const u32x4 *__restrict__ states, uint8_t *__restrict__ flags, long long *__restrict__ stats, size_t n)
{
	constexpr int GROUP = 128;
	__shared__ u32x4 s_sol[18];
	__shared__ uint8_t s_bad[GROUP * 18];
	if (threadIdx.x < 18) {
		const uint32_t *src = reinterpret_cast<const uint32_t *>(D_TAB.solved686) + 4 * threadIdx.x;
		s_sol[threadIdx.x] = u32x4{src[0], src[1], src[2], src[3]};
	}
	const size_t n_groups = (n + GROUP - 1) / GROUP;
	for (size_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
		const size_t first = g * GROUP;
		const int ng = (int)((n - first < (size_t)GROUP) ? (n - first) : (size_t)GROUP);
		__syncthreads();
		const u32x4 *src = states + first * 18;
		for (int q = threadIdx.x; q < ng * 18; q += 256) {
			const u32x4 v = src[q], w = s_sol[q % 18];
			s_bad[q] = (uint8_t)(((v.x ^ w.x) | (v.y ^ w.y) | (v.z ^ w.z) | (v.w ^ w.w)) != 0u);
		}
		__syncthreads();
		if ((int)threadIdx.x < ng) {
			uint32_t bad = 0;
			#pragma unroll
			for (int k = 0; k < 18; k++) bad |= s_bad[threadIdx.x * 18 + k];
			if (flags != nullptr) flags[first + threadIdx.x] = bad ? 0 : 1;
			if (!bad && stats != nullptr) {
				atomicAdd(reinterpret_cast<unsigned long long *>(&stats[0]), 1ull);
				atomicMin(&stats[1], (long long)(first + threadIdx.x));
			}
		}
	}
}

// The 6x8x6 fan-out in the paced form of k_expand12p: phases of `phase_groups` groups of GROUP parents, each opened by `pull_wgs`
// reading workgroups; every other workgroup owns ONE group -- stages it, reduces it to slot colours for the goal test, HOLDS until
// base + lead + slot x tau, then gathers and stores its GROUP x 12 children (GROUP x 3 456 B).  GROUP = 8: with four parents per
// workgroup the 2 048 resident workgroups hold 28 MB of children, which at this kernel's 5 us from start to last store is 5.6 TB/s
// -- 98 % of the groups missed their slots (profiles/r03_paced686_timeline.txt); eight parents per workgroup double what is in flight.
template <bool FLAGS, int GROUP>
__global__ __launch_bounds__(256)
void k_fanout686p(const uint16_t *__restrict__ states, u32x4 *__restrict__ out, size_t n_in, uint8_t *__restrict__ flags, long long *__restrict__ stats,
                  unsigned pull_wgs, unsigned phase_groups, unsigned tau_ps, unsigned lead, unsigned long long *pace)
{
	__shared__ __attribute__((aligned(16))) uint8_t s_src[N_ACTIONS * 144];
	__shared__ __attribute__((aligned(16))) uint16_t s_in[GROUP * 144];
	__shared__ __attribute__((aligned(16))) uint8_t s_near[FLAGS ? N_ACTIONS * S686_SLOTS : 16];
	__shared__ __attribute__((aligned(16))) uint8_t s_col[FLAGS ? GROUP * S686_SLOTS : 16];
	const int tid = threadIdx.x;
	const unsigned wgs_per_phase = pull_wgs + phase_groups;
	const unsigned phase = blockIdx.x / wgs_per_phase, r = blockIdx.x - phase * wgs_per_phase;
	if (r < pull_wgs) {                                                 // read phase; leaves before any barrier
		const size_t p_first = (size_t)phase * phase_groups * GROUP;
		const size_t p_count = n_in - p_first < (size_t)phase_groups * GROUP ? n_in - p_first : (size_t)phase_groups * GROUP;
		pull_front_bytes(states + p_first * 144, p_count * S686_BYTES, (size_t)r * 4 + (tid >> 6), (size_t)pull_wgs * 4, tid & 63);
		if (tid == 0 && ((r & 63) == 63 || r + 1 == pull_wgs))
			__hip_atomic_fetch_max(pace, (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		return;
	}
	const unsigned long long start = __builtin_amdgcn_s_memrealtime();
	const size_t slot = r - pull_wgs, g = (size_t)phase * phase_groups + slot;
	if (pull_wgs == 0 && slot == 0 && tid == 0) __hip_atomic_fetch_max(pace, start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	const unsigned long long base = __hip_atomic_load(pace, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	const size_t first = g * GROUP;
	const int ng = first >= n_in ? 0 : (int)((n_in - first < (size_t)GROUP) ? (n_in - first) : (size_t)GROUP);
	// the group's states (ng x 18 16-byte words) and the tables travel together
	constexpr int PW = (GROUP * 18 + 255) / 256;
	u32x4 pre[PW];
	const u32x4 *all4 = reinterpret_cast<const u32x4 *>(states) + first * 18;
	#pragma unroll
	for (int k = 0; k < PW; k++) { const int i = k * 256 + tid; if (i < ng * 18) pre[k] = all4[i]; }
	if (tid < N_ACTIONS * 144 / 16)
		reinterpret_cast<u32x4 *>(s_src)[tid] = reinterpret_cast<const u32x4 *>(&D_TAB.src686[0][0])[tid];
	if (FLAGS && tid >= 128 && tid < 128 + N_ACTIONS * S686_SLOTS / 16)
		reinterpret_cast<u32x4 *>(s_near)[tid - 128] = reinterpret_cast<const u32x4 *>(&D_TAB.near686[0][0])[tid - 128];
	#pragma unroll
	for (int k = 0; k < PW; k++) { const int i = k * 256 + tid; if (i < ng * 18) reinterpret_cast<u32x4 *>(s_in)[i] = pre[k]; }
	__syncthreads();
	if (FLAGS)
		for (int i = tid; i < ng * S686_SLOTS; i += 256) {                  // slot colours of the staged parents (k_rotate686)
			const uint16_t *h = s_in + i * 3;
			const uint32_t lo = (uint32_t)h[0] | ((uint32_t)h[1] << 16), hi = h[2];
			uint32_t col = 255u;
			if (hi == 0u) {
				if (lo == 0x00000001u) col = 0; else if (lo == 0x00000100u) col = 1;
				else if (lo == 0x00010000u) col = 2; else if (lo == 0x01000000u) col = 3;
			} else if (lo == 0u) {
				if (hi == 0x0001u) col = 4; else if (hi == 0x0100u) col = 5;
			}
			s_col[i] = (uint8_t)col;
		}
	PaceHold{base, start, slot, tau_ps, lead, tid & 63, pace}();
	u32x4 *dst = out + first * 12 * 18;
	const int n_chunks = ng * 12 * 18;
	for (int q = tid; q < n_chunks; q += 256) {
		const int row_i = q / 18, k = q - row_i * 18;
		const int local = row_i / 12;
		const uint32_t a = (uint32_t)(row_i - local * 12);
		const uint16_t *row = s_in + local * 144;
		const u32x2 offs = *reinterpret_cast<const u32x2 *>(&s_src[a * 144 + k * 8]);
		uint32_t h[8];
		#pragma unroll
		for (int i = 0; i < 8; i++) h[i] = row[((i < 4 ? offs.x : offs.y) >> (8 * (i & 3))) & 0xFFu];
		__builtin_nontemporal_store(u32x4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)}, dst + q);
	}
	if (FLAGS) {
		__syncthreads();                                                  // s_col complete
		for (int i = tid; i < ng * N_ACTIONS; i += 256) {
			const int local = i / N_ACTIONS, a = i - local * N_ACTIONS;
			const uint32_t *c = reinterpret_cast<const uint32_t *>(s_col + local * S686_SLOTS);
			const uint32_t *w = reinterpret_cast<const uint32_t *>(s_near + a * S686_SLOTS);
			uint32_t diff = 0;
			#pragma unroll
			for (int k = 0; k < S686_SLOTS / 4; k++) diff |= c[k] ^ w[k];
			const size_t o = first * N_ACTIONS + i;
			if (flags != nullptr) flags[o] = diff == 0u ? 1 : 0;
			if (diff == 0u && stats != nullptr) {
				atomicAdd(reinterpret_cast<unsigned long long *>(&stats[0]), 1ull);
				atomicMin(&stats[1], (long long)o);
			}
		}
	}
}

// int8 one-hot -> T one-hot, elementwise widening (cube.py:363-369)
template <typename T, int ELEM_BYTES>
__global__ __launch_bounds__(256)
void k_as_oh686(const uint32_t *__restrict__ states, u32x4 *__restrict__ out, size_t n_dwords)
{
	// each thread widens one dword (4 int8) into 4 elements
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_dwords; i += (size_t)gridDim.x * blockDim.x) {
		const uint32_t x = states[i];
		if (ELEM_BYTES == 4) {
			u32x4 v;
			v.x = (x & 0xFFu) ? OhOne<T>::bits : 0u;
			v.y = (x & 0xFF00u) ? OhOne<T>::bits : 0u;
			v.z = (x & 0xFF0000u) ? OhOne<T>::bits : 0u;
			v.w = (x & 0xFF000000u) ? OhOne<T>::bits : 0u;
			out[i] = v;
		} else {
			u32x2 v;
			v.x = ((x & 0xFFu) ? OhOne<T>::bits : 0u) | ((x & 0xFF00u) ? OhOne<T>::bits << 16 : 0u);
			v.y = ((x & 0xFF0000u) ? OhOne<T>::bits : 0u) | ((x & 0xFF000000u) ? OhOne<T>::bits << 16 : 0u);
			reinterpret_cast<u32x2 *>(out)[i] = v;
		}
	}
}

// as_correct (cube.py:371-380): +1 where slot (f, p) shows colour f, else -1
__global__ __launch_bounds__(256)
void k_as_correct686(const int8_t *__restrict__ states, float *__restrict__ out, size_t n_slots)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += (size_t)gridDim.x * blockDim.x) {
		const int slot = (int)(i % 48), face = slot >> 3;
		const int8_t *s = states + i * 6;
		bool ok = true;
		#pragma unroll
		for (int c = 0; c < 6; c++) ok &= s[c] == (c == face ? 1 : 0);
		out[i] = ok ? 1.0f : -1.0f;
	}
}

// ================================================================================================================
// launchers
// ================================================================================================================
static inline unsigned grid_for(size_t work_items, size_t per_block, unsigned cap)
{
	size_t b = (work_items + per_block - 1) / per_block;
	if (b < 1) b = 1;
	if (b > cap) b = cap;
	return (unsigned)b;
}

// The paced form's constants: tau = time per tile of the store schedule, the lead between the end of the read phase
// and tile 0's slot, readers per phase, tiles per phase.  RK_PACE=0 switches the form off, RK_PACE_TAU_PS / RK_PACE_LEAD /
// RK_PACE_PULL / RK_PACE_PHASE override (tuning).
struct PaceConfig { bool on; unsigned tau_ps, lead, pull_wgs, phase_tiles; size_t min_tiles; unsigned pull_first; bool tau_from_env, calibrate; };
static const PaceConfig &pace_config()
{
	static const PaceConfig cfg = [] {
		auto env = [](const char *name, long dflt) { const char *e = std::getenv(name); return e ? std::atol(e) : dflt; };
		PaceConfig c;
		c.on = env("RK_PACE", 1) != 0;
		c.tau_ps = (unsigned)env("RK_PACE_TAU_PS", PACE_TAU_PS);
		c.tau_from_env = std::getenv("RK_PACE_TAU_PS") != nullptr;
		c.calibrate = env("RK_PACE_CALIBRATE", 1) != 0;
		c.lead = (unsigned)env("RK_PACE_LEAD", PACE_LEAD_TICKS);
		c.pull_wgs = (unsigned)env("RK_PACE_PULL", PACE_PULL_WGS);
		c.phase_tiles = (unsigned)env("RK_PACE_PHASE", PACE_PHASE_TILES) / EXP_WAVES * EXP_WAVES;
		if (c.phase_tiles < 4096) c.phase_tiles = 4096;
		c.min_tiles = (size_t)env("RK_PACE_MIN", (long)PACE_MIN_TILES);
		c.pull_first = (unsigned)env("RK_PACE_PULL_FIRST", c.pull_wgs == PACE_PULL_WGS ? (long)PACE_PULL_WGS_FIRST : (long)c.pull_wgs);
		if (c.pull_first < c.pull_wgs) c.pull_first = c.pull_wgs;
		return c;
	}();
	return cfg;
}

// What the paced forms keep PER DEVICE (a process may drive several: rk_init selects per thread): the device address of that
// device's g_pace_cells (a __device__ array has one instance per device; round 4 cached the first device's address for the whole
// process -- the advisor's finding), the store schedule calibrated on it, and the gate of PacedTurn below.  Indexed by hipGetDevice.
constexpr int PACE_MAX_DEVICES = 32;
constexpr int PACE_CANDIDATES = 5;
constexpr unsigned PACE_CANDIDATE_TAU_PS[PACE_CANDIDATES] = {0 /* the unpaced ring form */, 2000, 2100, 2200, 2400};
struct PaceDevice {
	std::once_flag cells_once;
	unsigned long long *cells = nullptr;           // null: the lookup failed on this device -> its launches stay unpaced
	std::mutex cal_mu;
	std::atomic<int> source{0};                    // 0 compiled default, 1 calibrated on this device, 2 fixed by the environment / calibration not possible
	std::atomic<unsigned> tau_ps{PACE_TAU_PS};     // schedule of the fan-out on this device; 0 = the ring form won the calibration
	float us[PACE_CANDIDATES] = {0, 0, 0, 0, 0};
};
// slot of a device in the per-device tables below (time-base cells + calibration, turn gates): its own index, or -1 for a device the tables
// have no room for -- such a device runs the unpaced forms.  ONE function decides it for every table (rk_pace_slot_of_device exposes it to
// the CPU test that pins "two devices never share a slot": the round-4 advisor's finding was one cached pointer for all devices).
int pace_slot_of_device(int dev) { return dev >= 0 && dev < PACE_MAX_DEVICES ? dev : -1; }
static PaceDevice *pace_device()
{
	static PaceDevice devices[PACE_MAX_DEVICES];
	int dev = -1;
	if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
	const int slot = pace_slot_of_device(dev);
	return slot >= 0 ? &devices[slot] : nullptr;
}
static unsigned long long *pace_cells(PaceDevice *d)
{
	if (d == nullptr) return nullptr;
	std::call_once(d->cells_once, [d] {
		void *p = nullptr;
		if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_pace_cells)) == hipSuccess) d->cells = (unsigned long long *)p;
		else (void)hipGetLastError();
	});
	return d->cells;
}

// the cell of the next paced launch ON THE CURRENT DEVICE (see g_pace_cells): the ring's cells in turn, whatever the stream.
// Never null where a launcher asks for it: pace_on() is false on a device without cells, so its launches take the unpaced forms.
static unsigned long long *next_pace_cell()
{
	static std::atomic<unsigned> turn{0};
	unsigned long long *base = pace_cells(pace_device());
	return base == nullptr ? nullptr : base + (size_t)(turn.fetch_add(1, std::memory_order_relaxed) % PACE_CELLS) * PACE_CELL_STRIDE;
}

// rk_set_pacing: -1 = what the environment says (RK_PACE, default on), 0 = the unpaced forms, 1 = the paced forms
static std::atomic<int> g_pace_override{-1};
void set_pace_override(int mode) { g_pace_override.store(mode < 0 ? -1 : (mode ? 1 : 0), std::memory_order_relaxed); }
static inline bool pace_on(const PaceConfig &pc)
{
	const int o = g_pace_override.load(std::memory_order_relaxed);
	if (!(o < 0 ? pc.on : o != 0)) return false;
	return pace_cells(pace_device()) != nullptr;                          // no time-base cells on this device: unpaced, never a null cell
}
// tau of the fan-out's schedule on the current device: what calibrate_pacing() settled on (0 = the ring form won), else the
// compiled / environment value.  rk_set_pacing(1) asks for the paced form explicitly: a calibration that chose the ring form
// then falls back to the configured tau, so that both forms can still be timed side by side.
static inline unsigned fanout_tau(const PaceConfig &pc)
{
	PaceDevice *d = pace_device();
	if (d == nullptr || d->source.load(std::memory_order_relaxed) != 1) return pc.tau_ps;
	const unsigned t = d->tau_ps.load(std::memory_order_relaxed);
	return t == 0 && g_pace_override.load(std::memory_order_relaxed) == 1 ? pc.tau_ps : t;
}
// tau (per 16 128 bytes) for the other paced store streams (one-hot, 6x8x6 fan-out): they were tuned to the same HBM rate as the
// fan-out, so they follow a calibrated schedule; a calibration won by the ring form says nothing about them (default kept)
static inline unsigned stream_tau(const PaceConfig &pc)
{
	const unsigned t = fanout_tau(pc);
	return t != 0 ? t : pc.tau_ps;
}

// Paced launches take turns.  A paced kernel schedules its stores for the WHOLE memory system (7.7 TB/s of 8); two of them in
// flight on two streams oversubscribe it and both fall off their schedules -- measured (profiles/r04_pace_streams.json, makespan
// of the pair over the two run one after the other): fan-out beside as_oh 1.05, beside the 6x8x6 fan-out 1.08, two fan-outs
// 1.23 -- worse than the unpaced kernels.  HBM-bound launches gain nothing from overlapping each other, so the library makes
// them wait for each other across streams: a paced launch on another stream than the previous paced launch first waits (on the
// device: hipStreamWaitEvent) for that one's end.  Everything else on the streams overlaps as before.  Not while a stream is being
// captured into a hipGraph (an event from outside the capture cannot be waited for inside it); RK_PACE_SERIAL=0 switches it off.
struct PacedTurn {
	struct Gate { std::mutex mu; hipEvent_t ev = nullptr; hipStream_t last = nullptr; bool valid = false; };
	static Gate *gates() { static Gate g[PACE_MAX_DEVICES]; return g; }
	static Gate &gate()
	{
		int dev = 0;
		if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
		const int slot = pace_slot_of_device(dev);
		return gates()[slot >= 0 ? slot : 0];                          // (a device without a slot never launches a paced form: pace_on() is false there)
	}
	// Streams the library may REMEMBER: the null stream, and the streams a caller has registered (rk_stream_register) -- a promise
	// that the stream stays alive until rk_stream_forget.  Nothing else is ever kept across two calls.  Round 4 remembered whatever
	// stream the last paced launch ran on and recorded an event on it later; on this HIP runtime touching a destroyed stream's handle
	// (hipEventRecord, hipStreamQuery alike) is a segmentation fault, not an error return (tests/test_cube_gpu.py,
	// test_a_stream_destroyed_between_two_paced_launches was written against exactly that).  A paced launch on an unregistered stream
	// therefore takes no turn: it may overlap another paced launch (slower, never wrong) and leaves the gate alone.
	struct Registry { std::mutex mu; std::vector<hipStream_t> live; };
	static Registry &registry() { static Registry r; return r; }
	static bool registered(hipStream_t st)
	{
		if (st == nullptr) return true;
		Registry &r = registry();
		std::lock_guard<std::mutex> lk(r.mu);
		return std::find(r.live.begin(), r.live.end(), st) != r.live.end();
	}
	static void add(hipStream_t st)
	{
		if (st == nullptr) return;
		Registry &r = registry();
		std::lock_guard<std::mutex> lk(r.mu);
		if (std::find(r.live.begin(), r.live.end(), st) == r.live.end()) r.live.push_back(st);
	}
	// rk_stream_forget: the caller is about to destroy `st`; neither the registry nor a gate may name it any more
	static void forget(hipStream_t st)
	{
		if (st == nullptr) return;
		{
			Registry &r = registry();
			std::lock_guard<std::mutex> lk(r.mu);
			r.live.erase(std::remove(r.live.begin(), r.live.end(), st), r.live.end());
		}
		for (int i = 0; i < PACE_MAX_DEVICES; i++) {
			Gate &g = gates()[i];
			std::lock_guard<std::mutex> lk(g.mu);
			if (g.valid && g.last == st) { g.valid = false; g.last = nullptr; }
		}
	}
	static bool enabled()
	{
		static const bool on = [] { const char *e = std::getenv("RK_PACE_SERIAL"); return e == nullptr || std::atoi(e) != 0; }();
		return on;
	}
	static bool capturing(hipStream_t st)
	{
		if (st == nullptr) return false;
		hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
		if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); return false; }
		return cap != hipStreamCaptureStatusNone;
	}
	hipStream_t st;
	bool on;
	// The event is recorded LAZILY, on the previous paced launch's stream, only when a paced launch arrives on another stream: a
	// record after every paced launch put a marker between back-to-back launches of one stream and cost the bench 3.4 us per launch
	// (the next launch's read phase no longer overlapped the previous one's tail: 0.84 -> 0.77 of peak).  Recorded late, the event also
	// covers whatever else that stream was given since -- waiting for a little more than necessary, never for less.  The remembered
	// stream is alive by the registration contract (include/rubiks_hip.h, rk_stream_register / rk_stream_forget).
	PacedTurn(hipStream_t s, bool paced) : st(s), on(paced && enabled() && registered(s) && !capturing(s))
	{
		if (!on) return;
		Gate &g = gate();
		std::lock_guard<std::mutex> lk(g.mu);
		if (!g.valid || g.last == st || capturing(g.last)) return;
		if (g.ev == nullptr && hipEventCreateWithFlags(&g.ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); g.ev = nullptr; }
		if (g.ev != nullptr) {
			if (hipEventRecord(g.ev, g.last) == hipSuccess) (void)hipStreamWaitEvent(st, g.ev, 0);
			else { (void)hipGetLastError(); g.valid = false; g.last = nullptr; }
		}
	}
	~PacedTurn()
	{
		if (!on) return;
		Gate &g = gate();
		std::lock_guard<std::mutex> lk(g.mu);
		g.last = st;
		g.valid = true;
	}
};
void register_stream(hipStream_t st) { PacedTurn::add(st); }
void forget_stream(hipStream_t st) { PacedTurn::forget(st); }

static void launch_expand12_paced(const int8_t *parents, int8_t *children, uint8_t *solved, long long *stats, size_t n, const PaceConfig &pc,
                                  unsigned tau_ps, hipStream_t st, bool take_turn = true)
{
	PacedTurn turn(st, take_turn);
	const size_t n_tiles = (n + EXP_ROUND - 1) / EXP_ROUND;
	const size_t n_phases = (n_tiles + pc.phase_tiles - 1) / pc.phase_tiles;
	const size_t last_tiles = n_tiles - (n_phases - 1) * pc.phase_tiles;
	const unsigned extra = pc.pull_first - pc.pull_wgs;
	const size_t grid = extra + (n_phases - 1) * (pc.pull_wgs + pc.phase_tiles / EXP_WAVES) + pc.pull_wgs + (last_tiles + EXP_WAVES - 1) / EXP_WAVES;
	if (solved != nullptr)
		hipLaunchKernelGGL((k_expand12p<true>), dim3((unsigned)grid), dim3(EXP_WAVES * WAVE), 0, st, (const uint32_t *)parents, (u32x4 *)children,
			(uint32_t *)solved, stats, n, pc.pull_wgs, extra, pc.phase_tiles, tau_ps, pc.lead, next_pace_cell());
	else
		hipLaunchKernelGGL((k_expand12p<false>), dim3((unsigned)grid), dim3(EXP_WAVES * WAVE), 0, st, (const uint32_t *)parents, (u32x4 *)children,
			(uint32_t *)nullptr, (long long *)nullptr, n, pc.pull_wgs, extra, pc.phase_tiles, tau_ps * 15360u / 16128u, pc.lead, next_pace_cell());
}


// Shipping shapes (benchmarks/tune_expand.py, profiles/r03_tune_sizes.json: every size measured with the parents coming
// from HBM -- inputs rotating over >= 640 MB --, 250 k ... 32 M parents, fraction of the 8 TB/s peak, round-2 kernel first):
//   parents      r02 kernel   one tile per wave (depth 0)   ring 2, grid = tiles/8   ring 2, 3 072 workgroups
//   250 k        0.54         0.61                          0.62                     0.53
//   500 k        0.62         0.64                          0.68                     0.63
//   1 M          0.70         0.70                          0.67                     0.72
//   2 M / 4 M    0.68 / 0.67  0.74 / 0.76                   0.67 / 0.67              0.67 / 0.66
//   8 M          0.77         0.78                          0.79                     0.78
//   16 M / 32 M  0.75 / 0.74  0.79 / 0.78                   0.74 / 0.76              0.75 / 0.75
// A grid with ONE tile per wave wins wherever there are many more tiles than resident waves (2 048): fresh workgroups
// dispatched in address order keep the write front dense.  Around 1 M parents (a handful of tiles per resident wave) a
// persistent grid whose waves keep the next two tiles' parents in flight hides the HBM read latency that a one-tile wave
// would wait out: 3 072 workgroups (six full residencies) at 0.75-1.5 M parents, tiles/8 workgroups below that.
static void launch_expand12_unpaced(const int8_t *parents, int8_t *children, uint8_t *solved, long long *stats, size_t n, hipStream_t st)
{
	const size_t n_tiles = (n + EXP_ROUND - 1) / EXP_ROUND;
	const bool ring = n_tiles >= 3000 && n_tiles < 24000;
	unsigned grid;
	if (!ring) grid = grid_for(n_tiles, EXP_WAVES, 1u << 22);
	else if (n_tiles >= 12000) grid = (unsigned)EXP_GRID_PERSISTENT;
	else grid = (unsigned)(n_tiles / 8);
	#define RK_GO(FLAGS, DEPTH) hipLaunchKernelGGL((k_expand12r<FLAGS, DEPTH, true, false>), dim3(grid), dim3(EXP_WAVES * WAVE), 0, st, \
		(const uint32_t *)parents, (u32x4 *)children, (uint32_t *)(FLAGS ? solved : nullptr), FLAGS ? stats : (long long *)nullptr, n)
	if (solved != nullptr) { if (ring) RK_GO(true, 2); else RK_GO(true, 0); }
	else                   { if (ring) RK_GO(false, 2); else RK_GO(false, 0); }
	#undef RK_GO
}

void launch_expand12(const int8_t *parents, int8_t *children, uint8_t *solved, long long *stats, size_t n, hipStream_t st)
{
	const size_t n_tiles = (n + EXP_ROUND - 1) / EXP_ROUND;
	const PaceConfig &pc = pace_config();
	if (n_tiles >= pc.min_tiles && pace_on(pc)) {
		const unsigned tau = fanout_tau(pc);
		if (tau != 0) { launch_expand12_paced(parents, children, solved, stats, n, pc, tau, st); return; }
	}
	launch_expand12_unpaced(parents, children, solved, stats, n, st);
}

// ================================================================================================================
// Calibration of the store schedule.  tau = 2.10 ns per tile was tuned on this pool with a 2.4 % margin (2.05 ns:
// no box keeps the schedule, 0.79 of peak instead of 0.83).  A part whose HBM absorbs a little less -- another memory clock, a
// partitioned mode -- would fall off the schedule with no remedy but an environment variable.  So the schedule is MEASURED once
// per device and process: 1 Mi parents of scratch (272 MB, freed again), the fan-out in the ring form and at 2.0 / 2.1 / 2.2 /
// 2.4 ns per tile, four rounds of four back-to-back launches each on a stream of its own, best round per candidate.  The compiled
// schedule is KEPT unless another candidate beats it by more than 3 % (run-to-run noise of these launches is about 1 %): the records
// of profiles/ stay comparable on the boxes they were taken on, and a box where the schedule does not hold gets the candidate that
// does -- the ring form included.  About 4 ms.  Skipped (source 2) when RK_PACE_TAU_PS fixes the schedule, with RK_PACE=0 or
// RK_PACE_CALIBRATE=0, and when the scratch cannot be allocated.  Results never depend on any of it.
// ================================================================================================================
int calibrate_pacing(bool force)
{
	PaceDevice *d = pace_device();
	if (d == nullptr) return -1;
	std::lock_guard<std::mutex> lk(d->cal_mu);
	if (d->source.load() != 0 && !force) return 0;
	const PaceConfig &pc = pace_config();
	auto settle = [&](int source, unsigned tau) { d->tau_ps.store(tau); d->source.store(source); return 0; };
	if (!pc.on || pc.tau_from_env || !pc.calibrate || pace_cells(d) == nullptr) return settle(2, pc.tau_ps);
	constexpr size_t N = (size_t)1 << 20;
	constexpr size_t B_PARENTS = N * STATE_BYTES, B_CHILDREN = 12 * N * STATE_BYTES, B_FLAGS = 12 * N;
	char *buf = nullptr;
	hipStream_t st = nullptr;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	bool ok = hipMalloc((void **)&buf, B_PARENTS + B_CHILDREN + B_FLAGS + 256) == hipSuccess
	       && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess
	       && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
	float best[PACE_CANDIDATES];
	for (float &b : best) b = 1e30f;
	if (ok) {
		int8_t *parents = (int8_t *)buf, *children = (int8_t *)(buf + B_PARENTS);
		uint8_t *flags = (uint8_t *)(buf + B_PARENTS + B_CHILDREN);
		long long *stats = (long long *)(buf + B_PARENTS + B_CHILDREN + B_FLAGS);
		ok = hipMemsetAsync(buf, 0, B_PARENTS, st) == hipSuccess && hipMemsetAsync(stats, 0, 16, st) == hipSuccess;   // code 0 everywhere: valid cubies
		// (the instantiations WITHOUT flags: the same store stream less 5 %, its schedule scaled by the tile's bytes as everywhere -- and
		//  kernel names of their own, so that a profile of the caller's process averages the caller's launches only)
		(void)flags; (void)stats;
		auto go = [&](unsigned tau) {
			if (tau == 0) launch_expand12_unpaced(parents, children, nullptr, nullptr, N, st);
			else launch_expand12_paced(parents, children, nullptr, nullptr, N, pc, tau, st, false);
		};
		for (int round = 0; ok && round < 5; round++)                      // round 0 warms every candidate's code and the clocks up
			for (int c = 0; ok && c < PACE_CANDIDATES; c++) {
				ok = hipEventRecord(e0, st) == hipSuccess;
				for (int k = 0; k < 4; k++) go(PACE_CANDIDATE_TAU_PS[c]);
				ok = ok && hipGetLastError() == hipSuccess && hipEventRecord(e1, st) == hipSuccess && hipEventSynchronize(e1) == hipSuccess;
				float ms = 0;
				ok = ok && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
				if (ok && round > 0 && ms * 250.0f < best[c]) best[c] = ms * 250.0f;      // us per launch
			}
	}
	if (e0) (void)hipEventDestroy(e0);
	if (e1) (void)hipEventDestroy(e1);
	if (st) (void)hipStreamDestroy(st);
	if (buf) (void)hipFree(buf);
	if (!ok) { (void)hipGetLastError(); return settle(2, pc.tau_ps); }
	int dflt = -1, win = 0;
	for (int c = 0; c < PACE_CANDIDATES; c++) {
		d->us[c] = best[c];
		if (PACE_CANDIDATE_TAU_PS[c] == pc.tau_ps) dflt = c;
		if (best[c] < best[win]) win = c;
	}
	if (dflt >= 0 && best[dflt] <= 1.03f * best[win]) win = dflt;
	return settle(1, PACE_CANDIDATE_TAU_PS[win]);
}

// tau of the fan-out on the current device, where it came from (0 default / 1 calibrated / 2 environment), the measured us per
// 1 Mi-parent launch of the candidates (ring form, 2.0, 2.1, 2.2, 2.4 ns; zeros if nothing was measured)
void get_pacing(unsigned *tau_ps, int *source, float *us)
{
	const PaceConfig &pc = pace_config();
	PaceDevice *d = pace_device();
	const int src = d ? d->source.load() : 0;
	if (tau_ps) *tau_ps = !pc.on ? 0u : (d && src == 1 ? d->tau_ps.load() : pc.tau_ps);
	if (source) *source = src;
	if (us) for (int c = 0; c < PACE_CANDIDATES; c++) us[c] = d ? d->us[c] : 0.0f;
}

void launch_expand12_soa(const uint32_t *parents, uint32_t *children, uint8_t *solved, long long *stats, size_t n, hipStream_t st)
{
	const unsigned grid = grid_for(n, 256, 1u << 20);
	if (solved != nullptr)
		hipLaunchKernelGGL(k_expand12_soa<true>, dim3(grid), dim3(256), 0, st, parents, children, solved, stats, n);
	else
		hipLaunchKernelGGL(k_expand12_soa<false>, dim3(grid), dim3(256), 0, st, parents, children, (uint8_t *)nullptr, (long long *)nullptr, n);
}

void launch_states_soa(const int8_t *states, uint32_t *planes, size_t n, bool to_soa, hipStream_t st)
{
	const unsigned grid = grid_for(n * 5, 256, 256u * 16u);
	hipLaunchKernelGGL(k_states_to_soa, dim3(grid), dim3(256), 0, st, (const uint32_t *)states, planes, n, to_soa ? 1 : 0);
}

// Workgroups of the per-row kernels (multi_rotate, multi_is_solved): one 256-state tile per wave up to this many workgroups,
// a persistent grid whose waves request their next tile ahead beyond it.  RK_ROW_GRID overrides (tuning).
static unsigned row_grid_cap()
{
	static const unsigned cap = [] { const char *e = std::getenv("RK_ROW_GRID"); return e ? (unsigned)std::atoi(e) : (1u << 22); }();
	return cap > 0 ? cap : (1u << 22);
}

// Paced per-row launches (from 8 192 tiles = 2 Mi states on, one tile per wave): RK_PACE_ROT_TAU_PS / RK_PACE_ROT_NT for multi_rotate,
// RK_PACE_SOLVED_TAU_PS for multi_is_solved override the constants (0 = unpaced).
static unsigned env_u(const char *name, unsigned dflt) { const char *e = std::getenv(name); return e ? (unsigned)std::atol(e) : dflt; }

void launch_multi_rotate(const int8_t *states, const uint8_t *actions, const uint8_t *dirs, int8_t *out, size_t n, hipStream_t st,
                         uint8_t *flags, long long *stats, bool with_flags)
{
	static const unsigned tau_cfg = env_u("RK_PACE_ROT_TAU_PS", PACE_ROT_TAU_PS), nt_cfg = env_u("RK_PACE_ROT_NT", 1);
	const size_t n_tiles = (n + ROW_TILE - 1) / ROW_TILE;
	const unsigned grid = grid_for(n_tiles, ROW_WAVES, row_grid_cap());
	const PaceConfig &pc = pace_config();
	const bool paced = pace_on(pc) && tau_cfg > 0 && n_tiles >= 8192 && (size_t)grid * ROW_WAVES >= n_tiles;
	const unsigned tau = paced ? tau_cfg : 0u, nt = paced ? nt_cfg : 0u;
	unsigned long long *const cell = paced ? next_pace_cell() : nullptr;
	PacedTurn turn(st, paced);
	if (with_flags)
		hipLaunchKernelGGL((k_multi_rotate<false, true>), dim3(grid), dim3(ROW_WAVES * WAVE), 0, st, (const uint32_t *)states, actions,
		                   (const uint8_t *)nullptr, (uint32_t *)out, n, n_tiles, tau, pc.lead, nt, cell, flags, stats);
	else if (dirs != nullptr)
		hipLaunchKernelGGL((k_multi_rotate<true, false>), dim3(grid), dim3(ROW_WAVES * WAVE), 0, st, (const uint32_t *)states, actions, dirs,
		                   (uint32_t *)out, n, n_tiles, tau, pc.lead, nt, cell, (uint8_t *)nullptr, (long long *)nullptr);
	else
		hipLaunchKernelGGL((k_multi_rotate<false, false>), dim3(grid), dim3(ROW_WAVES * WAVE), 0, st, (const uint32_t *)states, actions,
		                   (const uint8_t *)nullptr, (uint32_t *)out, n, n_tiles, tau, pc.lead, nt, cell, (uint8_t *)nullptr, (long long *)nullptr);
}

// reads and clears the mark bad action codes leave (synchronises `st`); negative on a HIP error.  Read and clear are ONE
// atomic exchange on the device: a kernel on another stream that sets the mark meanwhile is either seen now or stays for the next
// call, never lost between a read and a separate clear.  The mark is per device, not per stream.
__global__ void k_take_bad_actions(unsigned *out) { if (threadIdx.x == 0 && blockIdx.x == 0) *out = atomicExch(&g_bad_actions, 0u); }

int read_bad_actions(hipStream_t st)
{
	unsigned *cell = nullptr;
	if (hipHostMalloc((void **)&cell, sizeof *cell, hipHostMallocDefault) != hipSuccess) return -1;
	*cell = 0;
	hipLaunchKernelGGL(k_take_bad_actions, dim3(1), dim3(64), 0, st, cell);
	const bool ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
	const unsigned h = *cell;
	(void)hipHostFree(cell);
	if (!ok) return -1;
	return h != 0 ? 1 : 0;
}

void launch_multi_is_solved(const int8_t *states, uint8_t *flags, long long *stats, size_t n, hipStream_t st)
{
	static const unsigned tau_cfg = env_u("RK_PACE_SOLVED_TAU_PS", PACE_SOLVED_TAU_PS);
	const size_t n_tiles = (n + ROW_TILE - 1) / ROW_TILE;
	const unsigned grid = grid_for(n_tiles, ROW_WAVES, row_grid_cap());
	const PaceConfig &pc = pace_config();
	const bool paced = pace_on(pc) && tau_cfg > 0 && n_tiles >= 8192 && (size_t)grid * ROW_WAVES >= n_tiles;
	PacedTurn turn(st, paced);
	hipLaunchKernelGGL(k_multi_is_solved, dim3(grid), dim3(ROW_WAVES * WAVE), 0, st, (const uint32_t *)states, flags, stats, n, n_tiles,
	                   paced ? tau_cfg : 0u, pc.lead, paced ? next_pace_cell() : (unsigned long long *)nullptr);
}

void launch_rollout_fanout(const uint8_t *actions, int moves, int games, int with_solved, int8_t *states, uint8_t *state_flags, int8_t *children,
                           uint8_t *child_flags, long long *stats, hipStream_t st)
{
	const size_t rows = (size_t)(moves + (with_solved ? 1 : 0)), n = (size_t)games * rows;
	const size_t tile_states = rows <= (size_t)EXP_ROUND ? (EXP_ROUND / rows) * rows : (size_t)EXP_ROUND;      // whole games per wave (the kernel's own rule)
	const unsigned grid = grid_for((n + tile_states - 1) / tile_states, EXP_WAVES, 1u << 22);
	hipLaunchKernelGGL(k_rollout_fanout, dim3(grid), dim3(EXP_WAVES * WAVE), 0, st, actions, moves, games, with_solved, (uint32_t *)states, state_flags,
	                   (u32x4 *)children, (uint32_t *)child_flags, stats);
}

void launch_apply_sequences(const uint8_t *actions, int moves, int games, int with_solved, int only_last, int8_t *out, hipStream_t st)
{
	// FEW games and more than a handful of moves: a wave per game and a scan over its moves -- the scan does more arithmetic (log2(64) x
	// twelve lut4 per move where a walking lane does five) to shorten the dependent chain from `moves` steps to six per 64 moves, which
	// pays while the waves do not compete for the SIMDs: 1 x 999 moves 44 us against about 450, 96 x 100 12 us against 45; at 7 500 x 30
	// the walk's 14 us beat the scan's 41 (rocprofv3, round 5).  Many games: a lane per game walking its moves.
	if (games <= 1024 && moves >= 8) {
		hipLaunchKernelGGL(k_apply_sequences_scan, dim3(grid_for((size_t)games, 4, 1u << 22)), dim3(256), 0, st, actions, moves, games, with_solved, only_last, (uint32_t *)out);
		return;
	}
	const unsigned grid = grid_for((size_t)games, 256, 1u << 22);
	hipLaunchKernelGGL(k_apply_sequences, dim3(grid), dim3(256), 0, st, actions, moves, games, with_solved, only_last, (uint32_t *)out);
}

// workgroups of a paced launch: per phase `pull` readers and one workgroup per tile
static unsigned oh_paced_grid(size_t n_tiles, unsigned pull, unsigned phase_tiles)
{
	const size_t n_phases = (n_tiles + phase_tiles - 1) / phase_tiles;
	return (unsigned)((n_phases - 1) * (size_t)(pull + phase_tiles) + pull + (n_tiles - (n_phases - 1) * phase_tiles));
}


// Shipping tiling (benchmarks/tune_oh.py, outputs rotated so that nothing is rewritten in cache): one workgroup per tile
// and about four 16-byte stores per thread -- 8 states per workgroup for f32, 16 for the 16-bit types: 5.6 / 6.1 TB/s
// against 5.3 / 5.0 TB/s for 64-state tiles on a persistent grid (the same "few stores per wave" effect as in the fan-out).
void launch_as_oh(const int8_t *states, void *out, int out_dtype, size_t n, hipStream_t st)
{
	// From 8 192 tiles on (a launch of >= 17 us) the stores are non-temporal and released on the fan-out's schedule (k_expand12p; a tile
	// is 15 360 B here, so tau scales by 15 360 / 16 128): 0.76 -> 0.87 (f32) and 0.78 -> 0.88-0.90 (bf16) of peak at 500 k states,
	// profiles/r03_oh_pace.json.  Unpaced, non-temporal stores lose to plain ones (0.72 / 0.70), and plain ones gain nothing from pacing.
	const PaceConfig &pc = pace_config();
	const size_t n_tiles = out_dtype == 0 ? (n + 7) / 8 : (n + 15) / 16;
	const bool paced = pace_on(pc) && n_tiles >= 8192;
	const unsigned tau = paced ? stream_tau(pc) * 15360u / 16128u : 0u;
	const unsigned phase_tiles = (unsigned)(((size_t)pc.phase_tiles * EXP_ROUND) / (out_dtype == 0 ? 8 : 16));    // the fan-out's phase in states: 1 Mi
	const unsigned grid = paced ? oh_paced_grid(n_tiles, pc.pull_wgs, phase_tiles) : grid_for(n_tiles, 1, 1u << 22);
	PacedTurn turn(st, paced);
	#define RK_OH(T, EB, TL) do { \
		if (paced) hipLaunchKernelGGL((k_as_oh<T, EB, TL, true>), dim3(grid), dim3(256), 0, st, (const uint32_t *)states, (u32x4 *)out, n, n_tiles, tau, pc.lead, pc.pull_wgs, phase_tiles, next_pace_cell()); \
		else hipLaunchKernelGGL((k_as_oh<T, EB, TL, false>), dim3(grid), dim3(256), 0, st, (const uint32_t *)states, (u32x4 *)out, n, n_tiles, 0u, 0u, 0u, 0u, (unsigned long long *)nullptr); } while (0)
	if (out_dtype == 0) RK_OH(float, 4, 8);
	else if (out_dtype == 1) RK_OH(_Float16, 2, 16);
	else RK_OH(bf16_tag, 2, 16);
	#undef RK_OH
}

void launch_rotate686(const int8_t *states, const uint8_t *actions, int8_t *out, size_t n_out, bool fanout, hipStream_t st,
                      uint8_t *flags, long long *stats)
{
	if (fanout) {
		const size_t n_in = n_out / 12;
		// Paced form (k_fanout686p, eight parents per workgroup) from 65 536 parents on; phases of 65 536 parents = 18.9 MB:
		// 0.71 -> 0.79-0.80 of peak cache-neutral at 200 k parents (four parents per workgroup 0.68, sixteen 0.74-0.76; profiles/r03_paced686.json).
		{
			static const unsigned min_parents = env_u("RK_PACE686_MIN", PACE686_MIN_PARENTS);
			const PaceConfig &pc = pace_config();
			if (pace_on(pc) && n_in >= min_parents) {
				constexpr unsigned G = 8;
				const size_t n_groups = (n_in + G - 1) / G;
				const unsigned phase_groups = (unsigned)(((size_t)pc.phase_tiles * 4) / G);
				const unsigned tau = (unsigned)((unsigned long long)stream_tau(pc) * (G * 12 * S686_BYTES + G * 12) / 16128u);
				const size_t n_phases = (n_groups + phase_groups - 1) / phase_groups;
				const unsigned grid = (unsigned)((n_phases - 1) * (size_t)(pc.pull_wgs + phase_groups) + pc.pull_wgs + (n_groups - (n_phases - 1) * phase_groups));
				PacedTurn turn(st, true);
				if (flags != nullptr || stats != nullptr)
					hipLaunchKernelGGL((k_fanout686p<true, G>), dim3(grid), dim3(256), 0, st, (const uint16_t *)states, (u32x4 *)out, n_in, flags, stats,
					                   pc.pull_wgs, phase_groups, tau, pc.lead, next_pace_cell());
				else
					hipLaunchKernelGGL((k_fanout686p<false, G>), dim3(grid), dim3(256), 0, st, (const uint16_t *)states, (u32x4 *)out, n_in, (uint8_t *)nullptr,
					                   (long long *)nullptr, pc.pull_wgs, phase_groups, tau, pc.lead, next_pace_cell());
				return;
			}
		}
		const unsigned grid = grid_for(n_in, 4, 8192u);               // persistent from 32 k parents on: the next group's states are in flight
		if (flags != nullptr || stats != nullptr)          // children and their solved flags in ONE launch
			hipLaunchKernelGGL((k_rotate686<true, true>), dim3(grid), dim3(256), 0, st, (const uint16_t *)states, actions, (u32x4 *)out, n_in, flags, stats);
		else
			hipLaunchKernelGGL((k_rotate686<true, false>), dim3(grid), dim3(256), 0, st, (const uint16_t *)states, actions, (u32x4 *)out, n_in,
			                   (uint8_t *)nullptr, (long long *)nullptr);
	} else {
		const unsigned grid = grid_for(n_out, 64, 2048u);
		hipLaunchKernelGGL((k_rotate686<false, false>), dim3(grid), dim3(256), 0, st, (const uint16_t *)states, actions, (u32x4 *)out, n_out,
		                   (uint8_t *)nullptr, (long long *)nullptr);
	}
}

void launch_is_solved686(const int8_t *states, uint8_t *flags, long long *stats, size_t n, hipStream_t st)
{
	const unsigned grid = grid_for(n, 128, 1u << 20);
	hipLaunchKernelGGL(k_is_solved686, dim3(grid), dim3(256), 0, st, (const u32x4 *)states, flags, stats, n);
}

void launch_as_oh686(const int8_t *states, void *out, int out_dtype, size_t n, hipStream_t st)
{
	const size_t ndw = n * 72;
	const unsigned grid = grid_for(ndw, 256, 256u * 8u);
	if (out_dtype == 0)
		hipLaunchKernelGGL((k_as_oh686<float, 4>), dim3(grid), dim3(256), 0, st, (const uint32_t *)states, (u32x4 *)out, ndw);
	else if (out_dtype == 1)
		hipLaunchKernelGGL((k_as_oh686<_Float16, 2>), dim3(grid), dim3(256), 0, st, (const uint32_t *)states, (u32x4 *)out, ndw);
	else
		hipLaunchKernelGGL((k_as_oh686<bf16_tag, 2>), dim3(grid), dim3(256), 0, st, (const uint32_t *)states, (u32x4 *)out, ndw);
}

void launch_as_correct686(const int8_t *states, float *out, size_t n, hipStream_t st)
{
	const unsigned grid = grid_for(n * 48, 256, 256u * 8u);
	hipLaunchKernelGGL(k_as_correct686, dim3(grid), dim3(256), 0, st, states, out, n * 48);
}

const Tables &host_tables()
{
	static const Tables t = make_tables();
	return t;
}

}  // namespace rk
