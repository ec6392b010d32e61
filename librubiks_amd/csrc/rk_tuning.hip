// Tuning build only (python -m librubiks_amd.build --tune -> benchmarks/librubiks_hip_tune.so; never part of librubiks_hip.so):
// the kernel-shape variants and memory-geometry diagnostics that benchmarks/tune_*.py time against the shipping kernels.  They
// are built from the shipping kernels' own device functions, which are file-local, so this translation unit INCLUDES
// rk_cube_kernels.hip (the tuning build compiles this file instead of that one) and adds to it:
//   1. the rounds 1-2 fan-out kernel k_expand12 (A/B reference of the ring form),
//   2. the geometry / store-stream diagnostics and launch_expand12_variant,
//   3. launch_as_oh_variant.
// The entry points rkx_* that reach them are in rk_api.hip under the same macro.
// Without -DRK_TUNING this file is an EMPTY translation unit (a plain `hipcc -c csrc/*.hip` of the whole directory still works and
// links); with it, it is compiled IN THE PLACE of rk_cube_kernels.hip (python -m librubiks_amd.build --tune), never beside it.
#ifdef RK_TUNING
#include "rk_cube_kernels.hip"

namespace rk {

// ================================================================================================================
// 1. rounds 1-2 kernel, kept as the A/B reference of the ring form
// ================================================================================================================
// The kernel's shape is a set of compile-time knobs; benchmarks/tune_expand.py A/Bs them (profiles/r01_tune_expand*.json,
// profiles/r02_tune_expand.json).  Two shapes ship (launch_expand12): <ROUNDS 1, NT, 4 waves> with one tile per wave for
// small batches, and the same with PRELOAD on a persistent grid of 3 072 workgroups from about half a million parents on --
// with inputs that really come from HBM (round 2 measures cache-neutral) a wave that only ever sees one tile waits a full
// memory latency for it; issuing the next tile's loads before expanding the current one takes 1 M parents from 50.0 to
// 43.5 us.  Shapes that were tried and dropped from the code because they lost or tied: an atomic tile counter (2-5x
// slower), per-lane strided input loads instead of the LDS transpose (3 % slower), and a "split" shape where a lane owns
// (parent, four children) and a wave writes only 3 840 B at 32 waves/CU (45.5 us vs 44.6 us).  The geometry-only
// diagnostics (tuning build) show why shapes stop mattering: the same loads and stores WITHOUT any table look-up,
// transpose or LDS traffic take within 1 % of the real kernel -- it is bound by its memory access pattern (7 % reads,
// three streams).
// ROUNDS = rounds of 64 parents per wave tile (4 -> 256-parent tiles with 16 B/lane input loads, 1 -> 64-parent tiles);
// NT = non-temporal output stores; NWAVES = waves per workgroup;
// PRELOAD = software pipeline of the input: a tile's parent loads are issued one tile ahead (the first before the move
// table is staged), so a wave that walks several tiles (persistent grid) never waits a full HBM latency per tile.
// HALVES = 2 stages and streams a round's children in two halves of 32 parents: half the LDS per wave (8.4 KB), which lets
// twice as many waves live on a CU (LDS, not registers, caps the occupancy of this kernel) at the price of 128 VGPRs.
template <bool WITH_FLAGS, int ROUNDS = 1, bool NT = true, int NWAVES = EXP_WAVES, bool PRELOAD = false, int HALVES = 1>
__global__ __launch_bounds__(NWAVES * WAVE, (HALVES == 2 ? 4 : 1))
void k_expand12(const uint32_t *__restrict__ parents, u32x4 *__restrict__ children, uint32_t *__restrict__ solved,
                long long *__restrict__ stats, size_t n, size_t n_tiles)
{
	constexpr int EXP_TILE = EXP_ROUND * ROUNDS;
	constexpr int EXP_WAVES = NWAVES;
	__shared__ u32x4 s_rows[48];
	typedef ExpandWaveLdsT<HALVES> ExpandWaveLds;
	__shared__ ExpandWaveLds s_wave[EXP_WAVES];

	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	size_t tile = (size_t)blockIdx.x * EXP_WAVES + wv;
	uint32_t pre[5 * ROUNDS];
	bool have_pre = false;
	if (PRELOAD && tile < n_tiles) {
		const size_t p0 = tile * EXP_TILE;
		const int ndw = (int)((n - p0 < (size_t)EXP_TILE) ? (n - p0) : (size_t)EXP_TILE) * STATE_DWORDS;
		const uint32_t *src = parents + p0 * STATE_DWORDS;
		#pragma unroll
		for (int k = 0; k < 5 * ROUNDS; k++) {
			const int idx = k * 64 + lane;
			pre[k] = idx < ndw ? src[idx] : 0u;
		}
		have_pre = true;
	}
	if (tid < 48) {
		const uint32_t *src = reinterpret_cast<const uint32_t *>(D_TAB.rows) + 4 * tid;
		s_rows[tid] = u32x4{src[0], src[1], src[2], src[3]};
	}
	__syncthreads();

	ExpandWaveLds &L = s_wave[wv];
	uint32_t *stage_dw = reinterpret_cast<uint32_t *>(L.stage);

	for (; tile < n_tiles; tile += (size_t)gridDim.x * EXP_WAVES) {
		const size_t p0 = tile * EXP_TILE;
		const int np = (int)((n - p0 < (size_t)EXP_TILE) ? (n - p0) : (size_t)EXP_TILE);   // parents in this tile

		// ---- parents in: coalesced (16 B/lane when the tile is 256 parents), through LDS, one state per lane per round ----
		uint32_t par[ROUNDS][5];
		{
			const uint32_t *src = parents + p0 * STATE_DWORDS;
			if (PRELOAD && have_pre) {
				#pragma unroll
				for (int k = 0; k < 5 * ROUNDS; k++) stage_dw[k * 64 + lane] = pre[k];
				// issue the NEXT tile's loads now; they land while this tile is being expanded
				const size_t nxt = tile + (size_t)gridDim.x * EXP_WAVES;
				have_pre = nxt < n_tiles;
				if (have_pre) {
					const size_t q0 = nxt * EXP_TILE;
					const int ndw2 = (int)((n - q0 < (size_t)EXP_TILE) ? (n - q0) : (size_t)EXP_TILE) * STATE_DWORDS;
					const uint32_t *src2 = parents + q0 * STATE_DWORDS;
					#pragma unroll
					for (int k = 0; k < 5 * ROUNDS; k++) {
						const int idx = k * 64 + lane;
						pre[k] = idx < ndw2 ? src2[idx] : 0u;
					}
				}
			} else if (ROUNDS == 4 && np == EXP_TILE && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
				const u32x4 *src4 = reinterpret_cast<const u32x4 *>(src);
				#pragma unroll
				for (int k = 0; k < 5; k++) L.stage[k * 64 + lane] = src4[k * 64 + lane];
			} else {
				const int ndw = np * STATE_DWORDS;
				#pragma unroll
				for (int k = 0; k < 5 * ROUNDS; k++) {
					const int idx = k * 64 + lane;
					stage_dw[idx] = idx < ndw ? src[idx] : 0u;
				}
			}
			wave_lds_fence();
			#pragma unroll
			for (int q = 0; q < ROUNDS; q++)
				#pragma unroll
				for (int j = 0; j < 5; j++) par[q][j] = stage_dw[(q * 64 + lane) * 5 + j];
			wave_lds_fence();
		}

		#pragma unroll
		for (int q = 0; q < ROUNDS; q++) {                                  // fully unrolled: par[q] stays in registers
			const int round_first = q * EXP_ROUND;
			if (round_first >= np) break;                                   // wave-uniform
			const int nr = (np - round_first < EXP_ROUND) ? (np - round_first) : EXP_ROUND;

			uint32_t out[60];                                               // out[a*5 + j] = dword j of child a
			#pragma unroll
			for (int j = 0; j < 5; j++) {
				const uint32_t x = par[q][j];
				const int kind_base = (j < 2) ? 0 : 24;
				const u32x4 r0 = s_rows[kind_base + (x & 0xFF)];
				const u32x4 r1 = s_rows[kind_base + ((x >> 8) & 0xFF)];
				const u32x4 r2 = s_rows[kind_base + ((x >> 16) & 0xFF)];
				const u32x4 r3 = s_rows[kind_base + (x >> 24)];
				transpose4x4(r0.x, r1.x, r2.x, r3.x, out[0 * 5 + j], out[1 * 5 + j], out[2 * 5 + j], out[3 * 5 + j]);
				transpose4x4(r0.y, r1.y, r2.y, r3.y, out[4 * 5 + j], out[5 * 5 + j], out[6 * 5 + j], out[7 * 5 + j]);
				transpose4x4(r0.z, r1.z, r2.z, r3.z, out[8 * 5 + j], out[9 * 5 + j], out[10 * 5 + j], out[11 * 5 + j]);
			}

			// ---- goal test of the 12 children ----
			uint32_t fl[3] = {0u, 0u, 0u};
			if (WITH_FLAGS) {
				#pragma unroll
				for (int a = 0; a < 12; a++)
					if (is_solved5(&out[a * 5])) fl[a >> 2] |= 1u << (8 * (a & 3));
			}

			// ---- children out: lane-major 240 B blocks -> wave-contiguous 1 KiB stores (in HALVES passes) ----
			if (WITH_FLAGS) {
				L.flags[lane * 3 + 0] = fl[0];
				L.flags[lane * 3 + 1] = fl[1];
				L.flags[lane * 3 + 2] = fl[2];
			}
			#pragma unroll
			for (int h = 0; h < HALVES; h++) {
				constexpr int LANES = EXP_ROUND / HALVES;               // parents per pass
				constexpr int NVEC = LANES * 15;                        // 16-byte chunks per pass
				if (HALVES == 1 || lane / LANES == h) {
					const int l = lane % LANES;
					#pragma unroll
					for (int v = 0; v < 15; v++)
						L.stage[l * 15 + v] = u32x4{out[4 * v], out[4 * v + 1], out[4 * v + 2], out[4 * v + 3]};
				}
				wave_lds_fence();
				u32x4 *dst = children + (p0 + round_first + h * LANES) * 15;
				int valid = nr - h * LANES;
				valid = valid < 0 ? 0 : (valid > LANES ? LANES : valid);
				const int nvec = valid * 15;
				#pragma unroll
				for (int v = 0; v < (NVEC + 63) / 64; v++) {
					const int idx = v * 64 + lane;
					if (idx < NVEC) {
						const u32x4 val = L.stage[idx];
						if (idx < nvec) {
							if (NT) __builtin_nontemporal_store(val, dst + idx);
							else dst[idx] = val;
						}
					}
				}
				if (HALVES > 1) wave_lds_fence();
			}
			if (WITH_FLAGS) {
				uint32_t *fdst = solved + (p0 + round_first) * 3;
				if (nr == EXP_ROUND && ((reinterpret_cast<uintptr_t>(fdst) & 15) == 0)) {
					if (lane < 48) {
						const u32x4 val = reinterpret_cast<const u32x4 *>(L.flags)[lane];
						if (NT) __builtin_nontemporal_store(val, reinterpret_cast<u32x4 *>(fdst) + lane);
						else reinterpret_cast<u32x4 *>(fdst)[lane] = val;
					}
				} else {
					#pragma unroll
					for (int k = 0; k < 3; k++) {
						const int idx = k * 64 + lane;
						if (idx < nr * 3) fdst[idx] = L.flags[idx];
					}
				}
				// solved children are rare: one ballot decides whether anybody reports
				const bool any = (fl[0] | fl[1] | fl[2]) != 0u && lane < nr;
				if (stats != nullptr && __ballot(any) != 0ull && any) {
					const int cnt = __popc(fl[0]) + __popc(fl[1]) + __popc(fl[2]);
					int first = 0;
					#pragma unroll
					for (int a = 11; a >= 0; a--)
						if (fl[a >> 2] & (1u << (8 * (a & 3)))) first = a;
					atomicAdd(reinterpret_cast<unsigned long long *>(&stats[0]), (unsigned long long)cnt);
					atomicMin(&stats[1], (long long)((p0 + round_first + lane) * 12 + first));
				}
			}
			wave_lds_fence();
		}
	}
}

// ================================================================================================================
// 2. tuning aids: geometry diagnostics, store-stream experiments, kernel-shape variants
// ================================================================================================================
// Diagnostic only (never used by the product): the fan-out kernel's memory geometry without its work.  Every wave
// reads its tile's 1 280 B, then writes 15 KiB + 768 B of junk derived from it with the same store instructions.
//   mode 0: straight from registers (no LDS staging)     mode 1: through the LDS staging round trip
//   mode 2: no parent loads (stores only)                mode 3: no parent loads and no flag stream (one pure store stream)
// Timing it against the real kernel separates "the store pattern" from "the table look-ups and transposes".
template <int MODE, bool NT>
__global__ __launch_bounds__(EXP_WAVES * WAVE)
void k_expand12_geometry(const uint32_t *__restrict__ parents, u32x4 *__restrict__ children, uint32_t *__restrict__ solved, size_t n_tiles)
{
	__shared__ ExpandWaveLdsT<1> s_wave[EXP_WAVES];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	ExpandWaveLdsT<1> &L = s_wave[wv];
	for (size_t tile = (size_t)blockIdx.x * EXP_WAVES + wv; tile < n_tiles; tile += (size_t)gridDim.x * EXP_WAVES) {
		const uint32_t *src = parents + tile * 64 * STATE_DWORDS;
		uint32_t x = (uint32_t)tile;
		if (MODE < 2) {
			#pragma unroll
			for (int k = 0; k < 5; k++) x ^= src[k * 64 + lane];
		}
		u32x4 *dst = children + tile * 64 * 15;
		if (MODE == 1) {
			#pragma unroll
			for (int v = 0; v < 15; v++) L.stage[lane * 15 + v] = u32x4{x, x + v, x ^ v, x};
			wave_lds_fence();
		}
		#pragma unroll
		for (int v = 0; v < 15; v++) {
			const u32x4 val = MODE == 1 ? L.stage[v * 64 + lane] : u32x4{x, x + v, x ^ v, x};
			if (NT) __builtin_nontemporal_store(val, dst + v * 64 + lane);
			else dst[v * 64 + lane] = val;
		}
		if (MODE != 3 && lane < 48) {
			const u32x4 val = u32x4{x, x, x, x};
			if (NT) __builtin_nontemporal_store(val, reinterpret_cast<u32x4 *>(solved + tile * 192) + lane);
			else reinterpret_cast<u32x4 *>(solved + tile * 192)[lane] = val;
		}
		if (MODE == 1) wave_lds_fence();
	}
}

// Diagnostic only: the memory geometry of a "one 1 KiB store per wave" shape.  A 16-wave workgroup owns 64 parents:
// waves 0..14 each read the parent dwords their 1 KiB chunk depends on (four cached 4-byte loads per lane) and issue ONE
// 16 B/lane store; wave 15 reads the 64 parents and writes their 768 flag bytes.  Junk data, real addresses.
template <bool NT>
__global__ __launch_bounds__(1024)
void k_expand12_geometry_chunk(const uint32_t *__restrict__ parents, u32x4 *__restrict__ children, uint32_t *__restrict__ solved, size_t n_groups)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const size_t g = blockIdx.x;
	if (g >= n_groups) return;
	const uint32_t *src = parents + g * 64 * STATE_DWORDS;
	if (wv < 15) {
		uint32_t v[4];
		#pragma unroll
		for (int i = 0; i < 4; i++) {
			const int D = wv * 256 + lane * 4 + i;          // dword of the group's 3 840-dword output
			const int child = D / 5, j = D - child * 5;
			const int parent = child / 12;
			v[i] = src[parent * 5 + j] + (uint32_t)(child - parent * 12);
		}
		const u32x4 val = u32x4{v[0], v[1], v[2], v[3]};
		u32x4 *dst = children + g * 960 + wv * 64 + lane;
		if (NT) __builtin_nontemporal_store(val, dst);
		else *dst = val;
	} else {
		uint32_t x = 0;
		#pragma unroll
		for (int j = 0; j < 5; j++) x ^= src[lane * 5 + j];
		if (lane < 48) {
			const u32x4 val = u32x4{x, x, x, x};
			u32x4 *dst = reinterpret_cast<u32x4 *>(solved + g * 192) + lane;
			if (NT) __builtin_nontemporal_store(val, dst);
			else *dst = val;
		}
	}
}

// Diagnostic only: a pure store stream of `total_kib` KiB in configurable geometry, no LDS, no loads.  Each wave writes
// CH chunks of 1 KiB, either as one contiguous run or interleaved with the other waves of its workgroup.
template <int CH, bool INTERLEAVE, bool NT>
__global__ __launch_bounds__(256)
void k_store_geometry(u32x4 *__restrict__ dst, size_t total_kib)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const size_t wave_global = (size_t)blockIdx.x * 4 + wv;
	const size_t n_waves = (total_kib + CH - 1) / CH;
	for (size_t w = wave_global; w < n_waves; w += (size_t)gridDim.x * 4) {
		const u32x4 val = u32x4{(uint32_t)w, (uint32_t)lane, 0u, 1u};
		#pragma unroll
		for (int v = 0; v < CH; v++) {
			const size_t kib = INTERLEAVE ? ((w / 4) * 4 * CH + (size_t)v * 4 + (w & 3)) : (w * CH + v);
			if (kib < total_kib) {
				if (NT) __builtin_nontemporal_store(val, dst + kib * 64 + lane);
				else dst[kib * 64 + lane] = val;
			}
		}
	}
}

// Diagnostic only: the pure store stream again, each wave's CH stores spaced by s_sleep(SLEEP) instead of back to back.
template <int CH, int SLEEP>
__global__ __launch_bounds__(256)
void k_store_geometry_paced(u32x4 *__restrict__ dst, size_t total_kib)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const size_t w = (size_t)blockIdx.x * 4 + wv;
	const u32x4 val = u32x4{(uint32_t)w, (uint32_t)lane, 0u, 1u};
	#pragma unroll
	for (int v = 0; v < CH; v++) {
		const size_t kib = w * CH + v;
		if (kib < total_kib) __builtin_nontemporal_store(val, dst + kib * 64 + lane);
		if (SLEEP > 0 && v + 1 < CH) __builtin_amdgcn_s_sleep(SLEEP);
	}
}

// Diagnostic only: a pure non-temporal store stream whose shape comes at run time: every wave writes `ch` stores of 16 B/lane to
// one contiguous span, the last of them with `last` lanes, spans packed back to back from `shift16` x 16 B into the buffer.
__global__ __launch_bounds__(256)
void k_store_geometry_rt(u32x4 *__restrict__ dst, size_t total16, int ch, int last, int shift16, int wait)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wpb = blockDim.x >> 6;
	const size_t w = (size_t)blockIdx.x * wpb + wv;
	const size_t span16 = (size_t)(ch - 1) * 64 + last;
	const size_t base = (size_t)shift16 + w * span16;
	const u32x4 val = u32x4{(uint32_t)w, (uint32_t)lane, 0u, 1u};
	for (int v = 0; v < ch; v++) {
		const size_t idx = base + (size_t)v * 64 + lane;
		if ((v + 1 < ch || lane < last) && idx < total16) __builtin_nontemporal_store(val, dst + idx);
		if (wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // one store in flight per wave
	}
}

// Diagnostic only: one 16 B/lane store per wave, a 4-wave workgroup per 4 KiB page, the page of workgroup i rotated inside its
// aligned group of `group` pages by `rot` -- does it matter WHICH workgroup (i.e. which XCD: i mod 8) writes a page?
__global__ __launch_bounds__(256)
void k_store_geometry_page(u32x4 *__restrict__ dst, size_t n_pages, int rot, int group)
{
	const size_t i = blockIdx.x;
	const size_t page = (i / group) * group + (i + rot) % group;
	if (page >= n_pages) return;
	__builtin_nontemporal_store(u32x4{(uint32_t)i, threadIdx.x, 0u, 1u}, dst + page * 256 + threadIdx.x);
}

// Diagnostic only: one 16 B/lane store per wave again, but the wave has a life before it: mode 0 every wave sleeps `sleep` x 64
// clocks first, mode 1 only wave 0 of the workgroup sleeps and the others wait for it at a barrier (a workgroup whose first
// wave computes and whose other waves only store), mode 2 every wave first waits for a global load.  blockDim = 64 x waves.
__global__ __launch_bounds__(1024)
void k_store_geometry_life(u32x4 *__restrict__ dst, const uint32_t *__restrict__ src, size_t n_kib, int sleep, int mode)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wpb = blockDim.x >> 6;
	const size_t kib = (size_t)blockIdx.x * wpb + wv;
	uint32_t x = (uint32_t)kib;
	if (mode == 0 || (mode == 1 && wv == 0)) for (int i = 0; i < sleep; i++) __builtin_amdgcn_s_sleep(1);
	if (mode == 1) __syncthreads();
	if (mode == 2) x ^= src[(kib * 5) & 0xfffff];
	if (kib < n_kib) __builtin_nontemporal_store(u32x4{x, (uint32_t)lane, 0u, 1u}, dst + kib * 64 + lane);
}

// Diagnostic only: k_store_geometry_rt with a dynamic LDS allocation whose only job is to bound the waves per CU.
__global__ __launch_bounds__(256)
void k_store_geometry_occ(u32x4 *__restrict__ dst, size_t total16, int ch, int touch)
{
	extern __shared__ uint32_t s_dyn[];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wpb = blockDim.x >> 6;
	if (touch) s_dyn[threadIdx.x] = 1u;                      // never taken: keeps the allocation referenced
	const size_t w = (size_t)blockIdx.x * wpb + wv;
	const size_t base = w * (size_t)ch * 64;
	const u32x4 val = u32x4{(uint32_t)w, (uint32_t)lane, 0u, 1u};
	for (int v = 0; v < ch; v++) {
		const size_t idx = base + (size_t)v * 64 + lane;
		if (idx < total16) __builtin_nontemporal_store(val, dst + idx);
	}
}

// Diagnostic only: 15 KiB per wave again, the waves per CU bounded by LDS, and every wave HOLDS its stores until a scheduled
// moment after its own start (constant-rate clock, 10 ns ticks): the first `resident` waves (those that start together
// when the kernel starts) are staggered by `tau_ps` per wave, every later wave waits `hold` ticks.  The resident tiles
// still span resident x 15 KiB, but the stores IN FLIGHT at any moment come from waves that started one after another.
__global__ __launch_bounds__(256)
void k_store_geometry_sched(u32x4 *__restrict__ dst, size_t total16, int ch, unsigned resident, unsigned tau_ps, unsigned hold)
{
	extern __shared__ uint32_t s_dyn[];
	const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wpb = blockDim.x >> 6;
	const size_t w = (size_t)blockIdx.x * wpb + wv;
	const uint64_t wait = w < resident ? (uint64_t)w * tau_ps / 10000u : hold;
	while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(1);
	const size_t base = w * (size_t)ch * 64;
	const u32x4 val = u32x4{(uint32_t)w, (uint32_t)lane, 0u, 1u};
	for (int v = 0; v < ch; v++) {
		const size_t idx = base + (size_t)v * 64 + lane;
		if (idx < total16) __builtin_nontemporal_store(val, dst + idx);
	}
}

// Diagnostic only: the same, on ONE schedule for the whole launch: wave w may store from t0 + lead + w x tau on, where t0 is
// the moment the launch began.  The wave of tile 0 publishes its start time in `cell`; the waves that start with it
// (w < resident) take their own start time, the later ones read the cell.  A wave behind the schedule stores at once.
// MODE 0: 15 KiB of children per wave and nothing else; MODE 1: the fan-out's three streams (parents in, children and flags out);
// MODE 2: children and flags; MODE 3: parents and children; MODE 4: parents and children, but the stores do not depend on the
// loads (their values are consumed after the stores are out): the traffic of MODE 3 without its dependency.
__device__ __forceinline__ void hold_until_slot(uint64_t start, size_t w, unsigned resident, unsigned tau_ps, unsigned lead, unsigned long long *cell)
{
	if (tau_ps == 0) return;
	uint64_t t0 = start;
	if (w >= resident) t0 = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	const uint64_t due = t0 + lead + (uint64_t)w * tau_ps / 10000u;
	if (due - start > 100000u) return;                     // more than 1 ms away (or in the past): the cell is not this launch's
	while (__builtin_amdgcn_s_memrealtime() < due) __builtin_amdgcn_s_sleep(1);
}

template <int MODE>
__global__ __launch_bounds__(256)
void k_store_geometry_slot(const uint32_t *__restrict__ parents, u32x4 *__restrict__ dst, uint32_t *__restrict__ solved, size_t n_tiles,
                           unsigned resident, unsigned tau_ps, unsigned lead, unsigned long long *cell, unsigned rd_period, unsigned rd_window, unsigned wr_guard)
{
	extern __shared__ uint32_t s_dyn[];
	const uint64_t start = __builtin_amdgcn_s_memrealtime();
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wpb = blockDim.x >> 6;
	const size_t w = (size_t)blockIdx.x * wpb + wv;
	if (w == 0 && lane == 0) __hip_atomic_store(cell, start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	if (w >= n_tiles) return;
	uint32_t x = (uint32_t)w, y = 0;
	if (MODE == 5 && w < resident) {
		// read phase: the waves that start with the launch read ALL parents once (1 KiB pieces, wave w takes w, w + resident, ...)
		// so that every later parent load is an Infinity-Cache hit and no HBM read mixes with the store stream
		const size_t pieces = (n_tiles * 64 * STATE_DWORDS * 4 + 1023) / 1024;
		const u32x4 *src16 = reinterpret_cast<const u32x4 *>(parents);
		u32x4 acc = {0, 0, 0, 0};
		for (size_t pc = w; pc < pieces; pc += resident) {
			const size_t idx = pc * 64 + lane;
			if (idx * 16 < n_tiles * 64 * STATE_DWORDS * 4) { const u32x4 q = src16[idx]; acc.x ^= q.x; acc.y ^= q.y; acc.z ^= q.z; acc.w ^= q.w; }
		}
		y = acc.x ^ acc.y ^ acc.z ^ acc.w;
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	}
	if (MODE == 4) {
		const uint32_t *src = parents + w * 64 * STATE_DWORDS;
		#pragma unroll
		for (int k = 0; k < 5; k++) y ^= __builtin_nontemporal_load(src + k * 64 + lane);
	}
	if (MODE == 1 || MODE == 3 || MODE == 5) {
		// reads in bursts: the parent loads of all waves go out in a window of `rd_window` ticks every `rd_period` ticks
		if (rd_period > 0) while (__builtin_amdgcn_s_memrealtime() % rd_period >= rd_window) __builtin_amdgcn_s_sleep(1);
		const uint32_t *src = parents + w * 64 * STATE_DWORDS;
		#pragma unroll
		for (int k = 0; k < 5; k++) x ^= src[k * 64 + lane];
	}
	// the hold sits where the real kernel has its staged children ready: after the loads have arrived
	if (MODE == 1 || MODE == 3 || MODE == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	hold_until_slot(start, w, resident, tau_ps, lead, cell);
	// time slicing: no store goes out during the read window nor in the `wr_guard` ticks before it
	if (rd_period > 0 && wr_guard > 0)
		for (;;) { const unsigned ph = (unsigned)(__builtin_amdgcn_s_memrealtime() % rd_period); if (ph >= rd_window && ph < rd_period - wr_guard) break; __builtin_amdgcn_s_sleep(1); }
	#pragma unroll
	for (int v = 0; v < 15; v++) __builtin_nontemporal_store(u32x4{x, x + v, x ^ v, x}, dst + w * 960 + v * 64 + lane);
	if ((MODE == 1 || MODE == 2 || MODE == 5) && lane < 48) __builtin_nontemporal_store(u32x4{x, x, x, x}, reinterpret_cast<u32x4 *>(solved + w * 192) + lane);
	if ((MODE == 4 || MODE == 5) && y == 0x12345u) solved[0] = y;          // keeps the loads alive
}

// Diagnostic only: a pure READ stream, CH KiB per wave (16 B/lane loads, values thrown away), one-shot grid; with tau_ps > 0 wave w
// issues its loads not before t0 + w x tau (same clock and base protocol as k_store_geometry_slot).  Do ordered, rate-limited
// reads do for HBM what ordered, rate-limited stores do?
template <int CH>
__global__ __launch_bounds__(256)
void k_read_geometry_slot(const u32x4 *__restrict__ src, size_t total16, unsigned resident, unsigned tau_ps, unsigned long long *cell)
{
	extern __shared__ uint32_t s_dyn[];
	const uint64_t start = __builtin_amdgcn_s_memrealtime();
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wpb = blockDim.x >> 6;
	const size_t w = (size_t)blockIdx.x * wpb + wv;
	if (w == 0 && lane == 0) __hip_atomic_store(cell, start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	hold_until_slot(start, w, resident, tau_ps, 0, cell);
	u32x4 r[CH];
	#pragma unroll
	for (int v = 0; v < CH; v++) { const size_t i = (w * CH + v) * 64 + lane; r[v] = src[i < total16 ? i : total16 - 1]; }
	#pragma unroll
	for (int v = 0; v < CH; v++) asm volatile("" :: "v"(r[v].x), "v"(r[v].y), "v"(r[v].z), "v"(r[v].w) : "memory");
}

// Diagnostic only: one 4 KiB page per 4-wave workgroup, one store per wave, the pages of every aligned block of `block` pages
// visited in a scattered order (odd stride) instead of ascending: over what distance does the ORDER of the pages matter?
__global__ __launch_bounds__(256)
void k_store_geometry_scatter(u32x4 *__restrict__ dst, size_t n_pages, unsigned block, unsigned stride)
{
	const size_t i = blockIdx.x;
	const size_t page = (i / block) * block + (size_t)(((i % block) * stride) % block);
	if (page >= n_pages) return;
	__builtin_nontemporal_store(u32x4{(uint32_t)i, threadIdx.x, 0u, 1u}, dst + page * 256 + threadIdx.x);
}

// Diagnostic only: the fan-out's three streams when a WORKGROUP of W waves owns 64 parents (junk data, real addresses): the
// waves read the 1 280 B together, meet at a barrier (where the real kernel would have staged 15 KiB + 768 B in LDS), and
// every wave stores 16 / W of the sixteen 1 KiB pieces (the sixteenth is the 768 B of flags).  PERSIST: the workgroup walks
// tiles blockIdx, blockIdx + grid, ... with the next tile's parents requested before the barrier.
template <int W, bool PERSIST>
__global__ __launch_bounds__(W * 64)
void k_expand12_geometry_wg(const uint32_t *__restrict__ parents, u32x4 *__restrict__ children, uint32_t *__restrict__ solved, size_t n_tiles)
{
	__shared__ uint32_t s_x[W];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	auto request = [&](size_t tile) -> uint32_t {           // 320 dwords over the first 320 threads
		return (tile < n_tiles && threadIdx.x < 320) ? parents[tile * 320 + threadIdx.x] : 0u;
	};
	size_t tile = blockIdx.x;
	uint32_t next = request(tile);
	for (; tile < n_tiles; tile += gridDim.x) {
		uint32_t x = next;
		if (PERSIST) next = request(tile + gridDim.x);
		#pragma unroll
		for (int o = 32; o >= 1; o >>= 1) x ^= __shfl_xor(x, o);
		if (lane == 0) s_x[wv] = x;
		__syncthreads();
		x = s_x[0] ^ s_x[W > 4 ? 4 : W - 1];
		#pragma unroll
		for (int k = 0; k < 16 / W; k++) {
			const int piece = wv * (16 / W) + k;
			const u32x4 val = u32x4{x, x + piece, x ^ piece, x};
			if (piece < 15) __builtin_nontemporal_store(val, children + tile * 960 + piece * 64 + lane);
			else if (lane < 48) __builtin_nontemporal_store(val, reinterpret_cast<u32x4 *>(solved + tile * 192) + lane);
		}
		if (!PERSIST) break;
		__syncthreads();
	}
}

// Diagnostic only: the fan-out's memory geometry with P parents per wave instead of 64 (junk data, real addresses): the wave
// reads its P x 20 B, writes P x 240 B as 16 B/lane stores and P x 12 flag bytes as 4 B/lane.  MODE 0: all three streams;
// MODE 3: the children stream alone.
template <int P, int MODE>
__global__ __launch_bounds__(256)
void k_expand12_geometry_small(const uint32_t *__restrict__ parents, u32x4 *__restrict__ children, uint32_t *__restrict__ solved, size_t n_tiles)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const size_t tile = (size_t)blockIdx.x * 4 + wv;
	if (tile >= n_tiles) return;
	uint32_t x = (uint32_t)tile;
	if (MODE == 0) {
		const u32x4 *src = reinterpret_cast<const u32x4 *>(parents + tile * P * STATE_DWORDS);
		if (lane < P * 5 / 4) { const u32x4 q = src[lane]; x ^= q.x ^ q.y ^ q.z ^ q.w; }
		x ^= __shfl(x, lane % (P * 5 / 4));          // every lane waits for the loads
	}
	u32x4 *dst = children + tile * P * 15;
	#pragma unroll
	for (int v = 0; v < (P * 15 + 63) / 64; v++) {
		const int idx = v * 64 + lane;
		if (idx < P * 15) __builtin_nontemporal_store(u32x4{x, x + v, x ^ v, x}, dst + idx);
	}
	if (MODE == 0 && lane < P * 3) __builtin_nontemporal_store(x, solved + tile * P * 3 + lane);
}

// Diagnostic: read `n16` 16-byte words and discard them (pulls a buffer into the Infinity Cache with a pure read stream).
__global__ __launch_bounds__(256)
void k_touch(const u32x4 *__restrict__ src, size_t n16)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
		const u32x4 v = src[i];
		asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
	}
}

static unsigned long long *g_tune_cell = nullptr;
void tune_cell(void *p) { g_tune_cell = static_cast<unsigned long long *>(p); }
void tune_pace_debug(void *p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pace_dbg), &p, sizeof p); }

// tuning aid (benchmarks/tune_expand.py): the same kernel in its other shapes; grid_blocks > 0 makes the grid persistent.
void launch_expand12_variant(int variant, const int8_t *parents, int8_t *children, uint8_t *solved, long long *stats, size_t n,
                             int grid_blocks, hipStream_t st)
{
	#define RK_LAUNCH(R, NTS, W, PRE, HLV) do { \
		const size_t n_tiles = (n + 64 * (R) - 1) / (64 * (R)); \
		unsigned grid = grid_for(n_tiles, (W), 1u << 20); \
		if (grid_blocks > 0) grid = (unsigned)grid_blocks; \
		hipLaunchKernelGGL((k_expand12<true, R, NTS, W, PRE, HLV>), dim3(grid), dim3((W) * WAVE), 0, st, \
			(const uint32_t *)parents, (u32x4 *)children, (uint32_t *)solved, stats, n, n_tiles); } while (0)
	switch (variant) {
		case 0: RK_LAUNCH(4, true, 4, false, 1); break;        // 256-parent tiles, non-temporal
		case 1: RK_LAUNCH(4, false, 4, false, 1); break;       // 256-parent tiles, plain stores
		case 3: RK_LAUNCH(1, false, 4, false, 1); break;       // 64-parent tiles, plain stores
		case 17: RK_LAUNCH(1, true, 2, false, 1); break;       // 2 waves per workgroup
		case 18: RK_LAUNCH(1, true, 8, false, 1); break;       // 8 waves per workgroup
		case 19: RK_LAUNCH(1, true, 1, false, 1); break;       // 1 wave per workgroup
		case 24: RK_LAUNCH(1, true, 4, true, 1); break;        // software-pipelined input
		case 26: RK_LAUNCH(4, true, 4, true, 1); break;        // ... with 256-parent tiles
		case 27: RK_LAUNCH(1, false, 4, true, 1); break;       // ... with plain stores
		case 28: RK_LAUNCH(1, true, 4, false, 2); break;       // half-round staging (16 waves/CU)
		case 29: RK_LAUNCH(1, true, 4, true, 2); break;        // ... pipelined
		case 40: case 41: case 42: case 43: {                  // geometry-only diagnostics (outputs are junk; n must be a multiple of 64)
			const size_t n_tiles = n / 64;
			unsigned grid = grid_blocks > 0 ? (unsigned)grid_blocks : grid_for(n_tiles, EXP_WAVES, 1u << 20);
			if (variant == 40) hipLaunchKernelGGL((k_expand12_geometry<0, true>), dim3(grid), dim3(EXP_WAVES * WAVE), 0, st, (const uint32_t *)parents, (u32x4 *)children, (uint32_t *)solved, n_tiles);
			if (variant == 41) hipLaunchKernelGGL((k_expand12_geometry<1, true>), dim3(grid), dim3(EXP_WAVES * WAVE), 0, st, (const uint32_t *)parents, (u32x4 *)children, (uint32_t *)solved, n_tiles);
			if (variant == 42) hipLaunchKernelGGL((k_expand12_geometry<0, false>), dim3(grid), dim3(EXP_WAVES * WAVE), 0, st, (const uint32_t *)parents, (u32x4 *)children, (uint32_t *)solved, n_tiles);
			if (variant == 43) hipLaunchKernelGGL((k_expand12_geometry<1, false>), dim3(grid), dim3(EXP_WAVES * WAVE), 0, st, (const uint32_t *)parents, (u32x4 *)children, (uint32_t *)solved, n_tiles);
			break;
		}
		case 44: case 45: case 46: case 47: {
			const size_t n_tiles = n / 64;
			unsigned grid = grid_blocks > 0 ? (unsigned)grid_blocks : grid_for(n_tiles, EXP_WAVES, 1u << 20);
			if (variant == 44) hipLaunchKernelGGL((k_expand12_geometry<2, true>), dim3(grid), dim3(EXP_WAVES * WAVE), 0, st, (const uint32_t *)parents, (u32x4 *)children, (uint32_t *)solved, n_tiles);
			if (variant == 45) hipLaunchKernelGGL((k_expand12_geometry<3, true>), dim3(grid), dim3(EXP_WAVES * WAVE), 0, st, (const uint32_t *)parents, (u32x4 *)children, (uint32_t *)solved, n_tiles);
			if (variant == 46) hipLaunchKernelGGL((k_expand12_geometry<2, false>), dim3(grid), dim3(EXP_WAVES * WAVE), 0, st, (const uint32_t *)parents, (u32x4 *)children, (uint32_t *)solved, n_tiles);
			if (variant == 47) hipLaunchKernelGGL((k_expand12_geometry<3, false>), dim3(grid), dim3(EXP_WAVES * WAVE), 0, st, (const uint32_t *)parents, (u32x4 *)children, (uint32_t *)solved, n_tiles);
			break;
		}
		case 50: case 51: case 52: case 53: case 54: case 55: case 56: case 57: case 58: case 59: {   // pure store streams over the children buffer
			const size_t kib = n * 240 / 1024;
			#define RK_ST(CH, IL, NTS) do { const size_t nw = (kib + CH - 1) / CH; unsigned grid = grid_blocks > 0 ? (unsigned)grid_blocks : grid_for(nw, 4, 1u << 22); \
				hipLaunchKernelGGL((k_store_geometry<CH, IL, NTS>), dim3(grid), dim3(256), 0, st, (u32x4 *)children, kib); } while (0)
			if (variant == 50) RK_ST(15, false, true);
			if (variant == 51) RK_ST(15, false, false);
			if (variant == 52) RK_ST(4, false, true);
			if (variant == 53) RK_ST(4, false, false);
			if (variant == 54) RK_ST(4, true, false);
			if (variant == 55) RK_ST(15, true, true);
			if (variant == 56) RK_ST(1, false, false);
			if (variant == 57) RK_ST(1, false, true);
			if (variant == 58) RK_ST(60, false, true);
			if (variant == 59) RK_ST(16, true, false);
			#undef RK_ST
			break;
		}
		case 80: case 81: case 84: {                                // run-time shaped pure store stream: grid_blocks = ch | last << 8 | shift16 << 16
			const int ch = grid_blocks & 255, last = ((grid_blocks >> 8) & 255) ? ((grid_blocks >> 8) & 255) : 64, shift16 = (grid_blocks >> 16) & 0x7fff;
			if (ch < 1) break;
			const size_t total16 = n * 15, span16 = (size_t)(ch - 1) * 64 + last;
			const size_t nw = (total16 + span16 - 1) / span16;
			const int wpb = variant == 81 ? 1 : 4;
			hipLaunchKernelGGL(k_store_geometry_rt, dim3((unsigned)((nw + wpb - 1) / wpb)), dim3(64 * wpb), 0, st, (u32x4 *)children, total16, ch, last, shift16, variant == 84 ? 1 : 0);
			break;
		}
		case 82: {                                             // grid_blocks = rot | group << 8
			const size_t n_pages = n * 240 / 4096;
			int group = (grid_blocks >> 8) & 0xffff; if (group < 1) group = 8;
			hipLaunchKernelGGL(k_store_geometry_page, dim3((unsigned)n_pages), dim3(256), 0, st, (u32x4 *)children, n_pages, grid_blocks & 255, group);
			break;
		}
		case 83: {                                             // grid_blocks = sleep | mode << 8 | waves per workgroup << 12
			const size_t n_kib = n * 240 / 1024;
			int wpb = (grid_blocks >> 12) & 31; if (wpb < 1) wpb = 4;
			hipLaunchKernelGGL(k_store_geometry_life, dim3((unsigned)((n_kib + wpb - 1) / wpb)), dim3(64 * wpb), 0, st, (u32x4 *)children, (const uint32_t *)parents, n_kib,
				grid_blocks & 255, (grid_blocks >> 8) & 15);
			break;
		}
		case 86: {                                             // grid_blocks = stores per wave | LDS KiB per workgroup << 8 | waves per workgroup << 16
			const int ch = grid_blocks & 255, lds_kib = (grid_blocks >> 8) & 255; int wpb = (grid_blocks >> 16) & 15; if (wpb < 1) wpb = 1;
			if (ch < 1) break;
			static bool once = false;
			if (!once) { (void)hipFuncSetAttribute((const void *)k_store_geometry_occ, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); (void)hipGetLastError(); once = true; }
			const size_t total16 = n * 15, nw = (total16 + (size_t)ch * 64 - 1) / ((size_t)ch * 64);
			hipLaunchKernelGGL(k_store_geometry_occ, dim3((unsigned)((nw + wpb - 1) / wpb)), dim3(64 * wpb), (size_t)lds_kib * 1024, st, (u32x4 *)children, total16, ch, 0);
			break;
		}
		case 87: {                                             // grid_blocks = hold ticks | tau in 0.1 ns << 12 | LDS KiB << 20 | (waves per workgroup == 4) << 28
			const unsigned hold = grid_blocks & 4095, tau_ps = ((grid_blocks >> 12) & 255) * 100u, lds_kib = (grid_blocks >> 20) & 255;
			const int wpb = ((grid_blocks >> 28) & 1) ? 4 : 1, ch = 15;
			static bool once = false;
			if (!once) { (void)hipFuncSetAttribute((const void *)k_store_geometry_sched, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); (void)hipGetLastError(); once = true; }
			int per_cu = 0, dev = 0, cus = 256;
			(void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_store_geometry_sched, 64 * wpb, (size_t)lds_kib * 1024);
			(void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
			const size_t total16 = n * 15, nw = (total16 + (size_t)ch * 64 - 1) / ((size_t)ch * 64);
			hipLaunchKernelGGL(k_store_geometry_sched, dim3((unsigned)((nw + wpb - 1) / wpb)), dim3(64 * wpb), (size_t)lds_kib * 1024, st, (u32x4 *)children, total16, ch,
				(unsigned)(per_cu * cus * wpb), tau_ps, hold);
			break;
		}
		case 300: {                                            // the paced kernel: grid_blocks = lead in 10 ns | tau in 0.01 ns << 10 | pull workgroups / 64 << 20
			const unsigned lead = grid_blocks & 1023, tau_ps = ((grid_blocks >> 10) & 1023) * 10u, pull = ((grid_blocks >> 20) & 63) * 64u;
			const size_t n_tiles = (n + EXP_ROUND - 1) / EXP_ROUND;
			const char *ph = std::getenv("RK_PACE_PHASE");
			const PaceConfig pc{true, tau_ps, lead, pull, ph ? (unsigned)std::atoi(ph) / EXP_WAVES * EXP_WAVES : (1u << 24), 0, pull};
			(void)n_tiles;
			launch_expand12_paced(parents, children, solved, stats, n, pc, tau_ps, st);
			break;
		}
		case 88: case 89: case 96: case 97: case 98: case 99: {                                    // grid_blocks = lead in 10 ns (10 bits) | tau in 0.01 ns << 10 (10 bits) | LDS KiB << 20 | 4 waves/WG << 28
			const unsigned lead = grid_blocks & 1023, tau_ps = ((grid_blocks >> 10) & 1023) * 10u, lds_kib = (grid_blocks >> 20) & 255;
			const int wpb = ((grid_blocks >> 28) & 1) ? 4 : 1;
			const void *fn = variant == 88 ? (const void *)k_store_geometry_slot<0> : variant == 89 ? (const void *)k_store_geometry_slot<1>
			               : variant == 96 ? (const void *)k_store_geometry_slot<2> : variant == 97 ? (const void *)k_store_geometry_slot<3>
			               : variant == 98 ? (const void *)k_store_geometry_slot<4> : (const void *)k_store_geometry_slot<5>;
			(void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); (void)hipGetLastError();
			int per_cu = 0, dev = 0, cus = 256;
			(void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * wpb, (size_t)lds_kib * 1024);
			(void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
			const size_t nt = n / 64;
			const dim3 grid((unsigned)((nt + wpb - 1) / wpb)), block(64 * wpb);
			const char *e1 = std::getenv("RK_RD_PERIOD"), *e2 = std::getenv("RK_RD_WINDOW");
			const unsigned rd_period = e1 ? (unsigned)std::atoi(e1) : 0u, rd_window = e2 ? (unsigned)std::atoi(e2) : 0u;
			const char *e3 = std::getenv("RK_WR_GUARD");
			const unsigned wr_guard = e3 ? (unsigned)std::atoi(e3) : 0u;
			#define RK_SLOT(M) hipLaunchKernelGGL(k_store_geometry_slot<M>, grid, block, (size_t)lds_kib * 1024, st, (const uint32_t *)parents, (u32x4 *)children, (uint32_t *)solved, nt, \
				(unsigned)(per_cu * cus * wpb), tau_ps, lead, g_tune_cell, rd_period, rd_window, wr_guard)
			if (variant == 88) RK_SLOT(0); else if (variant == 89) RK_SLOT(1); else if (variant == 96) RK_SLOT(2); else if (variant == 97) RK_SLOT(3); else if (variant == 98) RK_SLOT(4); else RK_SLOT(5);
			#undef RK_SLOT
			break;
		}
		case 310: case 311: case 312: {                        // paced pure read of the children buffer: grid_blocks = tau in 0.01 ns | LDS KiB << 12 | 4 waves/WG << 20
			const unsigned tau_ps = (grid_blocks & 4095) * 10u, lds_kib = (grid_blocks >> 12) & 255;
			const int wpb = ((grid_blocks >> 20) & 1) ? 4 : 1, ch = variant == 310 ? 5 : variant == 311 ? 1 : 16;
			const void *fn = variant == 310 ? (const void *)k_read_geometry_slot<5> : variant == 311 ? (const void *)k_read_geometry_slot<1> : (const void *)k_read_geometry_slot<16>;
			(void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); (void)hipGetLastError();
			int per_cu = 0, dev = 0, cus = 256;
			(void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * wpb, (size_t)lds_kib * 1024);
			(void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
			const size_t total16 = n * 15, nw = (total16 + (size_t)ch * 64 - 1) / ((size_t)ch * 64);
			const dim3 grid((unsigned)((nw + wpb - 1) / wpb)), block(64 * wpb);
			const unsigned res = (unsigned)(per_cu * cus * wpb);
			if (variant == 310) hipLaunchKernelGGL(k_read_geometry_slot<5>, grid, block, (size_t)lds_kib * 1024, st, (const u32x4 *)children, total16, res, tau_ps, g_tune_cell);
			else if (variant == 311) hipLaunchKernelGGL(k_read_geometry_slot<1>, grid, block, (size_t)lds_kib * 1024, st, (const u32x4 *)children, total16, res, tau_ps, g_tune_cell);
			else hipLaunchKernelGGL(k_read_geometry_slot<16>, grid, block, (size_t)lds_kib * 1024, st, (const u32x4 *)children, total16, res, tau_ps, g_tune_cell);
			break;
		}
		case 85: {                                             // grid_blocks = log2(block) | stride << 8 (stride odd)
			const size_t n_pages = n * 240 / 4096;
			hipLaunchKernelGGL(k_store_geometry_scatter, dim3((unsigned)n_pages), dim3(256), 0, st, (u32x4 *)children, n_pages, 1u << (grid_blocks & 31), (unsigned)(grid_blocks >> 8) | 1u);
			break;
		}
		case 90: case 91: case 92: case 93: case 94: case 95: {   // a workgroup of 16 / 8 / 4 waves per 64 parents; 93..95 persistent on grid_blocks workgroups
			const size_t nt = n / 64;
			const unsigned g1 = (unsigned)nt, gp = grid_blocks > 0 ? (unsigned)grid_blocks : 2048u;
			if (variant == 90) hipLaunchKernelGGL((k_expand12_geometry_wg<16, false>), dim3(g1), dim3(1024), 0, st, (const uint32_t *)parents, (u32x4 *)children, (uint32_t *)solved, nt);
			if (variant == 91) hipLaunchKernelGGL((k_expand12_geometry_wg<8, false>), dim3(g1), dim3(512), 0, st, (const uint32_t *)parents, (u32x4 *)children, (uint32_t *)solved, nt);
			if (variant == 92) hipLaunchKernelGGL((k_expand12_geometry_wg<4, false>), dim3(g1), dim3(256), 0, st, (const uint32_t *)parents, (u32x4 *)children, (uint32_t *)solved, nt);
			if (variant == 93) hipLaunchKernelGGL((k_expand12_geometry_wg<16, true>), dim3(gp), dim3(1024), 0, st, (const uint32_t *)parents, (u32x4 *)children, (uint32_t *)solved, nt);
			if (variant == 94) hipLaunchKernelGGL((k_expand12_geometry_wg<8, true>), dim3(gp), dim3(512), 0, st, (const uint32_t *)parents, (u32x4 *)children, (uint32_t *)solved, nt);
			if (variant == 95) hipLaunchKernelGGL((k_expand12_geometry_wg<4, true>), dim3(gp), dim3(256), 0, st, (const uint32_t *)parents, (u32x4 *)children, (uint32_t *)solved, nt);
			break;
		}
		case 60: case 61: case 62: case 63: {                  // paced pure store streams, 15 KiB per wave
			const size_t kib = n * 240 / 1024;
			const size_t nw = (kib + 14) / 15;
			const unsigned grid = (unsigned)((nw + 3) / 4);
			if (variant == 60) hipLaunchKernelGGL((k_store_geometry_paced<15, 0>), dim3(grid), dim3(256), 0, st, (u32x4 *)children, kib);
			if (variant == 61) hipLaunchKernelGGL((k_store_geometry_paced<15, 1>), dim3(grid), dim3(256), 0, st, (u32x4 *)children, kib);
			if (variant == 62) hipLaunchKernelGGL((k_store_geometry_paced<15, 2>), dim3(grid), dim3(256), 0, st, (u32x4 *)children, kib);
			if (variant == 63) hipLaunchKernelGGL((k_store_geometry_paced<15, 4>), dim3(grid), dim3(256), 0, st, (u32x4 *)children, kib);
			break;
		}
		case 72: case 73: case 74: case 75: case 76: case 77: {  // P parents per wave (n must be a multiple of 64)
			#define RK_SMALL(P, MODE) do { const size_t nt = n / (P); hipLaunchKernelGGL((k_expand12_geometry_small<P, MODE>), dim3((unsigned)((nt + 3) / 4)), dim3(256), 0, st, \
				(const uint32_t *)parents, (u32x4 *)children, (uint32_t *)solved, nt); } while (0)
			if (variant == 72) RK_SMALL(16, 0);
			if (variant == 73) RK_SMALL(32, 0);
			if (variant == 74) RK_SMALL(8, 0);
			if (variant == 75) RK_SMALL(16, 3);
			if (variant == 76) RK_SMALL(32, 3);
			if (variant == 77) RK_SMALL(8, 3);
			#undef RK_SMALL
			break;
		}
		case 70: case 71: {
			const size_t n_groups = n / 64;
			if (variant == 70) hipLaunchKernelGGL((k_expand12_geometry_chunk<true>), dim3((unsigned)n_groups), dim3(1024), 0, st, (const uint32_t *)parents, (u32x4 *)children, (uint32_t *)solved, n_groups);
			else hipLaunchKernelGGL((k_expand12_geometry_chunk<false>), dim3((unsigned)n_groups), dim3(1024), 0, st, (const uint32_t *)parents, (u32x4 *)children, (uint32_t *)solved, n_groups);
			break;
		}
		case 100: case 101: case 102: case 104: case 108: case 121: case 122: case 124: case 128: case 141: case 142: case 144: {   // ring form
			const size_t n_tiles = (n + EXP_ROUND - 1) / EXP_ROUND;
			const unsigned grid = grid_blocks > 0 ? (unsigned)grid_blocks : grid_for(n_tiles, EXP_WAVES, variant == 100 ? (1u << 20) : (unsigned)EXP_GRID_PERSISTENT);
			#define RK_RING(DP, NTS, NTL) hipLaunchKernelGGL((k_expand12r<true, DP, NTS, NTL>), dim3(grid), dim3(EXP_WAVES * WAVE), 0, st, \
				(const uint32_t *)parents, (u32x4 *)children, (uint32_t *)solved, stats, n)
			if (variant == 100) RK_RING(0, true, false);
			if (variant == 101) RK_RING(1, true, false);
			if (variant == 102) RK_RING(2, true, false);
			if (variant == 104) RK_RING(4, true, false);
			if (variant == 108) RK_RING(8, true, false);
			if (variant == 121) RK_RING(1, true, true);            // non-temporal parent loads
			if (variant == 122) RK_RING(2, true, true);
			if (variant == 124) RK_RING(4, true, true);
			if (variant == 128) RK_RING(8, true, true);
			if (variant == 141) RK_RING(1, false, false);          // plain stores
			if (variant == 142) RK_RING(2, false, false);
			if (variant == 144) RK_RING(4, false, false);
			#undef RK_RING
			break;
		}
		case 152: case 153: case 154: case 162: case 163: case 164: {       // store cache policies: sc1, sc0 sc1, sc1 nt (ring 0 one-shot / ring 2)
			const size_t n_tiles = (n + EXP_ROUND - 1) / EXP_ROUND;
			const bool ring = variant >= 160;
			const unsigned grid = grid_blocks > 0 ? (unsigned)grid_blocks : grid_for(n_tiles, EXP_WAVES, ring ? (unsigned)EXP_GRID_PERSISTENT : (1u << 20));
			#define RK_SP(DP, SP) hipLaunchKernelGGL((k_expand12r<true, DP, SP, false>), dim3(grid), dim3(EXP_WAVES * WAVE), 0, st, \
				(const uint32_t *)parents, (u32x4 *)children, (uint32_t *)solved, stats, n)
			if (variant == 152) RK_SP(0, 2);
			if (variant == 153) RK_SP(0, 3);
			if (variant == 154) RK_SP(0, 4);
			if (variant == 162) RK_SP(2, 2);
			if (variant == 163) RK_SP(2, 3);
			if (variant == 164) RK_SP(2, 4);
			#undef RK_SP
			break;
		}
		case 170: case 172: case 174: case 176: {                         // ring form + PULL workgroups in front of the grid
			const size_t n_tiles = (n + EXP_ROUND - 1) / EXP_ROUND;
			const unsigned base = grid_blocks > 0 ? (unsigned)grid_blocks : grid_for(n_tiles, EXP_WAVES, variant == 170 ? (1u << 20) : (unsigned)EXP_GRID_PERSISTENT);
			#define RK_PULL(DP, PL) hipLaunchKernelGGL((k_expand12r<true, DP, 1, false, PL>), dim3(base + (PL)), dim3(EXP_WAVES * WAVE), 0, st, \
				(const uint32_t *)parents, (u32x4 *)children, (uint32_t *)solved, stats, n)
			if (variant == 170) RK_PULL(0, 64);
			if (variant == 172) RK_PULL(2, 64);
			if (variant == 174) RK_PULL(2, 32);
			if (variant == 176) RK_PULL(2, 128);
			#undef RK_PULL
			break;
		}
		case 200: case 202: {                                  // read phase then write phase: touch <= 8 M parents (160 MB), then expand them
			const size_t chunk = (size_t)8 << 20;
			for (size_t p0 = 0; p0 < n; p0 += chunk) {
				const size_t m = n - p0 < chunk ? n - p0 : chunk;
				const size_t n16 = m * STATE_BYTES / 16;
				hipLaunchKernelGGL(k_touch, dim3(grid_for(n16, 256, 2048)), dim3(256), 0, st, (const u32x4 *)(parents + p0 * STATE_BYTES), n16);
				const size_t n_tiles = (m + EXP_ROUND - 1) / EXP_ROUND;
				if (variant == 200)
					hipLaunchKernelGGL((k_expand12r<true, 0, true, false>), dim3(grid_for(n_tiles, EXP_WAVES, 1u << 20)), dim3(EXP_WAVES * WAVE), 0, st,
					                   (const uint32_t *)(parents + p0 * STATE_BYTES), (u32x4 *)(children + p0 * 240), (uint32_t *)(solved + p0 * 12), stats, m);
				else
					hipLaunchKernelGGL((k_expand12r<true, 2, true, false>), dim3(grid_blocks > 0 ? grid_blocks : EXP_GRID_PERSISTENT), dim3(EXP_WAVES * WAVE), 0, st,
					                   (const uint32_t *)(parents + p0 * STATE_BYTES), (u32x4 *)(children + p0 * 240), (uint32_t *)(solved + p0 * 12), stats, m);
			}
			break;
		}
		default: RK_LAUNCH(1, true, 4, false, 1); break;       // 16: the shipping shape
	}
	#undef RK_LAUNCH
}


// ================================================================================================================
// 3. one-hot kernel shapes
// ================================================================================================================
// tuning aid (benchmarks/tune_oh.py): states per workgroup step and grid cap (0 = one workgroup per tile)
void launch_as_oh_variant(int tile, int grid_cap, const int8_t *states, void *out, int out_dtype, size_t n, hipStream_t st)
{
	const char *e1 = std::getenv("RK_OH_TAU_PS"), *e2 = std::getenv("RK_OH_LEAD"), *e3 = std::getenv("RK_OH_NT"), *e4 = std::getenv("RK_OH_THREADS");
	const unsigned tau = e1 && grid_cap <= 0 ? (unsigned)std::atoi(e1) : 0u, lead = e2 ? (unsigned)std::atoi(e2) : 50u;
	const bool nts = e3 && std::atoi(e3) != 0, one_wave = e4 && std::atoi(e4) == 64;
	const char *e5 = std::getenv("RK_OH_PULL"), *e6 = std::getenv("RK_OH_PHASE_STATES");
	const unsigned pull = tau > 0 && e5 ? (unsigned)std::atoi(e5) : 0u;
	const size_t phase_states = e6 ? (size_t)std::atol(e6) : ((size_t)1 << 20);
	#define RK_OH2(T, EB, TL, NTS, TH) hipLaunchKernelGGL((k_as_oh<T, EB, TL, NTS, TH>), dim3(grid), dim3(TH), 0, st, (const uint32_t *)states, (u32x4 *)out, n, nt, tau, lead, pull, phase_tiles, next_pace_cell())
	#define RK_OH(TL) do { const size_t nt = (n + (TL) - 1) / (TL); const unsigned phase_tiles = (unsigned)(phase_states / (TL)); \
		const unsigned grid = tau > 0 ? oh_paced_grid(nt, pull, phase_tiles) : grid_for(nt, 1, grid_cap > 0 ? (unsigned)grid_cap : (1u << 22)); \
		if (out_dtype == 0) { if (one_wave) { if (nts) RK_OH2(float, 4, TL, true, 64); else RK_OH2(float, 4, TL, false, 64); } \
		                      else          { if (nts) RK_OH2(float, 4, TL, true, 256); else RK_OH2(float, 4, TL, false, 256); } } \
		else                { if (one_wave) { if (nts) RK_OH2(bf16_tag, 2, TL, true, 64); else RK_OH2(bf16_tag, 2, TL, false, 64); } \
		                      else          { if (nts) RK_OH2(bf16_tag, 2, TL, true, 256); else RK_OH2(bf16_tag, 2, TL, false, 256); } } } while (0)
	if (tile == 8) RK_OH(8); else if (tile == 16) RK_OH(16); else if (tile == 32) RK_OH(32); else RK_OH(64);
	#undef RK_OH
	#undef RK_OH2
}

}  // namespace rk
#endif  // RK_TUNING
