// Fused one-hot -> first Linear layer of the value/policy net.
//
// The reference one-hot encodes every state into 480 floats (cube.py:265-277: oh[n, 24 i + state[n, i]] = 1) and feeds
// them to nn.Linear(480, H) (model.py:127 / :150, H = 4096 for fc_small): y = oh @ W^T + b.  A one-hot row has exactly 20
// ones, so  y[n, :] = b + sum_i W^T[24 i + state[n, i], :]  -- the 1 920-byte (f32) or 960-byte (bf16) row never has to
// exist in HBM.  Two routes, both reading the 20-byte states directly:
//
//   GATHER (exact f32)  a workgroup keeps a 64-column slice of W^T (480 x 64 f32 = 120 KB) in LDS and sums, for every
//                       row, its 20 weight rows in the fixed order  ((b + w_0) + w_1) + ... + w_19  with plain f32 adds.
//                       16 lanes x 16 B cover a 256-byte LDS row, so one ds_read_b128 wave-instruction serves four
//                       batch rows conflict-free at the full 256 B/clk.  LDS-read bound: 20 x 256 B per (row, 64 columns).
//   MFMA (bf16)         y = A B with v_mfma_f32_32x32x16_bf16 where the A operand (the one-hot tile) is SYNTHESISED IN
//                       REGISTERS: lane (r, h) of a wave needs A[row r][k = 16 s + 8 h .. +7], eight consecutive one-hot
//                       columns, which lie inside one cubie's 24 columns (8 | 24) -- so the fragment is "1.0 at position
//                       state[row][cubie] - offset if that is in 0..7, else zeros": a byte extract, a subtract, a clamp and
//                       one read of a 144-byte table of the nine possible fragments.  No one-hot tile in LDS or HBM.  B = a 64-column tile of W (bf16, 64 x 480, 61 KB) stays in LDS
//                       for all the rows a workgroup handles (row stride 976 B: 244 dwords = 52 mod 64, so the 16 lanes of
//                       a ds_read_b128 group hit 16 different bank quads); accumulators start at the bias; the epilogue
//                       rounds to bf16 (nearest even) and goes through LDS so that a wave stores whole 128-byte rows.
//
//   MFMA, few rows      (round 5) the LDS tile pays when many row tiles reuse it.  A search step's batch can be a few rows (one MCTS
//                       tree: 12; A* at the reference's N = 27: 324; up to 768 take this form): there 64 workgroups each fill
//                       61 KB of LDS for one pass.  k_ohl_mfma_direct gives every WAVE one 32 x 32 output tile
//                       and loads its thirty B fragments straight from a fragment-major copy of W (one contiguous 1 KB per
//                       wave instruction, all thirty in flight at once): no W in LDS, no barrier behind a load loop, 4 x as
//                       many workgroups.  Same MFMA, same k order, same epilogue arithmetic: bit-identical to the tiled form.
//
// Which one is faster where is measured by benchmarks/oh_linear.py (profiles/r02_oh_linear.json, r05_oh_linear_small.json).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/rubiks_hip.h"
#include "rk_device.h"
#include "rk_error.h"

namespace rk {

constexpr int OHL_K = 480;
constexpr int OHL_TN = 64;                   // output columns per workgroup (both routes)
constexpr size_t OHL_DIRECT_MAX_ROWS = 768;  // RK_OHL_MFMA: batches up to here take the direct form (one output tile per wave)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// f32 -> bf16, nearest even, by the hardware convert (v_cvt_pk_bf16_f32): a NaN stays a NaN.  Integer rounding on the f32
// bits -- (u + 0x7FFF + ((u >> 16) & 1)) >> 16, what this file did until round 3 -- turns some NaNs into zeros or
// infinities (MI355X_MICROARCH.md, correctness boundaries), which would hide a diverged net from whoever reads the layer's output.
__device__ __forceinline__ uint32_t f32_pair_to_bf16(float lo, float hi)
{
	const f32x2 pair = {lo, hi};
	return __builtin_bit_cast(uint32_t, __builtin_convertvector(pair, bf16x2));
}
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float x)
{
	return (uint16_t)(f32_pair_to_bf16(x, 0.0f) & 0xFFFFu);
}

// weight preparation: W (H, 480) in f32 or bf16 -> W^T (480, H) f32 and W (H, 480) bf16
// w_frag: the same bf16 values in the order the direct kernel's waves read them -- [column tile of 32][k-step][lane = 32 h + r]
// eight values each: W[32 ct + r][16 ks + 8 h .. + 7], i.e. lane (r, h)'s B operand of v_mfma_f32_32x32x16_bf16 at k-step ks.
__global__ void k_ohl_prepare(const void *w, int w_is_bf16, int H, float *wt_f32, uint16_t *w_bf16, uint16_t *w_frag)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= (size_t)H * OHL_K) return;
	const size_t h = i / OHL_K, k = i - h * OHL_K;
	float v;
	if (w_is_bf16) v = __builtin_bit_cast(float, (uint32_t)reinterpret_cast<const uint16_t *>(w)[i] << 16);
	else v = reinterpret_cast<const float *>(w)[i];
	wt_f32[k * (size_t)H + h] = v;
	const uint16_t b = f32_to_bf16_rne(v);
	w_bf16[i] = b;
	const size_t ct = h >> 5, r = h & 31, ks = k >> 4, hh = (k >> 3) & 1, e = k & 7;
	w_frag[(((ct * (OHL_K / 16) + ks) * 64) + 32 * hh + r) * 8 + e] = b;
}

__global__ void k_ohl_bias(const void *b, int b_is_bf16, int H, float *out)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= H) return;
	out[i] = b == nullptr ? 0.0f
	       : b_is_bf16 ? __builtin_bit_cast(float, (uint32_t)reinterpret_cast<const uint16_t *>(b)[i] << 16)
	                   : reinterpret_cast<const float *>(b)[i];
}

__device__ __forceinline__ uint32_t code_of(const uint32_t s5[5], int cubie)
{
	const uint32_t c = (s5[cubie >> 2] >> (8 * (cubie & 3))) & 0xFFu;
	return c < 24u ? c : 23u;                 // states hold codes 0..23; never index past the weight slice
}

// Optional epilogue  y = scale * act(x W^T + b) + shift  per output column: the activation that follows the layer
// (model.py:157: ELU by default) and the eval-mode BatchNorm1d behind it (model.py:158-159) as its affine map
// scale = gamma / sqrt(var + eps), shift = beta - mean * scale -- two more passes over the (n, H) activations that never
// happen.  ACT: 0 none, 1 ELU(alpha) = x > 0 ? x : alpha (exp(x) - 1), 2 ReLU.  FAST: v_exp_f32 (bf16 outputs keep 8 bits).
template <int ACT, bool FAST>
__device__ __forceinline__ float ohl_act(float v, float alpha)
{
	if (ACT == 1) {
		const float neg = v < 0.0f ? v : 0.0f;
		const float e = (FAST ? __expf(neg) : expf(neg)) - 1.0f;
		return v > 0.0f ? v : alpha * e;
	}
	if (ACT == 2) return v > 0.0f ? v : 0.0f;
	return v;
}

// ---- route GATHER ------------------------------------------------------------------------------------------------
constexpr int OHL_GATHER_THREADS = 512;      // 8 waves (two per SIMD, so that one wave's LDS latency hides behind the other's adds)
constexpr int OHL_GATHER_ROWS = OHL_GATHER_THREADS / 16;

template <bool OUT_BF16, int ACT, bool AFFINE>
__global__ __launch_bounds__(OHL_GATHER_THREADS)
void k_ohl_gather(const uint32_t *__restrict__ states, const float *__restrict__ wt, const float *__restrict__ bias, void *__restrict__ out,
                  size_t n, int H, size_t rows_per_group, float alpha, const float *__restrict__ scale, const float *__restrict__ shift)
{
	__shared__ __attribute__((aligned(16))) float s_w[OHL_K * OHL_TN];       // 122 880 B: this workgroup's 64 columns of W^T
	const int tid = threadIdx.x, c0 = blockIdx.x * OHL_TN;
	{
		// (all fifteen loads of a thread in flight before the first LDS store, as in k_ohl_mfma)
		constexpr int W_LOADS = OHL_K * (OHL_TN / 4) / OHL_GATHER_THREADS;
		static_assert(W_LOADS * OHL_GATHER_THREADS == OHL_K * (OHL_TN / 4), "the W^T slice divides evenly over the workgroup");
		f32x4 w[W_LOADS];
		#pragma unroll
		for (int j = 0; j < W_LOADS; j++) {
			const int i = tid + OHL_GATHER_THREADS * j, k = i >> 4, q = i & 15;
			w[j] = *reinterpret_cast<const f32x4 *>(wt + (size_t)k * H + c0 + 4 * q);
		}
		#pragma unroll
		for (int j = 0; j < W_LOADS; j++) reinterpret_cast<f32x4 *>(s_w)[tid + OHL_GATHER_THREADS * j] = w[j];
	}
	__syncthreads();
	const int lane = tid & 63, wv = tid >> 6, sub = lane >> 4, q = lane & 15;
	const f32x4 b = *reinterpret_cast<const f32x4 *>(bias + c0 + 4 * q);
	f32x4 sc = {1.0f, 1.0f, 1.0f, 1.0f}, sh = {0.0f, 0.0f, 0.0f, 0.0f};
	if (AFFINE) { sc = *reinterpret_cast<const f32x4 *>(scale + c0 + 4 * q); sh = *reinterpret_cast<const f32x4 *>(shift + c0 + 4 * q); }
	const size_t r_begin = (size_t)blockIdx.y * rows_per_group;
	const size_t r_end = r_begin + rows_per_group < n ? r_begin + rows_per_group : n;
	uint32_t nxt[5] = {0u, 0u, 0u, 0u, 0u};
	{
		const size_t r = r_begin + (size_t)(wv * 4 + sub);
		if (r < r_end) {
			#pragma unroll
			for (int j = 0; j < 5; j++) nxt[j] = states[r * 5 + j];
		}
	}
	for (size_t r = r_begin + (size_t)(wv * 4 + sub); r < r_end; r += OHL_GATHER_ROWS) {
		uint32_t s5[5];
		#pragma unroll
		for (int j = 0; j < 5; j++) s5[j] = nxt[j];
		if (r + OHL_GATHER_ROWS < r_end) {                                   // next row's states while this row is summed
			#pragma unroll
			for (int j = 0; j < 5; j++) nxt[j] = states[(r + OHL_GATHER_ROWS) * 5 + j];
		}
		f32x4 acc = b;
		#pragma unroll
		for (int i = 0; i < 20; i++) {                                       // fixed order: ((b + w_0) + w_1) + ...
			const f32x4 w = reinterpret_cast<const f32x4 *>(s_w)[(24 * i + (int)code_of(s5, i)) * 16 + q];
			acc.x += w.x; acc.y += w.y; acc.z += w.z; acc.w += w.w;
		}
		if (ACT != 0) {
			acc.x = ohl_act<ACT, false>(acc.x, alpha); acc.y = ohl_act<ACT, false>(acc.y, alpha);
			acc.z = ohl_act<ACT, false>(acc.z, alpha); acc.w = ohl_act<ACT, false>(acc.w, alpha);
		}
		if (AFFINE) {                                                        // (multiply, then add: no contraction, -ffp-contract=off)
			acc.x = acc.x * sc.x + sh.x; acc.y = acc.y * sc.y + sh.y; acc.z = acc.z * sc.z + sh.z; acc.w = acc.w * sc.w + sh.w;
		}
		if (OUT_BF16) {
			u32x2 v;
			v.x = f32_pair_to_bf16(acc.x, acc.y);
			v.y = f32_pair_to_bf16(acc.z, acc.w);
			*reinterpret_cast<u32x2 *>(reinterpret_cast<uint16_t *>(out) + r * (size_t)H + c0 + 4 * q) = v;
		} else {
			*reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(out) + r * (size_t)H + c0 + 4 * q) = acc;
		}
	}
}

// ---- route MFMA --------------------------------------------------------------------------------------------------
constexpr int OHL_WROW = 976;                // bytes per LDS row of the W tile (480 bf16 + 16 B pad)
constexpr int OHL_DROW = 144;                // bytes per LDS row of a wave's output staging (64 bf16 + 16 B pad)

template <int RT, int ACT, bool AFFINE>
__global__ __launch_bounds__(256, 2)
void k_ohl_mfma(const uint32_t *__restrict__ states, const uint16_t *__restrict__ wb, const float *__restrict__ bias, uint16_t *__restrict__ out,
                size_t n, int H, size_t rows_per_group, float alpha, const float *__restrict__ scale, const float *__restrict__ shift)
{
	__shared__ __attribute__((aligned(16))) uint8_t s_w[OHL_TN * OHL_WROW];      // 62 464 B
	__shared__ __attribute__((aligned(16))) uint8_t s_d[4][32 * OHL_DROW];       // 18 432 B
	__shared__ u32x4 s_frag[9];          // A fragments: entry p < 8 = eight bf16 zeros with 1.0 at position p, entry 8 = all zeros
	const int tid = threadIdx.x, c0 = blockIdx.x * OHL_TN;
	const int lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
	const size_t r_begin = (size_t)blockIdx.y * rows_per_group;
	const size_t r_end = r_begin + rows_per_group < n ? r_begin + rows_per_group : n;
	// A wave multiplies RT 32-row tiles per pass (rows m0 .. m0 + 32 RT - 1): the B fragments it reads from LDS serve all
	// of them (per k-step RT + 2 LDS reads feed 2 RT MFMAs; with one tile per pass the LDS pipe was co-limiting with the
	// matrix pipe) and 2 RT independent accumulator chains keep the MFMAs issuing back to back.  The states of the NEXT pass
	// are requested as soon as the multiplication is done with the current ones, so they land during the epilogue; those of the
	// FIRST pass (and the per-column constants) are requested before the W tile, so that they travel with it.
	uint32_t s5[RT][5];
	auto request = [&](size_t m0) {
		#pragma unroll
		for (int t = 0; t < RT; t++) {
			const size_t row = m0 + 32 * t + r < r_end ? m0 + 32 * t + r : r_end - 1;       // tail rows repeat the last row (never stored)
			#pragma unroll
			for (int j = 0; j < 5; j++) s5[t][j] = states[row * 5 + j];
		}
	};
	constexpr int WROWS = 32 * RT, PASS = 4 * WROWS;                     // rows per wave and per workgroup pass
	if (r_begin + (size_t)wv * WROWS < r_end) request(r_begin + (size_t)wv * WROWS);
	const float bias0 = bias[c0 + r], bias1 = bias[c0 + 32 + r];
	float sc0 = 1.0f, sc1 = 1.0f, sh0 = 0.0f, sh1 = 0.0f;
	if (AFFINE) { sc0 = scale[c0 + r]; sc1 = scale[c0 + 32 + r]; sh0 = shift[c0 + r]; sh1 = shift[c0 + 32 + r]; }
	if (tid < 9) {
		const uint32_t one = 0x3F80u << (16 * (tid & 1));
		const int slot = tid >> 1;
		s_frag[tid] = tid < 8 ? u32x4{slot == 0 ? one : 0u, slot == 1 ? one : 0u, slot == 2 ? one : 0u, slot == 3 ? one : 0u} : u32x4{0u, 0u, 0u, 0u};
	}
	{
		// the W tile: all of a thread's fifteen 16-byte loads in flight before the first LDS store.  As a rolled loop (until round 5) this was
		// load, wait, store fifteen times over: fifteen dependent round trips in front of every workgroup's first MFMA.
		constexpr int W_LOADS = OHL_TN * 60 / 256;
		static_assert(W_LOADS * 256 == OHL_TN * 60, "the W tile divides evenly over the workgroup");
		u32x4 w[W_LOADS];
		#pragma unroll
		for (int j = 0; j < W_LOADS; j++) {
			const int i = tid + 256 * j, row = i / 60, ch = i - row * 60;
			w[j] = *reinterpret_cast<const u32x4 *>(wb + (size_t)(c0 + row) * OHL_K + ch * 8);
		}
		#pragma unroll
		for (int j = 0; j < W_LOADS; j++) {
			const int i = tid + 256 * j, row = i / 60, ch = i - row * 60;
			*reinterpret_cast<u32x4 *>(s_w + row * OHL_WROW + ch * 16) = w[j];
		}
	}
	__syncthreads();
	// offset of this lane's eight columns inside their cubie, by k-step mod 3: (16 ks + 8 h) % 24
	const uint32_t off3[3] = {h ? 8u : 0u, h ? 0u : 16u, h ? 16u : 8u};
	const uint8_t *wrow0 = s_w + r * OHL_WROW + 16 * h, *wrow1 = s_w + (32 + r) * OHL_WROW + 16 * h;
	uint8_t *stage = s_d[wv];
	for (size_t m0 = r_begin + (size_t)wv * WROWS; m0 < r_end; m0 += PASS) {
		f32x16 acc[RT][2];
		#pragma unroll
		for (int t = 0; t < RT; t++)
			#pragma unroll
			for (int v = 0; v < 16; v++) { acc[t][0][v] = bias0; acc[t][1][v] = bias1; }
		#pragma unroll
		for (int ks = 0; ks < OHL_K / 16; ks++) {
			// This lane's eight one-hot columns start at 16 ks + 8 h: inside cubie (16 ks + 8 h) / 24 at offset 0, 8 or 16.
			// The fragment is "1.0 at position code - offset if that is in 0..7, else zeros": a byte extract, a subtract, a
			// clamp and ONE 16-byte LDS read from a nine-entry table (same entry -> broadcast, different entries ->
			// different banks).  Built from compares on the VALU it took ~20 operations per k-step and, with the bf16
			// rounding of the epilogue done by hand, made the kernel VALU-bound (37 % of the MFMA rate).
			const int k_lo = 16 * ks, k_hi = 16 * ks + 8;
			bf16x8 A[RT];
			#pragma unroll
			for (int t = 0; t < RT; t++) {
				const uint32_t c_lo = (s5[t][(k_lo / 24) >> 2] >> (8 * ((k_lo / 24) & 3))) & 0xFFu;
				const uint32_t c_hi = (s5[t][(k_hi / 24) >> 2] >> (8 * ((k_hi / 24) & 3))) & 0xFFu;
				uint32_t rel = (h ? c_hi : c_lo) - off3[ks % 3];         // wraps to a huge value when the code is below the offset
				rel = rel < 8u ? rel : 8u;                               // (codes >= 24 also end up at the zero entry)
				A[t] = __builtin_bit_cast(bf16x8, s_frag[rel]);
			}
			const bf16x8 B0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(wrow0 + 32 * ks));
			const bf16x8 B1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(wrow1 + 32 * ks));
			#pragma unroll
			for (int t = 0; t < RT; t++) acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[t], B0, acc[t][0], 0, 0, 0);
			#pragma unroll
			for (int t = 0; t < RT; t++) acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[t], B1, acc[t][1], 0, 0, 0);
		}
		if (m0 + PASS < r_end) request(m0 + PASS);
		// epilogue, one 32-row tile after the other through the wave's staging area: C/D element v of lane (r, h) is row
		// (v & 3) + 8 (v >> 2) + 4 h, column r; v_cvt_pk_bf16_f32 rounds two values per instruction (nearest even), the
		// halves go to LDS as 16-bit stores and leave as whole 128-byte rows
		#pragma unroll
		for (int t = 0; t < RT; t++) {
			#pragma unroll
			for (int v = 0; v < 16; v++) {
				const int i = (v & 3) + 8 * (v >> 2) + 4 * h;
				f32x2 pair = {acc[t][0][v], acc[t][1][v]};
				if (ACT != 0) { pair.x = ohl_act<ACT, true>(pair.x, alpha); pair.y = ohl_act<ACT, true>(pair.y, alpha); }
				if (AFFINE) { pair.x = pair.x * sc0 + sh0; pair.y = pair.y * sc1 + sh1; }
				const uint32_t packed = __builtin_bit_cast(uint32_t, __builtin_convertvector(pair, bf16x2));
				reinterpret_cast<uint16_t *>(stage + i * OHL_DROW)[r] = (uint16_t)packed;
				reinterpret_cast<uint16_t *>(stage + i * OHL_DROW)[32 + r] = (uint16_t)(packed >> 16);
			}
			wave_lds_fence();
			#pragma unroll
			for (int it = 0; it < 4; it++) {                                     // 32 rows x 128 B, 16 B per lane
				const int idx = it * 64 + lane, i = idx >> 3, ch = idx & 7;
				const u32x4 val = *reinterpret_cast<const u32x4 *>(stage + i * OHL_DROW + ch * 16);
				const size_t row = m0 + 32 * t + i;
				if (row < r_end) *reinterpret_cast<u32x4 *>(out + row * (size_t)H + c0 + ch * 8) = val;
			}
			wave_lds_fence();
		}
	}
}

// ---- route MFMA, few rows: one 32 x 32 output tile per wave, B fragments straight from global memory -----------------
constexpr int OHL_SROW = 80;                 // bytes per LDS row of a wave's output staging (32 bf16 + 16 B pad)

// Workgroup L of the 1-D grid: column group cg = L % col_groups (four 32-column tiles, one per wave), row tile L / col_groups.
// Workgroups go to the XCDs round-robin by L, so with col_groups a multiple of 8 every column group always lands on the same
// XCD: its 120 KB of fragments come from the fabric once and serve the other row tiles from that XCD's L2.
template <int ACT, bool AFFINE>
__global__ __launch_bounds__(256)
void k_ohl_mfma_direct(const uint32_t *__restrict__ states, const u32x4 *__restrict__ wfrag, const float *__restrict__ bias, uint16_t *__restrict__ out,
                       size_t n, int H, unsigned col_groups, float alpha, const float *__restrict__ scale, const float *__restrict__ shift)
{
	__shared__ __attribute__((aligned(16))) uint8_t s_d[4][32 * OHL_SROW];       // 10 240 B
	__shared__ u32x4 s_frag[9];          // A fragments, as in k_ohl_mfma
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
	const unsigned cg = blockIdx.x % col_groups;
	const size_t m0 = (size_t)(blockIdx.x / col_groups) * 32;
	const int ct = (int)cg * 4 + wv, c0 = ct * 32;
	const bool live = c0 < H;                                                     // H is a multiple of 64: the last group may hold two tiles
	// everything this wave will ever read from memory, requested before anything waits: thirty 16-byte fragments and one state
	u32x4 B[OHL_K / 16];
	uint32_t s5[5];
	{
		const u32x4 *wf = wfrag + (size_t)(live ? ct : 0) * (OHL_K / 16) * 64 + lane;
		#pragma unroll
		for (int ks = 0; ks < OHL_K / 16; ks++) B[ks] = wf[ks * 64];
		const size_t row = m0 + r < n ? m0 + r : n - 1;                           // tail rows repeat the last row (never stored)
		#pragma unroll
		for (int j = 0; j < 5; j++) s5[j] = states[row * 5 + j];
	}
	const int col = live ? c0 + r : r;
	const float bias0 = bias[col];
	float sc0 = 1.0f, sh0 = 0.0f;
	if (AFFINE) { sc0 = scale[col]; sh0 = shift[col]; }
	if (tid < 9) {
		const uint32_t one = 0x3F80u << (16 * (tid & 1));
		const int slot = tid >> 1;
		s_frag[tid] = tid < 8 ? u32x4{slot == 0 ? one : 0u, slot == 1 ? one : 0u, slot == 2 ? one : 0u, slot == 3 ? one : 0u} : u32x4{0u, 0u, 0u, 0u};
	}
	__syncthreads();
	if (!live) return;
	const uint32_t off3[3] = {h ? 8u : 0u, h ? 0u : 16u, h ? 16u : 8u};
	f32x16 acc;
	#pragma unroll
	for (int v = 0; v < 16; v++) acc[v] = bias0;
	#pragma unroll
	for (int ks = 0; ks < OHL_K / 16; ks++) {
		const int k_lo = 16 * ks, k_hi = 16 * ks + 8;
		const uint32_t c_lo = (s5[(k_lo / 24) >> 2] >> (8 * ((k_lo / 24) & 3))) & 0xFFu;
		const uint32_t c_hi = (s5[(k_hi / 24) >> 2] >> (8 * ((k_hi / 24) & 3))) & 0xFFu;
		uint32_t rel = (h ? c_hi : c_lo) - off3[ks % 3];
		rel = rel < 8u ? rel : 8u;
		const bf16x8 A = __builtin_bit_cast(bf16x8, s_frag[rel]);
		acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, __builtin_bit_cast(bf16x8, B[ks]), acc, 0, 0, 0);
	}
	// epilogue: C/D element v of lane (r, h) is row (v & 3) + 8 (v >> 2) + 4 h, column r; through the wave's staging area so
	// that the 32 x 64-byte tile leaves as 16-byte stores (two per lane) instead of sixteen 2-byte ones
	uint8_t *stage = s_d[wv];
	#pragma unroll
	for (int v = 0; v < 16; v++) {
		const int i = (v & 3) + 8 * (v >> 2) + 4 * h;
		f32x2 pair = {acc[v], 0.0f};
		if (ACT != 0) pair.x = ohl_act<ACT, true>(pair.x, alpha);
		if (AFFINE) pair.x = pair.x * sc0 + sh0;
		const uint32_t packed = __builtin_bit_cast(uint32_t, __builtin_convertvector(pair, bf16x2));
		reinterpret_cast<uint16_t *>(stage + i * OHL_SROW)[r] = (uint16_t)packed;
	}
	wave_lds_fence();
	#pragma unroll
	for (int it = 0; it < 2; it++) {                                              // 32 rows x 64 B, 16 B per lane
		const int idx = it * 64 + lane, i = idx >> 2, ch = idx & 3;
		const u32x4 val = *reinterpret_cast<const u32x4 *>(stage + i * OHL_SROW + ch * 16);
		const size_t row = m0 + i;
		if (row < n) *reinterpret_cast<u32x4 *>(out + row * (size_t)H + c0 + ch * 8) = val;
	}
}

// ---- the net's other end: the heads' last, narrow Linear with the activation in front of it --------------------------
// model.py:124-125,128-129: policy_net / value_net end in  activation -> [BatchNorm, folded away] -> Linear(K, 12) / Linear(K, 1).  As torch
// runs them that is an elementwise kernel over (n, K) and a GEMM whose output is 12 (or 1, or 13 for both heads side by side) columns
// wide: two launches for a layer that reads n K bf16 values once.  Here: y = act(x) W^T + b in one pass over x, M <= 16 outputs.
// Sixteen lanes share a row (four rows per wave): lane (row, q) owns the 8-element pieces k = (16 c + q) 8 .. + 7 of every 128-element
// chunk c, applies the activation in float32, rounds to bf16 (the value torch's activation kernel would have stored), and feeds
// v_dot2c_f32_bf16 with W from LDS (one ds_read_b128 per output and chunk, the four rows of a wave read the same addresses);
// four DPP row rotations sum the sixteen partial dots.  float32 accumulation, bf16 result.
template <int N> __device__ __forceinline__ float row_ror_f(float v)       // lane i of a 16-lane row reads lane (i + N) % 16 of its row
{
	return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xF, 0xF, false));
}

template <int CHUNKS, int ACT>
__global__ __launch_bounds__(256)
void k_tail_linear(const uint16_t *__restrict__ x, size_t n, size_t ldx, const u32x4 *__restrict__ w, const uint16_t *__restrict__ b, int M, float alpha,
                   uint16_t *__restrict__ out)
{
	extern __shared__ u32x4 s_tw[];                                     // 16 x (16 CHUNKS) pieces of eight bf16 (rows M .. 15 zero)
	constexpr int PER_ROW = 16 * CHUNKS;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, rr = lane >> 4, q = lane & 15;
	const size_t row = (size_t)blockIdx.x * 16 + (size_t)(wv * 4 + rr);
	const size_t ld_row = row < n ? row : n - 1;                        // rows past the end repeat the last row (never stored)
	// everything this thread reads from memory is requested before anything waits: its row's pieces, then the weights for LDS
	u32x4 xv[CHUNKS];
	#pragma unroll
	for (int c = 0; c < CHUNKS; c++) xv[c] = *reinterpret_cast<const u32x4 *>(x + ld_row * ldx + (size_t)(16 * c + q) * 8);
	// the weights into LDS, padded with zero rows to SIXTEEN outputs: the loops below run over all sixteen without a branch (with `if (m < M)`
	// around every output the compiler kept one ds_read_b128 and its four dots per basic block -- every LDS latency exposed, one wave per SIMD)
	const int total = M * PER_ROW;
	for (int base = 0; base < 16 * PER_ROW; base += 4 * 256) {
		u32x4 t[4];
		#pragma unroll
		for (int j = 0; j < 4; j++) { const int i = base + 256 * j + tid; t[j] = w[i < total ? i : total - 1]; }
		#pragma unroll
		for (int j = 0; j < 4; j++) { const int i = base + 256 * j + tid; if (i < 16 * PER_ROW) s_tw[i] = i < total ? t[j] : u32x4{0u, 0u, 0u, 0u}; }
	}
	const float bias = (b != nullptr && q < M) ? __builtin_bit_cast(float, (uint32_t)b[q] << 16) : 0.0f;
	__syncthreads();
	float acc[16];
	#pragma unroll
	for (int m = 0; m < 16; m++) acc[m] = 0.0f;
	#pragma unroll
	for (int c = 0; c < CHUNKS; c++) {
		const uint32_t raw[4] = {xv[c].x, xv[c].y, xv[c].z, xv[c].w};
		bf16x2 a[4];
		#pragma unroll
		for (int j = 0; j < 4; j++) {
			f32x2 v = {__builtin_bit_cast(float, raw[j] << 16), __builtin_bit_cast(float, raw[j] & 0xFFFF0000u)};
			if (ACT != 0) { v.x = ohl_act<ACT, true>(v.x, alpha); v.y = ohl_act<ACT, true>(v.y, alpha); }
			a[j] = __builtin_convertvector(v, bf16x2);
		}
		#pragma unroll
		for (int m = 0; m < 16; m++) {
			const u32x4 wv4 = s_tw[m * PER_ROW + 16 * c + q];
			// (components into scalars first: __builtin_bit_cast applied to `wv4.y` itself reads component x -- seen in the ISA, all
			// four dots had the same B operand and one ds_read_b32 fed them)
			const uint32_t w0 = wv4.x, w1 = wv4.y, w2 = wv4.z, w3 = wv4.w;
			acc[m] = __builtin_amdgcn_fdot2_f32_bf16(a[0], __builtin_bit_cast(bf16x2, w0), acc[m], false);
			acc[m] = __builtin_amdgcn_fdot2_f32_bf16(a[1], __builtin_bit_cast(bf16x2, w1), acc[m], false);
			acc[m] = __builtin_amdgcn_fdot2_f32_bf16(a[2], __builtin_bit_cast(bf16x2, w2), acc[m], false);
			acc[m] = __builtin_amdgcn_fdot2_f32_bf16(a[3], __builtin_bit_cast(bf16x2, w3), acc[m], false);
		}
	}
	float mine = 0.0f;                                                  // output q of this lane's row
	#pragma unroll
	for (int m = 0; m < 16; m++) {
		float v = acc[m];
		v += row_ror_f<8>(v); v += row_ror_f<4>(v); v += row_ror_f<2>(v); v += row_ror_f<1>(v);   // every lane of the row: the row's sum
		if (q == m) mine = v;
	}
	if (row < n && q < M) out[row * (size_t)M + q] = f32_to_bf16_rne(mine + bias);
}

}  // namespace rk

using namespace rk;

// Two 32-row tiles per wave pass ship.  Four (0.75 instead of 1 KB of LDS reads per MFMA, 254 VGPRs, no spills) measure
// the same at every batch size (2.7 M rows: 10.63 vs 10.54 ms): neither LDS nor registers bound the loop.  PMC passes
// (profiles/r02_oh_linear_pmc.json): the chip runs this kernel at an effective 1.82 GHz (DVFS under matrix load) and the
// matrix pipes are busy 66 % of the elapsed cycles -- 1.26 of the 1.9 PFLOP/s available at that clock, where the guide's
// tuned GEMM (63 %) and hipBLASLt's best GEMMs sit too.  The variant stays in the tuning build (RK_OHL_RT=4).
// Requesting the fragments of k-step ks + 1 before the MFMAs of k-step ks (scheduling barriers; the compiler puts every
// ds_read a few instructions in front of its MFMA) does not move it either (8.64 vs 8.5-8.7 ms): with two waves per SIMD
// the LDS latency was already covered.
static bool ohl_rt4(size_t)
{
#ifdef RK_TUNING
	if (const char *e = getenv("RK_OHL_RT")) return atoi(e) == 4;
#endif
	return false;
}

struct rk_ohl {
	int H = 0;
	float *wt_f32 = nullptr, *bias = nullptr;
	uint16_t *w_bf16 = nullptr, *w_frag = nullptr;    // (H, 480) row-major for the LDS-tiled form; fragment-major for the direct form
	int act = RK_OHL_ACT_NONE;                    // epilogue (rk_ohl_set_epilogue)
	float alpha = 1.0f;
	float *affine = nullptr;                      // scale[H] then shift[H], or null
};

extern "C" {

int rk_ohl_create(rk_ohl_t **out, const void *d_weight, int w_dtype, const void *d_bias, int H, void *stream)
{
	if (!out || !d_weight) return fail(RK_EINVAL, "rk_ohl_create: null argument");
	if (w_dtype != RK_OH_F32 && w_dtype != RK_OH_BF16) return fail(RK_EINVAL, "rk_ohl_create: weights must be float32 or bfloat16");
	if (H < OHL_TN || H % OHL_TN != 0 || H > (1 << 20)) return fail(RK_EINVAL, "rk_ohl_create: out_features %d must be a positive multiple of %d", H, OHL_TN);
	rk_ohl *h = new rk_ohl();
	h->H = H;
	hipStream_t st = (hipStream_t)stream;
	hipError_t e = hipMalloc((void **)&h->wt_f32, (size_t)H * OHL_K * sizeof(float));
	if (e == hipSuccess) e = hipMalloc((void **)&h->w_bf16, (size_t)H * OHL_K * sizeof(uint16_t));
	if (e == hipSuccess) e = hipMalloc((void **)&h->w_frag, (size_t)H * OHL_K * sizeof(uint16_t));
	if (e == hipSuccess) e = hipMalloc((void **)&h->bias, (size_t)H * sizeof(float));
	if (e != hipSuccess) { rk_ohl_destroy(h); return fail(RK_EHIP, "rk_ohl_create: hipMalloc failed: %s", hipGetErrorString(e)); }
	const size_t total = (size_t)H * OHL_K;
	hipLaunchKernelGGL(k_ohl_prepare, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_weight, w_dtype == RK_OH_BF16 ? 1 : 0, H, h->wt_f32, h->w_bf16, h->w_frag);
	hipLaunchKernelGGL(k_ohl_bias, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, st, d_bias, w_dtype == RK_OH_BF16 ? 1 : 0, H, h->bias);
	RK_HIP(hipGetLastError());
	RK_HIP(hipStreamSynchronize(st));              // the caller's weight tensor may go away
	*out = h;
	return RK_OK;
}

int rk_ohl_destroy(rk_ohl_t *h)
{
	if (!h) return RK_OK;
	(void)hipFree(h->wt_f32); (void)hipFree(h->w_bf16); (void)hipFree(h->w_frag); (void)hipFree(h->bias); (void)hipFree(h->affine);
	delete h;
	return RK_OK;
}

int rk_ohl_set_epilogue(rk_ohl_t *h, int act, float alpha, const float *d_scale, const float *d_shift, void *stream)
{
	if (!h) return fail(RK_EINVAL, "rk_ohl_set_epilogue: null handle");
	if (act != RK_OHL_ACT_NONE && act != RK_OHL_ACT_ELU && act != RK_OHL_ACT_RELU) return fail(RK_EINVAL, "rk_ohl_set_epilogue: unknown activation %d", act);
	if ((d_scale == nullptr) != (d_shift == nullptr)) return fail(RK_EINVAL, "rk_ohl_set_epilogue: scale and shift come together");
	hipStream_t st = (hipStream_t)stream;
	if (d_scale) {
		if (!h->affine) {
			hipError_t e = hipMalloc((void **)&h->affine, 2 * (size_t)h->H * sizeof(float));
			if (e != hipSuccess) { h->affine = nullptr; return fail(RK_EHIP, "rk_ohl_set_epilogue: hipMalloc failed: %s", hipGetErrorString(e)); }
		}
		RK_HIP(hipMemcpyAsync(h->affine, d_scale, (size_t)h->H * sizeof(float), hipMemcpyDeviceToDevice, st));
		RK_HIP(hipMemcpyAsync(h->affine + h->H, d_shift, (size_t)h->H * sizeof(float), hipMemcpyDeviceToDevice, st));
		RK_HIP(hipStreamSynchronize(st));          // the caller's tensors may go away
	} else if (h->affine) {
		(void)hipFree(h->affine);
		h->affine = nullptr;
	}
	h->act = act;
	h->alpha = alpha;
	return RK_OK;
}

int rk_ohl_forward(rk_ohl_t *h, const int8_t *d_states, void *d_out, int out_dtype, size_t n, int route, void *stream)
{
	if (!h) return fail(RK_EINVAL, "rk_ohl_forward: null handle");
	if (n == 0) return RK_OK;
	if (!d_states || !d_out) return fail(RK_EINVAL, "rk_ohl_forward: null pointer");
	if ((reinterpret_cast<uintptr_t>(d_states) & 3) || (reinterpret_cast<uintptr_t>(d_out) & 15))
		return fail(RK_EINVAL, "rk_ohl_forward: states must be 4-byte and the output 16-byte aligned");
	if (route != RK_OHL_GATHER && route != RK_OHL_MFMA && route != RK_OHL_MFMA_DIRECT && route != RK_OHL_MFMA_TILED)
		return fail(RK_EINVAL, "rk_ohl_forward: unknown route %d", route);
	if (out_dtype != RK_OH_F32 && out_dtype != RK_OH_BF16) return fail(RK_EINVAL, "rk_ohl_forward: output must be float32 or bfloat16");
	if (route != RK_OHL_GATHER && out_dtype != RK_OH_BF16) return fail(RK_EINVAL, "rk_ohl_forward: the MFMA route writes bfloat16");
	hipStream_t st = (hipStream_t)stream;
	// RK_OHL_MFMA picks the form by the batch: one output tile per wave straight from global memory while the waves of a batch
	// do not outnumber what the chip holds at once by much, the LDS-resident W tile (reused by many row tiles) beyond that.
	// Both give the same bits.  OHL_DIRECT_MAX_ROWS: the forms cross between 768 and 1 024 rows at H = 4096 (profiles/r05_oh_linear_small.json).
	const bool direct = route == RK_OHL_MFMA_DIRECT || (route == RK_OHL_MFMA && n <= OHL_DIRECT_MAX_ROWS);
	if (route != RK_OHL_GATHER) route = RK_OHL_MFMA;
	if (direct) {
		const unsigned col_groups = (unsigned)((h->H + 127) / 128);
		const size_t row_tiles = (n + 31) / 32;
		if (row_tiles * col_groups > 0x7FFFFFFFull) return fail(RK_EINVAL, "rk_ohl_forward: %zu rows are too many for the direct form", n);
		const dim3 grid((unsigned)(row_tiles * col_groups));
		const float *scale = h->affine, *shift = h->affine ? h->affine + h->H : nullptr;
		#define RK_OHL_DIRECT_GO(ACT, AFF) hipLaunchKernelGGL((k_ohl_mfma_direct<ACT, AFF>), grid, dim3(256), 0, st, \
			(const uint32_t *)d_states, (const u32x4 *)h->w_frag, h->bias, (uint16_t *)d_out, n, h->H, col_groups, h->alpha, scale, shift)
		if (scale) { if (h->act == RK_OHL_ACT_ELU) RK_OHL_DIRECT_GO(1, true); else if (h->act == RK_OHL_ACT_RELU) RK_OHL_DIRECT_GO(2, true); else RK_OHL_DIRECT_GO(0, true); }
		else       { if (h->act == RK_OHL_ACT_ELU) RK_OHL_DIRECT_GO(1, false); else if (h->act == RK_OHL_ACT_RELU) RK_OHL_DIRECT_GO(2, false); else RK_OHL_DIRECT_GO(0, false); }
		#undef RK_OHL_DIRECT_GO
		RK_HIP(hipGetLastError());
		return RK_OK;
	}
	const unsigned col_tiles = (unsigned)(h->H / OHL_TN);
	// about one workgroup per CU (gather: 120 KB of LDS each) or two (MFMA): the weight slice is loaded once per workgroup
	const bool wide = route == RK_OHL_MFMA && ohl_rt4(n);
	// rows of a workgroup: a multiple of what its four waves multiply per pass (`quantum` = 4 waves x RT tiles x 32 rows), in as many
	// row groups as give the chip's 2 x 256 workgroup slots one workgroup each
	auto plan = [&](size_t quantum, size_t &groups, size_t &rows) {
		groups = (route == RK_OHL_MFMA ? 512u : 256u) / col_tiles;
		if (groups < 1) groups = 1;
	#ifdef RK_TUNING
		if (const char *e = getenv("RK_OHL_GROUPS")) { const long g = atol(e); if (g > 0) groups = (size_t)g; }   // benchmarks/tune_ohl.py
	#endif
		const size_t max_groups = (n + quantum - 1) / quantum;
		if (groups > max_groups) groups = max_groups;
		if (groups > 65535) groups = 65535;
		rows = (n + groups - 1) / groups;
		rows = (rows + quantum - 1) / quantum * quantum;
		groups = (n + rows - 1) / rows;
		// 32-row tiles the busiest SIMD multiplies: workgroups per CU (256 CUs) x passes x tiles per pass
		return (groups * col_tiles + 255) / 256 * (rows / quantum) * (quantum / 128);
	};
	size_t quantum = route == RK_OHL_MFMA ? (wide ? 512 : 256) : OHL_GATHER_ROWS, groups, rows;
	const size_t cost2 = plan(quantum, groups, rows);
	// Three tiles per wave and pass instead of two where that balances the chip: 3 072 rows (an MCTS step of 256 trees) are 6 row groups of
	// 512 rows with two tiles -- 384 workgroups, half of the CUs with two of them and half with one, 18.3 us -- and 8 groups of 384 rows with
	// three: one workgroup per slot, one pass each (profiles/r05_oh_linear_small.json).  Same accumulation order per output: same bits.
	bool three = false;
	if (route == RK_OHL_MFMA && !wide) {
		size_t g3, r3;
		if (plan(384, g3, r3) < cost2) { three = true; quantum = 384; groups = g3; rows = r3; }
	}
	const dim3 grid(col_tiles, (unsigned)groups);
	const float *scale = h->affine, *shift = h->affine ? h->affine + h->H : nullptr;
	const float alpha = h->alpha;
	// the epilogue is a compile-time variant: {none, ELU, ReLU} x {no affine, affine}
	#define RK_OHL_GATHER_GO(BF, ACT, AFF) hipLaunchKernelGGL((k_ohl_gather<BF, ACT, AFF>), grid, dim3(OHL_GATHER_THREADS), 0, st, \
		(const uint32_t *)d_states, h->wt_f32, h->bias, d_out, n, h->H, rows, alpha, scale, shift)
#ifdef RK_TUNING
	#define RK_OHL_MFMA_WIDE(ACT, AFF) if (wide) hipLaunchKernelGGL((k_ohl_mfma<4, ACT, AFF>), grid, dim3(256), 0, st, \
		(const uint32_t *)d_states, h->w_bf16, h->bias, (uint16_t *)d_out, n, h->H, rows, alpha, scale, shift); else
#else
	#define RK_OHL_MFMA_WIDE(ACT, AFF)
#endif
	#define RK_OHL_MFMA_GO(ACT, AFF) do { RK_OHL_MFMA_WIDE(ACT, AFF) if (three) hipLaunchKernelGGL((k_ohl_mfma<3, ACT, AFF>), grid, dim3(256), 0, st, \
		(const uint32_t *)d_states, h->w_bf16, h->bias, (uint16_t *)d_out, n, h->H, rows, alpha, scale, shift); \
		else hipLaunchKernelGGL((k_ohl_mfma<2, ACT, AFF>), grid, dim3(256), 0, st, \
		(const uint32_t *)d_states, h->w_bf16, h->bias, (uint16_t *)d_out, n, h->H, rows, alpha, scale, shift); } while (0)
	#define RK_OHL_BY_EPILOGUE(GO, ...) do { \
		if (scale) { if (h->act == RK_OHL_ACT_ELU) GO(__VA_ARGS__ 1, true); else if (h->act == RK_OHL_ACT_RELU) GO(__VA_ARGS__ 2, true); else GO(__VA_ARGS__ 0, true); } \
		else       { if (h->act == RK_OHL_ACT_ELU) GO(__VA_ARGS__ 1, false); else if (h->act == RK_OHL_ACT_RELU) GO(__VA_ARGS__ 2, false); else GO(__VA_ARGS__ 0, false); } } while (0)
	if (route == RK_OHL_GATHER) {
		if (out_dtype == RK_OH_F32) RK_OHL_BY_EPILOGUE(RK_OHL_GATHER_GO, false,);
		else RK_OHL_BY_EPILOGUE(RK_OHL_GATHER_GO, true,);
	} else {
		RK_OHL_BY_EPILOGUE(RK_OHL_MFMA_GO,);
	}
	#undef RK_OHL_BY_EPILOGUE
	#undef RK_OHL_MFMA_GO
	#undef RK_OHL_MFMA_WIDE
	#undef RK_OHL_GATHER_GO
	RK_HIP(hipGetLastError());
	return RK_OK;
}

int rk_tail_linear(const void *d_x, size_t n, int K, size_t ldx, const void *d_weight, const void *d_bias, int M, int act, float alpha,
                   void *d_out, void *stream)
{
	if (n == 0) return RK_OK;
	if (!d_x || !d_weight || !d_out) return fail(RK_EINVAL, "rk_tail_linear: null pointer");
	if (K != 512 && K != 1024 && K != 2048) return fail(RK_EINVAL, "rk_tail_linear: in_features %d (512, 1024 and 2048 are built)", K);
	if (M < 1 || M > 16) return fail(RK_EINVAL, "rk_tail_linear: out_features %d must be 1 .. 16", M);
	if (ldx < (size_t)K || (ldx & 7) || (reinterpret_cast<uintptr_t>(d_x) & 15) || (reinterpret_cast<uintptr_t>(d_weight) & 15) || (reinterpret_cast<uintptr_t>(d_out) & 1))
		return fail(RK_EINVAL, "rk_tail_linear: rows of x and the weight must be 16-byte aligned (row stride a multiple of 8 elements)");
	if (act != RK_OHL_ACT_NONE && act != RK_OHL_ACT_ELU && act != RK_OHL_ACT_RELU) return fail(RK_EINVAL, "rk_tail_linear: unknown activation %d", act);
	if ((n + 15) / 16 > 0x7FFFFFFFull) return fail(RK_EINVAL, "rk_tail_linear: %zu rows are too many for one launch", n);
	hipStream_t st = (hipStream_t)stream;
	const dim3 grid((unsigned)((n + 15) / 16));
	const size_t lds = (size_t)16 * (size_t)(K / 8) * 16;                // 16 rows (M of them the weights, the rest zero): 16 / 32 / 64 KB
	#define RK_TAIL_GO(CH, ACT) hipLaunchKernelGGL((k_tail_linear<CH, ACT>), grid, dim3(256), lds, st, (const uint16_t *)d_x, n, ldx, \
		(const u32x4 *)d_weight, (const uint16_t *)d_bias, M, alpha, (uint16_t *)d_out)
	#define RK_TAIL_BY_ACT(CH) do { if (act == RK_OHL_ACT_ELU) RK_TAIL_GO(CH, 1); else if (act == RK_OHL_ACT_RELU) RK_TAIL_GO(CH, 2); else RK_TAIL_GO(CH, 0); } while (0)
	if (K == 512) RK_TAIL_BY_ACT(4); else if (K == 1024) RK_TAIL_BY_ACT(8); else RK_TAIL_BY_ACT(16);
	#undef RK_TAIL_BY_ACT
	#undef RK_TAIL_GO
	RK_HIP(hipGetLastError());
	return RK_OK;
}

}  // extern "C"
