// Device-side building blocks shared by the cube kernels and the search engines (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rk_tables.h"

namespace rk {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int WAVE = 64;

// The solved 20-byte state as five little-endian dwords (cube.py:58-65).
constexpr uint32_t SOLVED_DW[5] = {0x09060300u, 0x15120F0Cu, 0x06040200u, 0x0E0C0A08u, 0x16141210u};

// One copy per translation unit (3.4 KB), constant-initialised: no runtime set-up, no -fgpu-rdc.
static __constant__ Tables D_TAB = make_tables();

__device__ __forceinline__ bool is_solved5(const uint32_t s[5])
{
	return ((s[0] ^ SOLVED_DW[0]) | (s[1] ^ SOLVED_DW[1]) | (s[2] ^ SOLVED_DW[2]) |
	        (s[3] ^ SOLVED_DW[3]) | (s[4] ^ SOLVED_DW[4])) == 0;
}

// v_perm_b32: selector bytes 0..3 pick bytes of `lo`, 4..7 bytes of `hi`.
__device__ __forceinline__ uint32_t bperm(uint32_t hi, uint32_t lo, uint32_t sel)
{
	return __builtin_amdgcn_perm(hi, lo, sel);
}

// 4x4 byte transpose: y[c] byte r = x[r] byte c.  8 v_perm_b32.
__device__ __forceinline__ void transpose4x4(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3,
                                             uint32_t &y0, uint32_t &y1, uint32_t &y2, uint32_t &y3)
{
	const uint32_t t0 = bperm(x1, x0, 0x05010400u), t1 = bperm(x1, x0, 0x07030602u);
	const uint32_t t2 = bperm(x3, x2, 0x05010400u), t3 = bperm(x3, x2, 0x07030602u);
	y0 = bperm(t2, t0, 0x05040100u);
	y1 = bperm(t2, t0, 0x07060302u);
	y2 = bperm(t3, t1, 0x05040100u);
	y3 = bperm(t3, t1, 0x07060302u);
}

// Four table look-ups at once: x holds four codes (0..23), t the 24-byte table of one (action, kind) as six
// dwords.  The low three bits of a code select inside an 8-entry segment (one v_perm per segment), bits 3 and 4
// pick the segment.
__device__ __forceinline__ uint32_t lut4(uint32_t x, const uint32_t t[6])
{
	const uint32_t sel = x & 0x07070707u;
	const uint32_t r0 = bperm(t[1], t[0], sel);
	const uint32_t r1 = bperm(t[3], t[2], sel);
	const uint32_t r2 = bperm(t[5], t[4], sel);
	// per-byte masks 0x00 / 0xFF from bit 3 / bit 4 of every code, by v_perm_b32's constant selectors (a selector byte of 12 yields 0x00,
	// 13 yields 0xFF): one full-rate instruction.  As `b * 0xFF` they were v_mul_lo_u32, a quarter-rate instruction, and the two of them
	// a third of lut4's cycles (seen in the ISA of the round-5 scan kernels, where lut4 is the whole inner loop; the compiler turns
	// `(b << 8) - b` back into the multiply).
	const uint32_t m1 = bperm(0u, 0u, ((x >> 3) & 0x01010101u) | 0x0C0C0C0Cu);
	const uint32_t m2 = bperm(0u, 0u, ((x >> 4) & 0x01010101u) | 0x0C0C0C0Cu);
	const uint32_t r = (r1 & m1) | (r0 & ~m1);
	return (r2 & m2) | (r & ~m2);
}

// One move on a 20-byte state held as five dwords; tab = the 48-byte per-action table (24 corner + 24 edge codes).
__device__ __forceinline__ void move5(uint32_t s[5], const uint32_t tab[12])
{
	s[0] = lut4(s[0], tab);
	s[1] = lut4(s[1], tab);
	s[2] = lut4(s[2], tab + 6);
	s[3] = lut4(s[3], tab + 6);
	s[4] = lut4(s[4], tab + 6);
}

// 48-byte per-action rows: 12 x 48 B = 576 B in LDS.  Rows are 12 dwords apart, so the three 16-byte reads of
// different actions never share a bank (a*12 mod 64 is distinct for a = 0..11).
__device__ __forceinline__ void stage_action_tables(u32x4 *lds /* 36 x u32x4 */, int tid)
{
	if (tid < 36) {
		const int a = tid / 3, part = tid % 3;
		const uint32_t *src = reinterpret_cast<const uint32_t *>(D_TAB.per_action[a]) + 4 * part;
		lds[tid] = u32x4{src[0], src[1], src[2], src[3]};
	}
}

__device__ __forceinline__ void load_action_table(const u32x4 *lds, uint32_t a, uint32_t tab[12])
{
	const u32x4 r0 = lds[a * 3 + 0], r1 = lds[a * 3 + 1], r2 = lds[a * 3 + 2];
	tab[0] = r0.x; tab[1] = r0.y; tab[2] = r0.z;  tab[3] = r0.w;
	tab[4] = r1.x; tab[5] = r1.y; tab[6] = r1.z;  tab[7] = r1.w;
	tab[8] = r2.x; tab[9] = r2.y; tab[10] = r2.z; tab[11] = r2.w;
}

// LDS traffic between lanes of ONE wave needs no s_barrier (a wave's DS operations execute in order); this only
// stops the compiler from moving LDS accesses across the hand-off.
__device__ __forceinline__ void wave_lds_fence()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace rk
