// Move tables of the cube, generated at compile time from the six face definitions and placed in
// both host memory and the device's constant segment (no runtime initialisation, valid on every GPU).
//
// What the tables mean is fixed by the reference: maps.py:74-98 (face rings), maps.py:107-145
// (how a quarter turn re-codes a cubie), cube.py:330-347 + maps.py:149-156 (6x8x6 sticker cycles).
#pragma once
#include <stdint.h>

namespace rk {

constexpr int N_ACTIONS = 12;
constexpr int STATE_BYTES = 20;     // 8 corner codes + 12 edge codes (cube.py:58-65)
constexpr int STATE_DWORDS = 5;
constexpr int S686_BYTES = 288;     // 6 faces x 8 ring positions x 6 colours, int8 one-hot (cube.py:67-71)
constexpr int S686_SLOTS = 48;

struct Face { uint8_t corner[4], edge[4], fixed_ori, flip; };

// Order F, B, T, D, L, R (cube.py:30).  A positive turn moves the cubie in ring[j] to ring[j+1].
constexpr Face FACES[6] = {
	{{0, 1, 2, 3}, {0, 1, 2, 3},   0, 0},
	{{4, 7, 6, 5}, {8, 11, 10, 9}, 0, 0},
	{{0, 3, 7, 4}, {0, 7, 8, 4},   1, 1},
	{{1, 5, 6, 2}, {2, 5, 10, 6},  1, 1},
	{{0, 4, 5, 1}, {1, 4, 9, 5},   2, 0},
	{{7, 3, 2, 6}, {3, 6, 11, 7},  2, 0},
};
constexpr uint8_t NEIGHBOUR[6][4] = {{4, 3, 5, 2}, {3, 4, 2, 5}, {0, 5, 1, 4}, {5, 0, 4, 1}, {2, 1, 3, 0}, {1, 2, 0, 3}};
constexpr uint8_t STRIP[4][3] = {{6, 7, 0}, {2, 3, 4}, {4, 5, 6}, {0, 1, 2}};

struct alignas(16) Tables {
	// T[a][kind][v]: code of a cubie with code v after action a; kind 0 corner, 1 edge.  a = 2*face + (1-dir).
	uint8_t lut[N_ACTIONS][2][24];
	// rows[kind][v][a]: the same numbers regrouped so that ONE 16-byte LDS read returns the code of a cubie in
	// all 12 children (bytes 12..15 are zero).  This is the table the fan-out kernel stages in LDS.
	uint8_t rows[2][24][16];
	// per_action[a][24*kind + v]: 48-byte rows (6 dwords of corner codes, 6 of edge codes); what kernels whose
	// action differs per state stage in LDS.
	uint8_t per_action[N_ACTIONS][48];
	// 6x8x6: new[slot] = old[perm686[a][slot]], slot = 8*face + ring position
	uint8_t perm686[N_ACTIONS][S686_SLOTS];
	int8_t  solved[STATE_BYTES + 12];   // padded to 32 bytes
	int8_t  solved686[S686_BYTES];
	// 6x8x6 gather tables, ready to be copied to LDS with 16-byte loads: a state is 144 ushorts (48 slots x 3);
	// src686[a][u] = source ushort of output ushort u under action a = perm686[a][u/3]*3 + u%3
	alignas(16) uint8_t src686[N_ACTIONS][144];
	// near686[a][slot] = colour shown by `slot` in the state whose child `a` is solved (= move a^1 applied to solved)
	alignas(16) uint8_t near686[N_ACTIONS][S686_SLOTS];
};

constexpr Tables make_tables()
{
	Tables t{};
	for (int a = 0; a < N_ACTIONS; a++)
		for (int k = 0; k < 2; k++)
			for (int v = 0; v < 24; v++) t.lut[a][k][v] = (uint8_t)v;
	for (int f = 0; f < 6; f++) {
		const int p = 2 * f, n = 2 * f + 1;       // positive (dir 1) and negative (dir 0) action of face f
		for (int j = 0; j < 4; j++) {
			const int cf = FACES[f].corner[j], ct = FACES[f].corner[(j + 1) & 3];
			for (int k = 0; k < 3; k++) {
				const int kn = (k == FACES[f].fixed_ori) ? k : 3 - FACES[f].fixed_ori - k;
				t.lut[p][0][3 * cf + k] = (uint8_t)(3 * ct + kn);
				t.lut[n][0][3 * ct + kn] = (uint8_t)(3 * cf + k);
			}
			const int ef = FACES[f].edge[j], et = FACES[f].edge[(j + 1) & 3];
			for (int k = 0; k < 2; k++) {
				const int kn = k ^ FACES[f].flip;
				t.lut[p][1][2 * ef + k] = (uint8_t)(2 * et + kn);
				t.lut[n][1][2 * et + kn] = (uint8_t)(2 * ef + k);
			}
		}
	}
	for (int k = 0; k < 2; k++)
		for (int v = 0; v < 24; v++)
			for (int a = 0; a < N_ACTIONS; a++) {
				t.rows[k][v][a] = t.lut[a][k][v];
				t.per_action[a][24 * k + v] = t.lut[a][k][v];
			}
	for (int i = 0; i < 8; i++) t.solved[i] = (int8_t)(3 * i);
	for (int i = 0; i < 12; i++) t.solved[8 + i] = (int8_t)(2 * i);

	for (int f = 0; f < 6; f++) {
		uint8_t *p = t.perm686[2 * f], *q = t.perm686[2 * f + 1];
		for (int s = 0; s < S686_SLOTS; s++) p[s] = (uint8_t)s;
		for (int pos = 0; pos < 8; pos++) p[8 * f + pos] = (uint8_t)(8 * f + ((pos + 6) & 7));
		for (int k = 0; k < 4; k++)
			for (int s = 0; s < 3; s++)
				p[8 * NEIGHBOUR[f][k] + STRIP[k][s]] = (uint8_t)(8 * NEIGHBOUR[f][(k + 3) & 3] + STRIP[(k + 3) & 3][s]);
		for (int s = 0; s < S686_SLOTS; s++) q[p[s]] = (uint8_t)s;
		for (int pos = 0; pos < 8; pos++) t.solved686[(8 * f + pos) * 6 + f] = 1;
	}
	for (int a = 0; a < N_ACTIONS; a++) {
		for (int u = 0; u < 144; u++) t.src686[a][u] = (uint8_t)(t.perm686[a][u / 3] * 3 + u % 3);
		for (int slot = 0; slot < S686_SLOTS; slot++) t.near686[a][slot] = (uint8_t)(t.perm686[a ^ 1][slot] >> 3);
	}
	return t;
}

}  // namespace rk
