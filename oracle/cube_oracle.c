/*
 * CPU oracle for the cube hot path, plain C -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Restates the algorithm of the reference's librubiks/cube/{maps,cube}.py so that parity can be
 * checked at the full 1 M-state size in seconds, and so that bench.py has a multi-core CPU
 * baseline on the GPU box (the Python reference cannot travel).  Only tests/, smoke() and
 * bench.py's cpu_baseline leg load this library; the product never does.
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py checks every entry point against the
 * golden vectors that oracle/gen_golden.py captured from the real reference.
 *
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC oracle/cube_oracle.c -o oracle/_build/liboracle.so
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Face definitions: corner ring, edge ring, fixed corner orientation, edge flip.
 * Facts of the puzzle as the reference states them at maps.py:74-98; order F,B,T,D,L,R. */
static const struct { uint8_t c[4], e[4], fixed, flip; } FACES[6] = {
	{{0, 1, 2, 3}, {0, 1, 2, 3},   0, 0},
	{{4, 7, 6, 5}, {8, 11, 10, 9}, 0, 0},
	{{0, 3, 7, 4}, {0, 7, 8, 4},   1, 1},
	{{1, 5, 6, 2}, {2, 5, 10, 6},  1, 1},
	{{0, 4, 5, 1}, {1, 4, 9, 5},   2, 0},
	{{7, 3, 2, 6}, {3, 6, 11, 7},  2, 0},
};

static uint8_t LUT[12][2][24];
static uint8_t PERM686[12][48];
static int8_t  SOLVED[20];
static int     ready = 0;

/* maps.py:107-145 restated as an absolute table: action a = 2*face + (1-dir). */
static void build_tables(void)
{
	if (ready) return;
	for (int a = 0; a < 12; a++)
		for (int k = 0; k < 2; k++)
			for (int v = 0; v < 24; v++) LUT[a][k][v] = (uint8_t)v;
	for (int f = 0; f < 6; f++) {
		uint8_t (*pos)[24] = LUT[2 * f], (*neg)[24] = LUT[2 * f + 1];
		for (int j = 0; j < 4; j++) {
			int cf = FACES[f].c[j], ct = FACES[f].c[(j + 1) & 3];
			for (int k = 0; k < 3; k++) {
				int kn = (k == FACES[f].fixed) ? k : 3 - FACES[f].fixed - k;
				pos[0][3 * cf + k] = (uint8_t)(3 * ct + kn);
				neg[0][3 * ct + kn] = (uint8_t)(3 * cf + k);
			}
			int ef = FACES[f].e[j], et = FACES[f].e[(j + 1) & 3];
			for (int k = 0; k < 2; k++) {
				int kn = k ^ FACES[f].flip;
				pos[1][2 * ef + k] = (uint8_t)(2 * et + kn);
				neg[1][2 * et + kn] = (uint8_t)(2 * ef + k);
			}
		}
	}
	/* cube.py:58-65 */
	for (int i = 0; i < 8; i++) SOLVED[i] = (int8_t)(3 * i);
	for (int i = 0; i < 12; i++) SOLVED[8 + i] = (int8_t)(2 * i);

	/* 6x8x6 slot permutations, new[slot] = old[perm[slot]] (cube.py:330-347, maps.py:149-156) */
	static const uint8_t NB[6][4] = {{4, 3, 5, 2}, {3, 4, 2, 5}, {0, 5, 1, 4}, {5, 0, 4, 1}, {2, 1, 3, 0}, {1, 2, 0, 3}};
	static const uint8_t STRIP[4][3] = {{6, 7, 0}, {2, 3, 4}, {4, 5, 6}, {0, 1, 2}};
	for (int f = 0; f < 6; f++) {
		uint8_t *p = PERM686[2 * f], *q = PERM686[2 * f + 1];
		for (int s = 0; s < 48; s++) p[s] = (uint8_t)s;
		for (int pos = 0; pos < 8; pos++) p[8 * f + pos] = (uint8_t)(8 * f + ((pos + 6) & 7));
		for (int k = 0; k < 4; k++)
			for (int t = 0; t < 3; t++)
				p[8 * NB[f][k] + STRIP[k][t]] = (uint8_t)(8 * NB[f][(k + 3) & 3] + STRIP[(k + 3) & 3][t]);
		for (int s = 0; s < 48; s++) q[p[s]] = (uint8_t)s;
	}
	ready = 1;
}

void orc_tables(uint8_t *lut_out /* 576 */, uint8_t *perm686_out /* 576 */, int8_t *solved_out /* 20 */)
{
	build_tables();
	if (lut_out) memcpy(lut_out, LUT, sizeof LUT);
	if (perm686_out) memcpy(perm686_out, PERM686, sizeof PERM686);
	if (solved_out) memcpy(solved_out, SOLVED, sizeof SOLVED);
}

static inline void move20(const int8_t *s, int a, int8_t *o)
{
	for (int i = 0; i < 8; i++)  o[i] = (int8_t)LUT[a][0][(uint8_t)s[i]];
	for (int i = 8; i < 20; i++) o[i] = (int8_t)LUT[a][1][(uint8_t)s[i]];
}

/* cube.py:256-263; actions[i] = 2*face + (1-dir) */
void orc_multi_rotate(const int8_t *states, const uint8_t *actions, int8_t *out, size_t n, int threads)
{
	build_tables();
	#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
	for (ptrdiff_t i = 0; i < (ptrdiff_t)n; i++) move20(states + 20 * i, actions[i], out + 20 * i);
}

/* The fan-out idiom of agents.py:277-281 / train.py:285 fused with the goal test of cube.py:88-89:
 * children parent-major / action-minor; solved[12 i + a] = 1 iff that child equals the solved state. */
void orc_expand12(const int8_t *states, int8_t *children, uint8_t *solved, size_t n, int threads)
{
	build_tables();
	#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
	for (ptrdiff_t i = 0; i < (ptrdiff_t)n; i++)
		for (int a = 0; a < 12; a++) {
			int8_t *o = children + (12 * i + a) * 20;
			move20(states + 20 * i, a, o);
			if (solved) solved[12 * i + a] = (uint8_t)(memcmp(o, SOLVED, 20) == 0);
		}
}

/* cube.py:88-89.  Returns the number of solved rows; *first = index of the first one or -1. */
long long orc_multi_is_solved(const int8_t *states, uint8_t *flags, size_t n, long long *first)
{
	build_tables();
	long long cnt = 0, fst = -1;
	for (size_t i = 0; i < n; i++) {
		int s = memcmp(states + 20 * i, SOLVED, 20) == 0;
		if (flags) flags[i] = (uint8_t)s;
		if (s) { if (fst < 0) fst = (long long)i; cnt++; }
	}
	if (first) *first = fst;
	return cnt;
}

/* cube.py:265-277: oh[n][24 i + s[n][i]] = 1 */
void orc_as_oh(const int8_t *states, float *oh, size_t n)
{
	memset(oh, 0, n * 480 * sizeof(float));
	for (size_t r = 0; r < n; r++)
		for (int i = 0; i < 20; i++) oh[r * 480 + 24 * i + (uint8_t)states[r * 20 + i]] = 1.0f;
}

/* cube.py:349-361 on the flattened one-hot (n, 48 slots, 6 colours) int8 layout */
void orc_multi_rotate686(const int8_t *states, const uint8_t *actions, int8_t *out, size_t n)
{
	build_tables();
	for (size_t r = 0; r < n; r++) {
		const uint8_t *p = PERM686[actions[r]];
		for (int s = 0; s < 48; s++) memcpy(out + r * 288 + 6 * s, states + r * 288 + 6 * p[s], 6);
	}
}

/* Order-independent 64-bit digest of a (n, 20) array plus an order-dependent one; lets a test compare
 * a 240 MB children array against the oracle without holding both. */
void orc_digest(const int8_t *rows, size_t n, uint64_t *sum_out, uint64_t *chain_out)
{
	uint64_t sum = 0, chain = 1469598103934665603ull;
	for (size_t r = 0; r < n; r++) {
		uint64_t h = 1469598103934665603ull;
		for (int i = 0; i < 20; i++) { h ^= (uint8_t)rows[20 * r + i]; h *= 1099511628211ull; }
		sum += h;
		chain = (chain ^ h) * 1099511628211ull;
	}
	*sum_out = sum; *chain_out = chain;
}

int orc_max_threads(void)
{
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}
