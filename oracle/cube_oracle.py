"""
CPU oracle for the cube hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This is a plain NumPy restatement of the algorithm in the reference's
`librubiks/cube/{maps,cube}.py`.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it; the product (`librubiks_amd`) never does
and fails loudly when its HIP library is missing.

Parity status: PINNED.  `oracle/gen_golden.py` imports the real reference in the build
container and writes `tests/golden/*`; `tests/test_oracle_golden.py` checks every function
below against those vectors and against the data the reference itself ships
(`frontend/src/assets/maps.json`, the literals of `tests/test_cube.py`).

Every function cites the reference lines it restates (paths relative to /root/reference).
Nothing here is copied: the tables are regenerated from the six face definitions.
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------------------
# Face definitions (facts about the puzzle; reference states them at maps.py:74-98).
# Per face: the 4 corner slots and 4 edge slots visited by a positive quarter turn (slot
# x moves to the slot listed after it, cyclically), the corner orientation that is left
# unchanged by the turn (the other two swap), and whether edge orientation flips.
# Face order F, B, T, D, L, R  (cube.py:30).
# --------------------------------------------------------------------------------------
FACE_NAMES = "FBTDLR"
_FACES = (
	# corner ring      edge ring        fixed-ori  edge-flip
	((0, 1, 2, 3),   (0, 1, 2, 3),    0,         False),  # F
	((4, 7, 6, 5),   (8, 11, 10, 9),  0,         False),  # B
	((0, 3, 7, 4),   (0, 7, 8, 4),    1,         True),   # T
	((1, 5, 6, 2),   (2, 5, 10, 6),   1,         True),   # D
	((0, 4, 5, 1),   (1, 4, 9, 5),    2,         False),  # L
	((7, 3, 2, 6),   (3, 6, 11, 7),   2,         False),  # R
)

N_ACTIONS = 12
STATE_BYTES = 20
#: kind[i] = 0 for the 8 corner cubies, 1 for the 12 edge cubies (cube.py:240)
KIND = np.array([0] * 8 + [1] * 12, dtype=np.intp)


def action_face_dir(a):
	"""Action index -> (face, dir).  a = 2*face + (1-dir)  (cube.py:33-34)."""
	a = np.asarray(a)
	return a // 2, 1 - (a % 2)


def face_dir_action(face, direction):
	"""(face, dir) -> action index; inverse of `action_face_dir`."""
	return 2 * np.asarray(face) + (1 - np.asarray(direction))


def build_lut() -> np.ndarray:
	"""
	Absolute move table T[a][kind][v] -> new code, uint8 (12, 2, 24).
	Corner code = slot*3 + orientation, edge code = slot*2 + orientation
	(maps.py:101-105).  Built from the face rings exactly as maps.py:121-140 prescribes:
	a positive turn carries a cubie from ring[j] to ring[j+1]; corner orientation k stays
	if k is the face's fixed orientation, otherwise becomes the third one; edge
	orientation toggles on T and D.  The negative turn is the inverse permutation.
	"""
	lut = np.tile(np.arange(24, dtype=np.uint8), (N_ACTIONS, 2, 1))
	for face, (cring, ering, fixed, flip) in enumerate(_FACES):
		pos = lut[2 * face]       # dir = 1
		neg = lut[2 * face + 1]   # dir = 0
		for j in range(4):
			c_from, c_to = cring[j], cring[(j + 1) % 4]
			for k in range(3):
				k_new = k if k == fixed else 3 - fixed - k
				pos[0, 3 * c_from + k] = 3 * c_to + k_new
				neg[0, 3 * c_to + k_new] = 3 * c_from + k
			e_from, e_to = ering[j], ering[(j + 1) % 4]
			for k in range(2):
				k_new = k ^ int(flip)
				pos[1, 2 * e_from + k] = 2 * e_to + k_new
				neg[1, 2 * e_to + k_new] = 2 * e_from + k
	return lut


def build_delta_maps() -> np.ndarray:
	"""
	The reference's own table layout: maps[dir][face][kind][v] = new_v - v, int8
	(2, 6, 2, 24), dir 0 = negative, 1 = positive (maps.py:107-145, esp. :143).
	"""
	lut = build_lut().astype(np.int16)
	ident = np.arange(24, dtype=np.int16)
	maps = np.zeros((2, 6, 2, 24), dtype=np.int8)
	for a in range(N_ACTIONS):
		face, d = a // 2, 1 - a % 2
		maps[d, face] = (lut[a] - ident).astype(np.int8)
	return maps


LUT = build_lut()
DELTA = build_delta_maps()


def child_rows() -> np.ndarray:
	"""
	Re-indexing of LUT used by the device kernels: ROWS[kind][v][a] (2, 24, 16) uint8,
	the code of a cubie in each of its 12 children, padded to 16 bytes.  Pure transpose of
	`LUT`; lives here so tests can compare the table the library exports.
	"""
	rows = np.zeros((2, 24, 16), dtype=np.uint8)
	rows[:, :, :12] = LUT.transpose(1, 2, 0)
	return rows


# --------------------------------------------------------------------------------------
# 20-byte representation
# --------------------------------------------------------------------------------------
def solved_2024() -> np.ndarray:
	"""Solved state: corner i in slot i, edge j in slot j, all orientation 0 (cube.py:58-65)."""
	return np.concatenate([3 * np.arange(8), 2 * np.arange(12)]).astype(np.int8)


SOLVED = solved_2024()


def rotate(state: np.ndarray, face: int, direction: int) -> np.ndarray:
	"""One move on one state, out of place (cube.py:244-254)."""
	a = int(face) * 2 + (1 - int(direction))
	return LUT[a, KIND, state].astype(np.int8)


def multi_rotate(states: np.ndarray, faces, dirs) -> np.ndarray:
	"""Row i gets move (faces[i], dirs[i]) (cube.py:256-263).  Any integer dtype accepted."""
	states = np.asarray(states)
	a = (np.asarray(faces).astype(np.intp) * 2 + (1 - np.asarray(dirs).astype(np.intp)))
	return LUT[a[:, None], KIND[None, :], states].astype(np.int8)


def expand12(states: np.ndarray) -> np.ndarray:
	"""
	All 12 children of every state, parent-major / action-minor -- the result of the
	idiom `multi_rotate(np.repeat(S, 12, 0), *iter_actions(len(S)))`
	(agents.py:277-281, :513; train.py:285).  Returns (12*N, 20) int8.
	"""
	states = np.asarray(states)
	n = len(states)
	out = LUT[np.arange(N_ACTIONS)[None, :, None], KIND[None, None, :], states[:, None, :]]
	return out.reshape(n * N_ACTIONS, STATE_BYTES).astype(np.int8)


def is_solved(state: np.ndarray) -> bool:
	"""cube.py:85-86."""
	return bool((np.asarray(state) == SOLVED).all())


def multi_is_solved(states: np.ndarray) -> np.ndarray:
	"""cube.py:88-89 -> bool[N]."""
	return (np.asarray(states) == SOLVED).all(axis=1)


def iter_actions(n: int = 1) -> np.ndarray:
	"""uint8 (2, 12 n): faces row 0,0,1,1,..,5,5 and dirs row 1,0,1,0,.. tiled n times (cube.py:179-184)."""
	one = np.array([np.repeat(np.arange(6), 2), np.tile([1, 0], 6)], dtype=np.uint8)
	return np.tile(one, (1, n))


def indices_to_actions(indices: np.ndarray):
	"""cube.py:186-192."""
	indices = np.asarray(indices)
	return indices // 2, 1 - indices % 2


def rev_action(a: int) -> int:
	"""Inverse move: even <-> odd partner (cube.py:194-195)."""
	return a ^ 1


def rev_actions(a: np.ndarray) -> np.ndarray:
	"""cube.py:197-200."""
	return np.asarray(a) ^ 1


def repeat_state(state: np.ndarray, n: int = N_ACTIONS) -> np.ndarray:
	"""cube.py:142-147."""
	state = np.asarray(state)
	return np.broadcast_to(state, (n, *state.shape)).copy()


def scramble(depth: int, force_not_solved: bool = False):
	"""
	cube.py:206-216.  Draw order matters for seed parity: all faces first, then all dirs,
	from the legacy global NumPy generator; re-draw while forced and solved.
	"""
	while True:
		faces = np.random.randint(6, size=(depth,))
		dirs = np.random.randint(2, size=(depth,))
		state = SOLVED.copy()
		for f, d in zip(faces, dirs):
			state = rotate(state, f, d)
		if not (force_not_solved and depth != 0 and is_solved(state)):
			return state, faces, dirs


def sequence_states(faces: np.ndarray, dirs: np.ndarray, with_solved: bool) -> np.ndarray:
	"""
	Deterministic core of `sequence_scrambler` (cube.py:224-232): faces/dirs are
	(depth, games); game g emits `depth` states -- optionally the solved one first, then
	the prefix states of its move sequence.  Output is game-major (games*depth, 20).
	"""
	depth, games = faces.shape
	cur = repeat_state(SOLVED, games)
	seq = [cur] if with_solved else []
	for d in range(depth - int(with_solved)):
		cur = multi_rotate(cur, faces[d], dirs[d])
		seq.append(cur)
	if not seq:
		return np.empty((0, STATE_BYTES), dtype=np.int8)
	return np.stack(seq, axis=1).reshape(games * len(seq), STATE_BYTES)


def sequence_scrambler(games: int, depth: int, with_solved: bool):
	"""cube.py:218-234: RNG draw order faces (depth, games) then dirs (depth, games)."""
	faces = np.random.randint(0, 6, (depth, games))
	dirs = np.random.randint(0, 2, (depth, games))
	states = sequence_states(faces, dirs, with_solved)
	return states, as_oh(states)


def as_oh(states: np.ndarray) -> np.ndarray:
	"""
	One-hot (N, 480) float32 with oh[n, 24*i + states[n, i]] = 1 (cube.py:265-277).
	A single state gives (1, 480).  Returned as NumPy (the reference returns torch).
	"""
	states = np.atleast_2d(np.asarray(states))
	n = len(states)
	oh = np.zeros((n, 480), dtype=np.float32)
	cols = 24 * np.arange(STATE_BYTES)[None, :] + states
	oh[np.arange(n)[:, None], cols] = 1
	return oh


# --------------------------------------------------------------------------------------
# 6x8x6 representation (cube.py:311-388, maps.py:149-156)
# --------------------------------------------------------------------------------------
#: neighbours of each face in the order their sticker strips cycle under a positive turn
_NEIGHBOURS = (
	(4, 3, 5, 2),  # F
	(3, 4, 2, 5),  # B
	(0, 5, 1, 4),  # T
	(5, 0, 4, 1),  # D
	(2, 1, 3, 0),  # L
	(1, 2, 0, 3),  # R
)
#: ring positions (0..7) of the three stickers each neighbour contributes, strip k on neighbour k
_STRIPS = ((6, 7, 0), (2, 3, 4), (4, 5, 6), (0, 1, 2))


def build_perm686() -> np.ndarray:
	"""
	Every move in the 6x8x6 representation is a permutation of the 48 (face, ring-position)
	sticker slots: new[slot] = old[perm[a][slot]], slot = 8*face + pos, uint8 (12, 48).
	Positive turn (cube.py:340-342): the face's own ring advances two places
	(new[pos] = old[pos-2]) and strip k of the neighbour cycle takes strip k-1.
	Negative turn (cube.py:343-345) is the inverse.
	"""
	perm = np.tile(np.arange(48, dtype=np.uint8), (N_ACTIONS, 1))
	for face in range(6):
		p = perm[2 * face]
		for pos in range(8):
			p[8 * face + pos] = 8 * face + (pos - 2) % 8
		nb = _NEIGHBOURS[face]
		for k in range(4):
			for dst, src in zip(_STRIPS[k], _STRIPS[k - 1]):
				p[8 * nb[k] + dst] = 8 * nb[k - 1] + src
		inv = perm[2 * face + 1]
		inv[p] = np.arange(48, dtype=np.uint8)
	return perm


PERM686 = build_perm686()


def solved_686() -> np.ndarray:
	"""cube.py:67-71: sticker (f, p) has colour f, one-hot over the last axis."""
	s = np.zeros((6, 8, 6), dtype=np.int8)
	for f in range(6):
		s[f, :, f] = 1
	return s


SOLVED686 = solved_686()


def rotate686(state: np.ndarray, face: int, direction: int) -> np.ndarray:
	"""cube.py:330-347."""
	a = int(face) * 2 + (1 - int(direction))
	return state.reshape(48, 6)[PERM686[a]].reshape(6, 8, 6)


def multi_rotate686(states: np.ndarray, faces, dirs) -> np.ndarray:
	"""cube.py:349-361 (a Python loop in the reference; a gather here)."""
	n = len(states)
	a = (np.asarray(faces).astype(np.intp) * 2 + (1 - np.asarray(dirs).astype(np.intp)))
	flat = np.asarray(states).reshape(n, 48, 6)
	return flat[np.arange(n)[:, None], PERM686[a]].reshape(n, 6, 8, 6)


def multi_is_solved686(states: np.ndarray) -> np.ndarray:
	"""cube.py:88-89 with the 6x8x6 solved instance."""
	return (np.asarray(states) == SOLVED686).all(axis=(1, 2, 3))


def as_oh686(states: np.ndarray) -> np.ndarray:
	"""cube.py:363-369: flatten to (N, 288) float32."""
	states = np.asarray(states)
	if states.ndim == 3:
		states = states[None]
	return states.reshape(len(states), 288).astype(np.float32)


def as_correct686(oh: np.ndarray) -> np.ndarray:
	"""cube.py:371-380: (N, 6, 8) float32, +1 where the sticker has its face's colour, else -1."""
	oh = np.asarray(oh).reshape(len(oh), 6, 8, 6)
	ok = (oh == SOLVED686).all(axis=3)
	return np.where(ok, 1.0, -1.0).astype(np.float32)


# --------------------------------------------------------------------------------------
# Rendering (cube.py:149-173, 279-307, 382-388; maps.py:26-51)
# --------------------------------------------------------------------------------------
def _sticker_maps():
	"""(face, row, col) of each sticker of each cubie slot in the 6x3x3 picture (maps.py:26-51)."""
	F, B, T, D, L, R = range(6)
	corners = (
		((F, 0, 0), (L, 0, 2), (T, 2, 0)), ((F, 2, 0), (D, 0, 0), (L, 2, 2)),
		((F, 2, 2), (R, 2, 0), (D, 0, 2)), ((F, 0, 2), (T, 2, 2), (R, 0, 0)),
		((B, 0, 2), (T, 0, 0), (L, 0, 0)), ((B, 2, 2), (L, 2, 0), (D, 2, 0)),
		((B, 2, 0), (D, 2, 2), (R, 2, 2)), ((B, 0, 0), (R, 0, 2), (T, 0, 2)),
	)
	edges = (
		((F, 0, 1), (T, 2, 1)), ((F, 1, 0), (L, 1, 2)), ((F, 2, 1), (D, 0, 1)), ((F, 1, 2), (R, 1, 0)),
		((T, 1, 0), (L, 0, 1)), ((D, 1, 0), (L, 2, 1)), ((D, 1, 2), (R, 2, 1)), ((T, 1, 2), (R, 0, 1)),
		((B, 0, 1), (T, 0, 1)), ((B, 1, 2), (L, 1, 0)), ((B, 2, 1), (D, 2, 1)), ((B, 1, 0), (R, 1, 2)),
	)
	return corners, edges


_CORNER_STICKERS, _EDGE_STICKERS = _sticker_maps()


def as633(state: np.ndarray) -> np.ndarray:
	"""20-byte state -> (6, 3, 3) colour picture, faces F,B,T,D,L,R (cube.py:279-307)."""
	pic = np.repeat(np.arange(6), 9).reshape(6, 3, 3)
	for i in range(8):
		slot, ori = divmod(int(state[i]), 3)
		if slot in (0, 2, 5, 7):  # mirrored winding of these slots (cube.py:291-293)
			ori = -ori
		colours = np.roll([s[0] for s in _CORNER_STICKERS[i]], ori)
		for where, col in zip(_CORNER_STICKERS[slot], colours):
			pic[where] = col
	for i in range(12):
		slot, ori = divmod(int(state[8 + i]), 2)
		colours = np.roll([s[0] for s in _EDGE_STICKERS[i]], ori)
		for where, col in zip(_EDGE_STICKERS[slot], colours):
			pic[where] = col
	return pic


#: ring position p of a 686 face sits at this flat index of the 3x3 face after the face's shift
_RING_TO_33 = np.array([0, 3, 6, 7, 8, 5, 2, 1])
_RING_SHIFT = np.array([0, 6, 6, 4, 2, 4])


def as633_686(state: np.ndarray) -> np.ndarray:
	"""6x8x6 state -> (6, 3, 3) (cube.py:382-388)."""
	colours = np.argmax(state, axis=2)  # (6, 8)
	pic = np.repeat(np.arange(6), 9).reshape(6, 9)
	for f in range(6):
		pic[f, _RING_TO_33] = np.roll(colours[f], -_RING_SHIFT[f])
	return pic.reshape(6, 3, 3)


def stringify(pic633: np.ndarray) -> str:
	"""Unfolded-cross text of a (6,3,3) picture (cube.py:160-173)."""
	grid = np.full((9, 12), " ", dtype="<U1")
	where = {2: (0, 1), 4: (1, 0), 0: (1, 1), 5: (1, 2), 1: (1, 3), 3: (2, 1)}
	for face, (r, c) in where.items():
		grid[3 * r:3 * r + 3, 3 * c:3 * c + 3] = pic633[face].astype(str)
	return "\n".join(" ".join(row) for row in grid)
