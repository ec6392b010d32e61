"""Shared test helpers: seeded inputs built with the oracle (test infrastructure)."""
import hashlib

import numpy as np

from oracle import cube_oracle as orc


def sha(a) -> str:
	return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def random_walk(n: int, depth: int, seed: int) -> np.ndarray:
	"""K3 recipe: n solved states, `depth` rounds of per-row random moves, legacy NumPy RNG."""
	np.random.seed(seed)
	s = orc.repeat_state(orc.SOLVED, n)
	for _ in range(depth):
		f = np.random.randint(0, 6, n)
		d = np.random.randint(0, 2, n)
		s = orc.multi_rotate(s, f, d)
	return s


def random_walk_c(n: int, depth: int, seed: int) -> np.ndarray:
	"""Same draws as random_walk but the moves are applied by the C oracle (fast at n = 1 M)."""
	from oracle import c_oracle
	np.random.seed(seed)
	s = orc.repeat_state(orc.SOLVED, n)
	for _ in range(depth):
		f = np.random.randint(0, 6, n)
		d = np.random.randint(0, 2, n)
		s = c_oracle.multi_rotate(s, (2 * f + (1 - d)).astype(np.uint8), threads=8)
	return s
