"""Per-call latency of the NumPy drop-in surface for tiny inputs (what reference code that loops over `cube.rotate` sees)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librubiks_amd import cube  # noqa: E402


def per_call(fn, reps=2000):
	for _ in range(50):
		fn()
	t0 = time.perf_counter()
	for _ in range(reps):
		fn()
	return (time.perf_counter() - t0) / reps * 1e6


if __name__ == "__main__":
	s = cube.get_solved()
	states = cube.repeat_state(s, 12)
	f, d = cube.iter_actions()
	print(json.dumps({
		"rotate(1 state) us": per_call(lambda: cube.rotate(s, 2, 1)),
		"is_solved(1 state) us": per_call(lambda: cube.is_solved(s)),
		"multi_rotate(12 states) us": per_call(lambda: cube.multi_rotate(states, f, d)),
		"multi_is_solved(12 states) us": per_call(lambda: cube.multi_is_solved(states)),
		"expand(1 state, with flags) us": per_call(lambda: cube.expand(s[None], return_solved=True), 500),
		"scramble(20) us": per_call(lambda: cube.scramble(20), 500),
		"as_oh(12 states) us": per_call(lambda: cube.as_oh(states), 500),
	}))
