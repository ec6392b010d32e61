"""
The cube part of one ADI rollout (ref:librubiks/train.py:277-292) at the reference's size, 7 500 games x 30 rows = 225 000 states,
2.7 M children: ONE launch (rk_rollout_fanout, round 5) against the three launches it replaces (rk_apply_sequences,
rk_multi_is_solved, rk_expand12; round 4).  HIP events around back-to-back repetitions with preallocated outputs (the kernels
alone) and through the Python surface (with torch's allocations).  VERDICT r4 #4b.

    python benchmarks/adi_cube.py > profiles/r05_adi_cube.json
    rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 benchmarks/adi_cube.py     (kernel times per launch)
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librubiks_amd import _ffi, cube  # noqa: E402

_ffi.check(_ffi.lib().rk_init(0))
games, depth = 7500, 30
n = games * depth
acts = torch.randint(0, 12, (depth, games), device="cuda", dtype=torch.uint8)
lib, st = _ffi.lib(), _ffi.stream_ptr
states = torch.empty((n, 20), dtype=torch.int8, device="cuda")
sfl = torch.empty(n, dtype=torch.uint8, device="cuda")
children = torch.empty((12 * n, 20), dtype=torch.int8, device="cuda")
cfl = torch.empty(12 * n, dtype=torch.uint8, device="cuda")


def timed(fn, reps=200):
	for _ in range(5):
		fn()
	a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	torch.cuda.synchronize()
	a.record()
	for _ in range(reps):
		fn()
	b.record()
	torch.cuda.synchronize()
	return a.elapsed_time(b) / reps * 1e3


def one_launch():
	_ffi.check(lib.rk_rollout_fanout(0, acts.data_ptr(), depth, games, 1, states.data_ptr(), sfl.data_ptr(), children.data_ptr(), cfl.data_ptr(), None, st()))


def three_launches():
	_ffi.check(lib.rk_apply_sequences(0, acts.data_ptr(), depth, games, 1, 0, states.data_ptr(), st()))
	_ffi.check(lib.rk_multi_is_solved(0, states.data_ptr(), sfl.data_ptr(), None, n, st()))
	_ffi.check(lib.rk_expand12(0, states.data_ptr(), children.data_ptr(), cfl.data_ptr(), None, n, st()))


def surface_three():
	s = cube.device.apply_sequences(acts, True, False)
	return s, cube.device.multi_is_solved(s), cube.device.expand12(s)


bytes_out = n * (20 + 1 + 240 + 12)
row = {"bench": "adi_cube", "games": games, "rows_per_game": depth, "states": n, "children": 12 * n, "bytes_written": bytes_out,
       "one_launch_us": timed(one_launch), "three_launches_us": timed(three_launches),
       "one_launch_us_python_surface": timed(lambda: cube.device.rollout_fanout(acts, True)), "three_launches_us_python_surface": timed(surface_three)}
row["speedup"] = row["three_launches_us"] / row["one_launch_us"]
row["one_launch_frac_of_hbm_peak"] = bytes_out / (row["one_launch_us"] * 1e-6) / 8e12
row["three_launches_frac_of_hbm_peak"] = (bytes_out + 2 * n * 20) / (row["three_launches_us"] * 1e-6) / 8e12          # the states are written once and read twice
print(json.dumps(row), flush=True)
