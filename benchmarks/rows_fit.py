"""
Where the time of the per-row kernels goes at the headline config's own size (VERDICT r3 #5 / "weak" 3): `multi_rotate`,
`multi_is_solved` and the fused `multi_rotate_solved` on 125 k ... 2 M rows, CACHE-NEUTRAL (every launch reads a slice of a
672 MB pool it has not touched for 32 launches or more), timed per size with HIP events here and -- the same command under
`rocprofv3 --kernel-trace` -- per launch by the profiler (benchmarks/kernel_trace_by_grid.py --segments separates the sizes).

    python benchmarks/rows_fit.py                       one JSON object per (kernel, size) + one fit per kernel
    python benchmarks/rows_fit.py --fit CSV             the same fit from the profiler's per-run averages

The fit: t(n) = t0 + bytes(n) / B.  t0 is what a launch costs whatever its size -- dispatch ramp of the grid, the first HBM round
trip, the tail of the last workgroups --, B the rate at which the kernel streams once it runs.  The fraction of the 8 TB/s peak a
launch of n rows can reach is bytes / (t0 + bytes / B) / 8e12: with t0 of a few microseconds a 21 MB launch cannot reach what a
250 MB launch does, whatever the kernel does per byte.
"""
import argparse
import csv
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SIZES = [125_000, 250_000, 500_000, 1_000_000, 1_500_000, 2_000_000]
BYTES = {"multi_rotate": 41, "multi_is_solved": 21, "multi_rotate_solved": 42}      # algorithmic bytes per row
KERNEL_OF = {"rk::k_multi_rotate<false, false>": "multi_rotate", "rk::k_multi_is_solved": "multi_is_solved",
             "rk::k_multi_rotate<false, true>": "multi_rotate_solved"}


def fit(points):
	"""least squares t = t0 + bytes / B over (bytes, seconds) points"""
	x = np.array([p[0] for p in points], dtype=np.float64)
	y = np.array([p[1] for p in points], dtype=np.float64)
	A = np.stack([np.ones_like(x), x], axis=1)
	(t0, inv_b), *_ = np.linalg.lstsq(A, y, rcond=None)
	return float(t0), float(1.0 / inv_b)


def report(kind, points, source):
	t0, B = fit(points)
	row = {"fit": kind, "source": source, "t0_us": t0 * 1e6, "stream_TBps": B / 1e12, "stream_frac_of_8TBps": B / 8e12,
	       "points_us": {str(int(b // BYTES[kind])): t * 1e6 for b, t in points},
	       "frac_of_8TBps_measured": {str(int(b // BYTES[kind])): b / t / 8e12 for b, t in points},
	       "frac_ceiling_at_1M_rows_with_this_t0": (BYTES[kind] * 1e6) / (t0 + BYTES[kind] * 1e6 / B) / 8e12,
	       "frac_at_1M_rows_if_t0_were_zero": B / 8e12}
	print(json.dumps(row), flush=True)


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--fit", help="CSV of benchmarks/kernel_trace_by_grid.py --segments for this script run under rocprofv3")
	a = ap.parse_args()
	if a.fit:
		per = {}
		with open(a.fit, newline="") as f:
			for r in csv.DictReader(f):
				kind = KERNEL_OF.get(r["Kernel"])
				if kind is None:
					continue
				n_tiles = int(r["Grid_Size"]) // 256 * 4                       # 4 waves per workgroup, one 256-row tile per wave
				n = min(SIZES, key=lambda s: abs((s + 255) // 256 - n_tiles) if abs((s + 255) // 256 - n_tiles) < 8 else 1 << 30)
				per.setdefault(kind, {})[n] = float(r["AverageNs"]) * 1e-9   # the last run of a size wins (the timed loop, not the warm-up)
		for kind, d in per.items():
			report(kind, [(BYTES[kind] * n, t) for n, t in sorted(d.items())], "rocprofv3 kernel trace, average per launch")
		return
	import torch
	from librubiks_amd import _ffi, cube
	_ffi.check(_ffi.lib().rk_init(0))
	g = torch.Generator(device="cuda")
	g.manual_seed(8)
	POOL = 33_600_000                                                     # 672 MB of states: a slice comes round again after >= 16 launches of 2 M rows
	pool = torch.empty((POOL, 20), dtype=torch.int8, device="cuda")
	for lo in range(0, POOL, 4_200_000):
		pool[lo:lo + 4_200_000] = cube.device.apply_sequences(torch.randint(0, 12, (10, 4_200_000), device="cuda", dtype=torch.uint8, generator=g), False, True)
	acts = torch.randint(0, 12, (POOL,), device="cuda", dtype=torch.uint8, generator=g)
	outs = [torch.empty((2_000_000, 20), dtype=torch.int8, device="cuda") for _ in range(2)]
	flags = [torch.empty(2_000_000, dtype=torch.uint8, device="cuda") for _ in range(2)]
	stats = torch.tensor([0, _ffi.INT64_MAX], dtype=torch.int64, device="cuda")
	at = [0]

	def slices(n):
		lo = at[0]
		if lo + n > POOL:
			lo = 0
		at[0] = lo + n
		return pool[lo:lo + n], acts[lo:lo + n]

	kinds = {
		"multi_rotate": lambda n, i: cube.device.multi_rotate(*slices(n), out=outs[i % 2][:n]),
		"multi_is_solved": lambda n, i: cube.device.multi_is_solved(slices(n)[0], flags[i % 2][:n]),
		"multi_rotate_solved": lambda n, i: cube.device.multi_rotate_solved(*slices(n), out=outs[i % 2][:n], flags=flags[i % 2][:n], stats=stats),
	}
	for kind, fn in kinds.items():
		pts = []
		for n in SIZES:
			reps = 300 if n <= 500_000 else 150
			for i in range(10):
				fn(n, i)
			e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
			torch.cuda.synchronize()
			e0.record()
			for i in range(reps):
				fn(n, i)
			e1.record()
			torch.cuda.synchronize()
			t = e0.elapsed_time(e1) / reps * 1e-3
			pts.append((BYTES[kind] * n, t))
			print(json.dumps({"kernel": kind, "rows": n, "us_per_launch_events": t * 1e6, "frac_of_8TBps": BYTES[kind] * n / t / 8e12,
			                  "note": "eager launches through the Python shim: below ~500 k rows the host's launch rate is part of the figure"}), flush=True)
		report(kind, pts, "HIP events around back-to-back eager launches")


if __name__ == "__main__":
	main()
