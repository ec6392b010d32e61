"""
Per-kernel roofline sweep on one MI355X: every cube kernel on 1 M states (device-resident, back-to-back launches,
HIP events on the launch stream), plus the PCIe-inclusive cost of the NumPy drop-in surface.

    python benchmarks/kernels.py [--n 1000000] [--reps 100]

Prints one JSON object per kernel: algorithmic bytes per state, time per launch, GB/s, fraction of the
8 TB/s HBM peak.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librubiks_amd import _ffi, cube  # noqa: E402

PEAK = 8000.0


def timed(fn, reps, warm=10):
	for _ in range(warm):
		fn()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	torch.cuda.synchronize()
	e0.record()
	for _ in range(reps):
		fn()
	e1.record()
	torch.cuda.synchronize()
	return e0.elapsed_time(e1) / reps * 1e-3


def timed_graph(fn, reps, warm=3):
	"""The same launches captured ONCE into a hipGraph (through torch.cuda.CUDAGraph) and replayed: the host issues one
	graph launch for `reps` kernels, so a small kernel is timed at the GPU's pace and not at the pace of Python + ctypes
	(a launch through the Python shim costs the host about 8 us -- the LAUNCH FLOOR row -- which is what round 2's figures
	for the 1 M-row kernels, 8-10 us each, actually measured)."""
	side = torch.cuda.Stream()
	side.wait_stream(torch.cuda.current_stream())
	with torch.cuda.stream(side):
		for _ in range(warm):
			fn()
	torch.cuda.current_stream().wait_stream(side)
	graph = torch.cuda.CUDAGraph()
	with torch.cuda.graph(graph):
		for _ in range(reps):
			fn()
	for _ in range(3):
		graph.replay()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	torch.cuda.synchronize()
	e0.record()
	for _ in range(5):
		graph.replay()
	e1.record()
	torch.cuda.synchronize()
	return e0.elapsed_time(e1) / (5 * reps) * 1e-3


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--n", type=int, default=1_000_000)
	ap.add_argument("--reps", type=int, default=100)
	args = ap.parse_args()
	n = args.n
	_ffi.check(_ffi.lib().rk_init(0))
	g = torch.Generator(device="cuda")
	g.manual_seed(1)
	acts20 = torch.randint(0, 12, (20, n), device="cuda", dtype=torch.uint8, generator=g)
	states = cube.device.apply_sequences(acts20, False, True)
	acts = acts20[0].contiguous()
	children = torch.empty((12 * n, 20), dtype=torch.int8, device="cuda")
	solved = torch.empty(12 * n, dtype=torch.uint8, device="cuda")
	out = torch.empty_like(states)
	flags = torch.empty(n, dtype=torch.uint8, device="cuda")
	n_oh = min(n, 500_000)
	oh = torch.empty((n_oh, 480), dtype=torch.float32, device="cuda")
	oh16 = torch.empty((n_oh, 480), dtype=torch.bfloat16, device="cuda")
	a = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
	b = torch.empty_like(a)

	rows = []

	def report(name, bytes_per_launch, t, units, unit_name):
		gbs = bytes_per_launch / t / 1e9
		rows.append({"kernel": name, "ms": t * 1e3, "GB/s": round(gbs, 1), "frac_of_8TBs": round(gbs / PEAK, 4),
		             "algorithmic_bytes": bytes_per_launch, f"{unit_name}/s": units / t})
		print(json.dumps(rows[-1]), flush=True)

	# what a small launch costs whatever it does: back-to-back launches of the goal test on ONE 256-state tile, and torch's own
	# copy / fill of buffers as small as the 1 M-row kernels' (20 MB): the per-launch floor those kernels sit on
	tiny, tiny_fl = states[:256].contiguous(), torch.empty(256, dtype=torch.uint8, device="cuda")
	t_floor = timed(lambda: cube.device.multi_is_solved(tiny, tiny_fl), 400)
	rows.append({"kernel": "LAUNCH FLOOR: multi_is_solved on 256 states, back-to-back", "ms": t_floor * 1e3})
	print(json.dumps(rows[-1]), flush=True)
	small_a, small_b = torch.empty(20 * n, dtype=torch.uint8, device="cuda"), torch.empty(20 * n, dtype=torch.uint8, device="cuda")
	report(f"device copy {20 * n // 1_000_000} MB (torch; same bytes as multi_rotate at this size)", 40 * n, timed(lambda: small_b.copy_(small_a), args.reps), 20 * n, "bytes")
	report(f"device fill {20 * n // 1_000_000} MB (torch fill_)", 20 * n, timed(lambda: small_b.fill_(1), args.reps), 20 * n, "bytes")
	del small_a, small_b
	report("device copy 256 MiB (torch, reference point)", 2 * a.numel(), timed(lambda: b.copy_(a), 50), a.numel(), "bytes")
	report("device fill 240 MB (torch fill_, pure store stream)", children.numel(), timed(lambda: children.fill_(1), 50), children.numel(), "bytes")
	report("expand12 + goal test", 272 * n, timed(lambda: cube.device.expand12(states, children, solved), args.reps), n, "expansions")
	report("expand12 without flags", 260 * n, timed(lambda: cube.device.expand12(states, children, want_flags=False), args.reps), n, "expansions")
	planes = cube.device.to_soa(states)
	ch_soa = torch.empty((12, 5, n), dtype=torch.int32, device="cuda")
	fl_soa = torch.empty((12, n), dtype=torch.uint8, device="cuda")
	report("expand12 + goal test, structure-of-arrays planes", 272 * n, timed(lambda: cube.device.expand12_soa(planes, ch_soa, fl_soa), args.reps), n, "expansions")
	del ch_soa, fl_soa
	report("multi_rotate (per-state action)", 41 * n, timed(lambda: cube.device.multi_rotate(states, acts, out), args.reps), n, "transitions")
	# the same with the states coming from HBM (32 distinct sets = 640 MB rotating, as bench.py does for the fan-out)
	rot = [states] + [cube.device.apply_sequences(torch.randint(0, 12, (20, n), device="cuda", dtype=torch.uint8, generator=g), False, True) for _ in range(31)]
	k = [0]
	def rot_rotate():
		k[0] += 1
		cube.device.multi_rotate(rot[k[0] % 32], acts, out)
	def rot_solved():
		k[0] += 1
		cube.device.multi_is_solved(rot[k[0] % 32], flags)
	report("multi_rotate (per-state action), states rotating over 640 MB", 41 * n, timed(rot_rotate, 3 * args.reps), n, "transitions")
	report("multi_is_solved, states rotating over 640 MB", 21 * n, timed(rot_solved, 3 * args.reps), n, "states")
	report("multi_rotate on 12 M rows", 41 * 12 * n, timed(lambda: cube.device.multi_rotate(children, acts.repeat(12), children), 20), 12 * n, "transitions")
	report("multi_is_solved", 21 * n, timed(lambda: cube.device.multi_is_solved(states, flags), args.reps), n, "states")
	# the small launches again, replayed from a hipGraph (GPU pace, see timed_graph)
	report("LAUNCH FLOOR replayed from a hipGraph: multi_is_solved on 256 states", 21 * 256, timed_graph(lambda: cube.device.multi_is_solved(tiny, tiny_fl), 200), 256, "states")
	report("multi_is_solved, replayed from a hipGraph", 21 * n, timed_graph(lambda: cube.device.multi_is_solved(states, flags), 100), n, "states")
	report("multi_rotate (per-state action), replayed from a hipGraph", 41 * n, timed_graph(lambda: cube.device.multi_rotate(states, acts, out), 100), n, "transitions")
	n100 = min(n, 100_000)
	report("expand12 + goal test on 100 k parents, replayed from a hipGraph", 272 * n100,
	       timed_graph(lambda: cube.device.expand12(states[:n100], children[:12 * n100], solved[:12 * n100]), 100), n100, "expansions")
	report("expand12 + goal test on 100 k parents (eager launches)", 272 * n100,
	       timed(lambda: cube.device.expand12(states[:n100], children[:12 * n100], solved[:12 * n100]), args.reps), n100, "expansions")
	report("multi_is_solved on 12 M rows", 21 * 12 * n, timed(lambda: cube.device.multi_is_solved(children, solved), 50), 12 * n, "states")
	report("as_oh f32", (20 + 1920) * n_oh, timed(lambda: cube.device.as_oh(states[:n_oh], oh), 50), n_oh, "states")
	report("as_oh bf16", (20 + 960) * n_oh, timed(lambda: cube.device.as_oh(states[:n_oh], oh16, torch.bfloat16), 50), n_oh, "states")
	report("apply_sequences depth 20 (last state only)", (20 + 20) * n, timed(lambda: cube.device.apply_sequences(acts20, False, True), 50), 20 * n, "transitions")
	# the large kernels again CACHE-NEUTRAL (the rows above reuse one 240 MB / 10 MB input every launch, which the 256 MiB Infinity
	# Cache serves at least in part): three 240 MB row buffers in turn, one-hot inputs from the 32 rotating state sets, two outputs each
	ch3 = [children, torch.empty_like(children), torch.empty_like(children)]
	for c in ch3[1:]:
		cube.device.expand12(rot[1], c, solved)
	acts12 = acts.repeat(12)
	def rot12_rotate():
		k[0] += 1
		cube.device.multi_rotate(ch3[k[0] % 3], acts12, ch3[k[0] % 3])
	def rot12_solved():
		k[0] += 1
		cube.device.multi_is_solved(ch3[k[0] % 3], solved)
	report("multi_rotate on 12 M rows, cache-neutral (3 buffers = 720 MB in turn)", 41 * 12 * n, timed(rot12_rotate, 30), 12 * n, "transitions")
	report("multi_is_solved on 12 M rows, cache-neutral (3 buffers in turn)", 21 * 12 * n, timed(rot12_solved, 60), 12 * n, "states")
	del ch3[1:]
	oh_b, oh16_b = torch.empty_like(oh), torch.empty_like(oh16)
	def rot_oh():
		k[0] += 1
		cube.device.as_oh(rot[k[0] % 32][:n_oh], (oh, oh_b)[k[0] % 2])
	def rot_oh16():
		k[0] += 1
		cube.device.as_oh(rot[k[0] % 32][:n_oh], (oh16, oh16_b)[k[0] % 2], torch.bfloat16)
	report("as_oh f32, cache-neutral (inputs over 32 sets, 2 outputs)", (20 + 1920) * n_oh, timed(rot_oh, 64), n_oh, "states")
	report("as_oh bf16, cache-neutral (inputs over 32 sets, 2 outputs)", (20 + 960) * n_oh, timed(rot_oh16, 64), n_oh, "states")
	del rot, oh_b, oh16_b

	# PCIe-inclusive: the NumPy drop-in surface (host array in, host array out).  The first call of a size allocates its
	# page-locked result buffer (torch's caching host allocator); the figure is the median of the calls after it.
	host = states.cpu().numpy()
	f, d = np.random.randint(0, 6, n), np.random.randint(0, 2, n)
	def wall(fn, reps=5):
		ts = []
		for _ in range(reps + 2):
			t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0); del r
		return ts[0], sorted(ts[2:])[reps // 2]
	first_e, t_expand = wall(lambda: cube.expand(host))
	first_r, t_rot = wall(lambda: cube.multi_rotate(host, f, d))
	ch = None
	print(json.dumps({"host_path": "cube.expand(numpy 1M) incl. H2D 20 MB + D2H 240 MB", "s": t_expand, "first_call_s": first_e, "expansions/s": n / t_expand,
	                  "multi_rotate(numpy 1M) s": t_rot, "multi_rotate first_call_s": first_r, "transitions/s": n / t_rot}), flush=True)
	del ch


if __name__ == "__main__":
	main()
