"""
A/B of the fan-out kernel's shapes on one MI355X (interleaved, same process): non-temporal vs plain stores, 256- vs
64-parent wave tiles, static vs dynamically scheduled persistent grid.  Every variant's output is checked against the
shipping kernel's.  Uses the tuning hook rkx_expand12_variant (not part of the public C ABI).
"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librubiks_amd import _ffi, cube  # noqa: E402

N = 1_000_000
lib = _ffi.lib()
lib.rkx_expand12_variant.restype = C.c_int
lib.rkx_expand12_variant.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]


def main():
	_ffi.check(lib.rk_init(0))
	g = torch.Generator(device="cuda")
	g.manual_seed(1)
	acts = torch.randint(0, 12, (20, N), device="cuda", dtype=torch.uint8, generator=g)
	parents = cube.device.apply_sequences(acts, False, True)
	ref_c, ref_f = cube.device.expand12(parents)
	children = torch.empty_like(ref_c)
	solved = torch.empty_like(ref_f)
	counter = torch.zeros(4, dtype=torch.int32, device="cuda")
	names = {0: "nt, tile256, static", 1: "plain, tile256, static", 2: "nt, tile64, static", 3: "plain, tile64, static",
	         4: "nt, tile256, dynamic", 5: "plain, tile256, dynamic", 6: "nt, tile64, dynamic", 7: "plain, tile64, dynamic"}
	configs = [(v, 0) for v in (0, 1, 2, 3)]

	def run(v, gb):
		_ffi.check(lib.rkx_expand12_variant(v, parents.data_ptr(), children.data_ptr(), solved.data_ptr(), None, N, counter.data_ptr(), gb,
		                                    _ffi.stream_ptr()))

	results = {}
	for v, gb in configs:
		children.zero_(); solved.fill_(9)
		run(v, gb)
		ok = torch.equal(children, ref_c) and torch.equal(solved, ref_f)
		results[(v, gb)] = {"variant": names[v], "grid_blocks": gb or "auto", "correct": bool(ok), "ms": []}
	for rep in range(5):                       # interleaved repetitions
		for v, gb in configs:
			for _ in range(5):
				run(v, gb)
			e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
			e0.record()
			for _ in range(40):
				run(v, gb)
			e1.record()
			torch.cuda.synchronize()
			results[(v, gb)]["ms"].append(e0.elapsed_time(e1) / 40)
	# the same with the output rotating over 6 buffer sets (1.5 GB of children: nothing written stays in the 256 MiB
	# Infinity Cache between launches) -- the fair test for cached (plain) vs streaming (non-temporal) stores
	bufs = [(torch.empty_like(ref_c), torch.empty_like(ref_f)) for _ in range(6)]
	for v, gb in [(0, 0), (1, 0), (2, 0), (3, 0)]:
		def run_rot(i):
			c, f = bufs[i % 6]
			_ffi.check(lib.rkx_expand12_variant(v, parents.data_ptr(), c.data_ptr(), f.data_ptr(), None, N, counter.data_ptr(), gb, _ffi.stream_ptr()))
		times = []
		for rep in range(5):
			for i in range(6):
				run_rot(i)
			e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
			e0.record()
			for i in range(42):
				run_rot(i)
			e1.record()
			torch.cuda.synchronize()
			times.append(e0.elapsed_time(e1) / 42)
		ms = sorted(times)[2]
		print(json.dumps({"variant": names[v] + ", ROTATING 6 output buffers", "ms_median": ms, "GB/s": round(272e6 / (ms * 1e-3) / 1e9, 1),
		                  "frac_of_8TBs": round(272e6 / (ms * 1e-3) / 1e9 / 8000, 4)}), flush=True)
	for r in results.values():
		ms = sorted(r["ms"])[len(r["ms"]) // 2]
		r["ms_median"] = ms
		r["GB/s"] = round(272e6 / (ms * 1e-3) / 1e9, 1)
		r["frac_of_8TBs"] = round(r["GB/s"] / 8000, 4)
		r["ms"] = [round(x, 5) for x in r["ms"]]
		print(json.dumps(r), flush=True)


if __name__ == "__main__":
	main()
