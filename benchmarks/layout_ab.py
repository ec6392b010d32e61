"""
Structure-of-arrays vs array-of-structures fan-out, settled the way the round-2 verdict asks: both kernels INTERLEAVED in
one process on one box, under bench.py's cache-neutral rotation (parents over 640 MB of distinct input, outputs over 4
sets), >= 5 alternations, median and spread.

    python benchmarks/layout_ab.py [--alternations 7]

Three batch sizes probe the suspected cause of the SoA kernel's box-to-box swing (0.70-0.85 in round 2): its 60 + 12
output planes lie n*4 (n) bytes apart, so all of a wavefront's 72 concurrent store streams hit addresses that differ by a
multiple of the plane stride -- if that stride is a multiple of the memory system's channel-interleave period the streams
alias onto the same channels.  n = 1 000 000 (stride 4 000 000 B), n = 1 048 576 (stride exactly 4 MiB: worst case) and
n = 1 000 448 (stride 4 001 792 B = 977 x 4 KiB + 0: a multiple of 4 KiB but odd in units of 4 KiB).
"""
import argparse
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librubiks_amd import _ffi, cube  # noqa: E402


def parents(n, seed):
	g = torch.Generator(device="cuda")
	g.manual_seed(seed)
	acts = torch.randint(0, 12, (20, n), device="cuda", dtype=torch.uint8, generator=g)
	return cube.device.apply_sequences(acts, False, True)


def timed(fn, launches):
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	for i in range(8):
		fn(i)
	torch.cuda.synchronize()
	e0.record()
	for i in range(launches):
		fn(i)
	e1.record()
	torch.cuda.synchronize()
	return e0.elapsed_time(e1) / launches


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--alternations", type=int, default=7)
	ap.add_argument("--sizes", default="1000000,1048576,1000448")
	args = ap.parse_args()
	_ffi.check(_ffi.lib().rk_init(0))
	for n in (int(x) for x in args.sizes.split(",")):
		n_in = -(-640_000_000 // (20 * n))
		aos_in = [parents(n, 5 + k) for k in range(n_in)]
		soa_in = [cube.device.to_soa(p) for p in aos_in]
		aos_out = [(torch.empty((12 * n, 20), dtype=torch.int8, device="cuda"), torch.empty(12 * n, dtype=torch.uint8, device="cuda")) for _ in range(4)]
		soa_out = [(torch.empty((12, 5, n), dtype=torch.int32, device="cuda"), torch.empty((12, n), dtype=torch.uint8, device="cuda")) for _ in range(4)]
		# same children either way (checked once on set 0)
		c, f = cube.device.expand12(aos_in[0], *aos_out[0])
		cs, fs = cube.device.expand12_soa(soa_in[0], *soa_out[0])
		back = cube.device.from_soa(cs)                                           # (12, n, 20)
		same = bool(torch.equal(back.permute(1, 0, 2).reshape(12 * n, 20), c) and torch.equal(fs.t().reshape(-1), f))
		del back
		aos = lambda i: cube.device.expand12(aos_in[i % n_in], *aos_out[i % 4])
		soa = lambda i: cube.device.expand12_soa(soa_in[i % n_in], *soa_out[i % 4])
		t_aos, t_soa = [], []
		for _ in range(args.alternations):
			t_aos.append(timed(aos, 3 * n_in))
			t_soa.append(timed(soa, 3 * n_in))
		rec = {"parents": n, "plane_stride_bytes": 4 * n, "input_sets": n_in, "output_sets": 4, "alternations": args.alternations, "same_children": same}
		for name, t in (("aos", t_aos), ("soa", t_soa)):
			med = statistics.median(t)
			rec[name] = {"ms_median": med, "ms_min": min(t), "ms_max": max(t), "frac_of_8TBs_median": 272.0 * n / (med * 1e-3) / 8e12,
			             "frac_range": [272.0 * n / (max(t) * 1e-3) / 8e12, 272.0 * n / (min(t) * 1e-3) / 8e12]}
		print(json.dumps(rec), flush=True)
		del aos_in, soa_in, aos_out, soa_out, c, f, cs, fs
		torch.cuda.empty_cache()


if __name__ == "__main__":
	main()
