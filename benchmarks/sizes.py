"""
Fan-out throughput against batch size with the parents coming from HBM: inputs rotate over >= 640 MB of distinct parent
sets (more than twice the 256 MiB Infinity Cache), outputs over as many sets as fit 2 GB (at least one -- a single output
set is already larger than the cache from 1.1 M parents on).  A plain device fill of the same output is the reference point.
`same_input_frac` repeats the measurement with ONE input set: up to 8 M parents (160 MB) it then stays in the Infinity
Cache across launches, which is what round 2's sweep measured.
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librubiks_amd import cube  # noqa: E402


def timed(fn, reps, warm):
	"""Median of five timed runs of `reps` back-to-back launches, behind at least 20 ms of the same launches: a short burst
	from an idle chip is timed at ramping clocks (the first version of this sweep read 0.69 at 1 M parents where bench.py,
	on the same box, read 0.75)."""
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for i in range(warm):
		fn(i)
	e1.record()
	torch.cuda.synchronize()
	per = max(e0.elapsed_time(e1) / warm, 1e-3)
	for i in range(int(20.0 / per) + 1):
		fn(i)
	out = []
	for _ in range(5):
		torch.cuda.synchronize()
		e0.record()
		for i in range(reps):
			fn(i)
		e1.record()
		torch.cuda.synchronize()
		out.append(e0.elapsed_time(e1) / reps * 1e-3)
	return sorted(out)[2]


def parents(n, seed):
	g = torch.Generator(device="cuda")
	g.manual_seed(seed)
	acts = torch.randint(0, 12, (20, n), device="cuda", dtype=torch.uint8, generator=g)
	return cube.device.apply_sequences(acts, False, True)


SIZES = [int(x) for x in os.environ.get("RK_SIZES", "100000,250000,500000,1000000,2000000,4000000,8000000,16000000,32000000").split(",")]
for n in SIZES:
	n_in = max(2, -(-640_000_000 // (20 * n)))
	n_out = max(1, min(4, 2_000_000_000 // (252 * n)))
	ins = [parents(n, 3 + k) for k in range(n_in)]
	outs = [(torch.empty((12 * n, 20), dtype=torch.int8, device="cuda"), torch.empty(12 * n, dtype=torch.uint8, device="cuda")) for _ in range(n_out)]
	reps = max(12, min(3 * n_in, 300))
	t = timed(lambda i: cube.device.expand12(ins[i % n_in], *outs[i % n_out]), reps, 6)
	t1 = timed(lambda i: cube.device.expand12(ins[0], *outs[i % n_out]), reps, 6)
	tf = timed(lambda i: outs[i % n_out][0].fill_(3), max(12, reps // 2), 3)
	print(json.dumps({"parents": n, "input_sets": n_in, "output_sets": n_out, "ms": t * 1e3, "GB/s": round(272 * n / t / 1e9, 1),
	                  "frac": round(272 * n / t / 8e12, 4), "expansions/s": n / t, "same_input_frac": round(272 * n / t1 / 8e12, 4),
	                  "torch_fill_same_output_GB/s": round(240 * n / tf / 1e9, 1)}), flush=True)
	del ins, outs
	torch.cuda.empty_cache()
