"""
Roofline sweep of the 6x8x6-representation kernels (288-byte states) on one MI355X.  Every kernel is timed twice: on the SAME
buffers every launch (round 1-2's protocol: the 57.6 MB input, and for the move kernel also its output, then sit in the 256 MiB
Infinity Cache) and CACHE-NEUTRAL, with inputs rotating over 12 sets (691 MB of distinct states) and outputs over several
sets, as bench.py does for the headline.
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librubiks_amd import _ffi, cube  # noqa: E402
from benchmarks.kernels import timed  # noqa: E402

N_IN = 12


def main(n=200_000):
	_ffi.check(_ffi.lib().rk_init(0))
	cube.set_is2024(False)
	solved = torch.from_numpy(cube.get_solved()).cuda()
	g = torch.Generator(device="cuda"); g.manual_seed(0)
	ins = []
	for k in range(N_IN):
		states = solved.unsqueeze(0).repeat(n, 1, 1, 1).contiguous()
		for _ in range(12):
			acts = torch.randint(0, 12, (n,), device="cuda", dtype=torch.uint8, generator=g)
			states = cube.device.multi_rotate(states, acts)
		ins.append(states)
	states = ins[0]
	outs = [torch.empty_like(states) for _ in range(4)]
	children = [torch.empty((12 * n, 6, 8, 6), dtype=torch.int8, device="cuda") for _ in range(2)]
	for c in children:                                                   # real children in both sets (the goal test reads them)
		cube.device.expand12(ins[1], c, torch.empty(12 * n, dtype=torch.uint8, device="cuda"))
	flags = torch.empty(12 * n, dtype=torch.uint8, device="cuda")
	ohs = [torch.empty((n, 288), dtype=torch.float32, device="cuda") for _ in range(3)]
	k = [0]

	def rot(fn):
		def run():
			k[0] += 1
			fn(k[0])
		return run

	for name, nbytes, same, neutral in [
		("686 multi_rotate", (288 + 1 + 288) * n,
		 lambda: cube.device.multi_rotate(states, acts, outs[0]), rot(lambda i: cube.device.multi_rotate(ins[i % N_IN], acts, outs[i % 4]))),
		("686 expand12 + goal test", (288 + 12 * 288 + 12) * n,
		 lambda: cube.device.expand12(states, children[0], flags), rot(lambda i: cube.device.expand12(ins[i % N_IN], children[i % 2], flags))),
		("686 multi_is_solved (12 n rows)", (288 + 1) * 12 * n,
		 lambda: cube.device.multi_is_solved(children[0], flags), rot(lambda i: cube.device.multi_is_solved(children[i % 2], flags))),
		("686 as_oh f32", (288 + 1152) * n,
		 lambda: cube.device.as_oh(states, ohs[0]), rot(lambda i: cube.device.as_oh(ins[i % N_IN], ohs[i % 3]))),
	]:
		t_same, t = timed(same, 30), timed(neutral, 3 * N_IN)
		print(json.dumps({"kernel": name, "states": n, "ms": t * 1e3, "GB/s": round(nbytes / t / 1e9, 1), "frac_of_8TBs": round(nbytes / t / 8e12, 4),
		                  "protocol": f"cache-neutral: inputs over {N_IN} sets, outputs rotating",
		                  "same_buffers_ms": t_same * 1e3, "same_buffers_frac_of_8TBs": round(nbytes / t_same / 8e12, 4)}), flush=True)


if __name__ == "__main__":
	main()
