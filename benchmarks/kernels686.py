"""Roofline sweep of the 6x8x6-representation kernels (288-byte states) on one MI355X."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librubiks_amd import _ffi, cube  # noqa: E402
from benchmarks.kernels import timed  # noqa: E402


def main(n=200_000):
	_ffi.check(_ffi.lib().rk_init(0))
	cube.set_is2024(False)
	solved = torch.from_numpy(cube.get_solved()).cuda()
	states = solved.unsqueeze(0).repeat(n, 1, 1, 1).contiguous()
	g = torch.Generator(device="cuda"); g.manual_seed(0)
	for _ in range(12):
		acts = torch.randint(0, 12, (n,), device="cuda", dtype=torch.uint8, generator=g)
		states = cube.device.multi_rotate(states, acts)
	out = torch.empty_like(states)
	children = torch.empty((12 * n, 6, 8, 6), dtype=torch.int8, device="cuda")
	flags = torch.empty(12 * n, dtype=torch.uint8, device="cuda")
	oh = torch.empty((n, 288), dtype=torch.float32, device="cuda")
	for name, nbytes, fn in [
		("686 multi_rotate", (288 + 1 + 288) * n, lambda: cube.device.multi_rotate(states, acts, out)),
		("686 expand12 + goal test", (288 + 12 * 288 + 12) * n, lambda: cube.device.expand12(states, children, flags)),
		("686 multi_is_solved (12 n rows)", (288 + 1) * 12 * n, lambda: cube.device.multi_is_solved(children, flags)),
		("686 as_oh f32", (288 + 1152) * n, lambda: cube.device.as_oh(states, oh)),
	]:
		t = timed(fn, 30)
		print(json.dumps({"kernel": name, "ms": t * 1e3, "GB/s": round(nbytes / t / 1e9, 1), "frac_of_8TBs": round(nbytes / t / 8e12, 4), "states": n}), flush=True)


if __name__ == "__main__":
	main()
