"""
The fan-out kernel in the cache regimes the round-2 verdict asks about, as ONE command for rocprofv3 (`--kernel-trace`
alone for durations, or `--pmc <counters> --kernel-trace` for one counter group per pass -- never both kinds at once):

    rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum --kernel-trace --output-format csv -d DIR -- python3 benchmarks/pmc_regimes.py

Phases (each a run of back-to-back rk_expand12 launches; the manifest written to --manifest lists them in order so that
benchmarks/pmc_regimes_summary.py can slice the kernel's dispatches):

  1M_in4     1 M parents, parents rotating over 4 sets (80 MB: may stay in the 256 MiB Infinity Cache), outputs over 4 sets
  1M_in32    1 M parents, parents rotating over 32 sets (640 MB of distinct input: cannot be cache hits), outputs over 4 sets
  8M_same    8 M parents, the same 160 MB input every launch (fits the Infinity Cache if the stores do not evict it)
  8M_in4     8 M parents, parents rotating over 4 sets (640 MB: from HBM)
  16M_same   16 M parents, the same 320 MB input every launch (does not fit)
  32M_same   32 M parents, the same 640 MB input every launch
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librubiks_amd import _ffi, cube  # noqa: E402


def parents(n, seed):
	g = torch.Generator(device="cuda")
	g.manual_seed(seed)
	acts = torch.randint(0, 12, (20, n), device="cuda", dtype=torch.uint8, generator=g)
	return cube.device.apply_sequences(acts, False, True)


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--manifest", default="gpurun_out/pmc_regimes_manifest.json")
	ap.add_argument("--phases", default="1M_in4,1M_in32,8M_same,8M_in4,16M_same,32M_same")
	args = ap.parse_args()
	_ffi.check(_ffi.lib().rk_init(0))
	manifest = []
	ev = lambda: torch.cuda.Event(enable_timing=True)

	def phase(name, n, n_in, n_out, launches, warm):
		ins = [parents(n, 77 + k) for k in range(n_in)]
		outs = [(torch.empty((12 * n, 20), dtype=torch.int8, device="cuda"), torch.empty(12 * n, dtype=torch.uint8, device="cuda"))
		        for _ in range(n_out)]
		torch.cuda.synchronize()
		e0, e1 = ev(), ev()
		for i in range(warm + launches):
			if i == warm:
				e0.record()
			c, f = outs[i % n_out]
			cube.device.expand12(ins[i % n_in], c, f)
		e1.record()
		torch.cuda.synchronize()
		ms = e0.elapsed_time(e1) / launches
		manifest.append({"phase": name, "parents": n, "input_sets": n_in, "output_sets": n_out, "warm": warm, "launches": launches,
		                 "ms_per_launch_hip_events": ms, "frac_of_8TBs": 272.0 * n / (ms * 1e-3) / 8e12})
		print(json.dumps(manifest[-1]), flush=True)
		del ins, outs
		torch.cuda.empty_cache()

	table = {
		"1M_in4": (1_000_000, 4, 4, 48, 8),
		"1M_in32": (1_000_000, 32, 4, 64, 8),
		"8M_same": (8_000_000, 1, 1, 8, 2),
		"8M_in4": (8_000_000, 4, 1, 8, 2),
		"16M_same": (16_000_000, 1, 1, 6, 2),
		"32M_same": (32_000_000, 1, 1, 4, 2),
	}
	for name in args.phases.split(","):
		phase(name, *table[name])
	os.makedirs(os.path.dirname(args.manifest) or ".", exist_ok=True)
	with open(args.manifest, "w") as f:
		json.dump(manifest, f, indent=1)


if __name__ == "__main__":
	main()
