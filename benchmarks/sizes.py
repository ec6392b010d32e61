"""Fan-out throughput against batch size (and a plain device fill of the same output size as the reference point)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librubiks_amd import cube  # noqa: E402
from benchmarks.kernels import timed  # noqa: E402

for n in (100_000, 1_000_000, 4_000_000, 8_000_000, 16_000_000, 32_000_000):
	g = torch.Generator(device="cuda")
	g.manual_seed(1)
	acts = torch.randint(0, 12, (20, n), device="cuda", dtype=torch.uint8, generator=g)
	parents = cube.device.apply_sequences(acts, False, True)
	del acts
	ch = torch.empty((12 * n, 20), dtype=torch.int8, device="cuda")
	fl = torch.empty(12 * n, dtype=torch.uint8, device="cuda")
	reps = 20 if n > 1_000_000 else 100
	t = timed(lambda: cube.device.expand12(parents, ch, fl), reps)
	tf = timed(lambda: ch.fill_(3), reps)
	print(json.dumps({"parents": n, "ms": t * 1e3, "GB/s": round(272 * n / t / 1e9, 1), "frac": round(272 * n / t / 8e12, 4),
	                  "expansions/s": n / t, "torch_fill_same_output_GB/s": round(240 * n / tf / 1e9, 1)}), flush=True)
	del parents, ch, fl
	torch.cuda.empty_cache()
