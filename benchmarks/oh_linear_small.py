"""
The fused first layer (f1) on SEARCH-STEP batches: the two forms of the MFMA route side by side -- "mfma_direct" (one 32 x 32
output tile per wave, weights straight from a fragment-major copy in global memory) against "mfma_tiled" (a 64-column weight
tile resident in LDS) -- from 12 rows (one MCTS tree's step) to 12 000 (an A* iteration at N = 1000), with the epilogue the
folded net uses (ELU + BatchNorm affine).  Both forms must give the same bits; the crossover sets OHL_DIRECT_MAX_ROWS in
csrc/rk_oh_linear.hip.  `--gemms` also times the net's torch GEMMs behind the layer at the same batches, as hipBLASLt's heuristic
picks them and as torch.cuda.tunable picks them (for the record: which part of a step is whose).

    python benchmarks/oh_linear_small.py [--gemms] > profiles/r05_oh_linear_small.json
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ROWS = [12, 120, 256, 512, 768, 1024, 1344, 1536, 2048, 3072, 4096, 12000]


def timed(fn, launches=200, reps=5):
	"""median over reps of (HIP events around `launches` back-to-back calls) / launches, microseconds"""
	for _ in range(10):
		fn()
	out = []
	for _ in range(reps):
		a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
		torch.cuda.synchronize()
		a.record()
		for _ in range(launches):
			fn()
		b.record()
		torch.cuda.synchronize()
		out.append(a.elapsed_time(b) / launches * 1e3)
	return sorted(out)[len(out) // 2]


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--gemms", action="store_true")
	ap.add_argument("--H", type=int, default=4096)
	args = ap.parse_args()
	from librubiks_amd.oh_linear import OhLinear
	from tests.helpers import random_walk
	torch.manual_seed(0)
	H = args.H
	lin = torch.nn.Linear(480, H).cuda().to(torch.bfloat16)
	bn = torch.nn.BatchNorm1d(H).cuda().eval()
	with torch.no_grad():
		bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0); bn.weight.normal_(); bn.bias.normal_()
	layer = OhLinear(lin, route="mfma").set_epilogue(torch.nn.ELU(), bn)
	rec = {"H": H, "epilogue": "ELU + BatchNorm affine", "unit": "us per launch (median of 5 x 200 back-to-back launches, HIP events)", "rows": {}}
	for n in ROWS:
		st = torch.from_numpy(random_walk(n, 24, seed=n)).cuda()
		outs = {r: torch.empty((n, H), dtype=torch.bfloat16, device="cuda") for r in ("mfma_direct", "mfma_tiled", "mfma")}
		row = {}
		for r, o in outs.items():
			row[r] = timed(lambda: layer(st, out=o, route=r))
		row["same_bits"] = bool(torch.equal(outs["mfma_direct"].view(torch.int16), outs["mfma_tiled"].view(torch.int16))
		                        and torch.equal(outs["mfma"].view(torch.int16), outs["mfma_tiled"].view(torch.int16)))
		rec["rows"][str(n)] = row
		print(n, row, file=sys.stderr, flush=True)
	if args.gemms:
		shapes = {"layer 2 (4096 -> 2048)": (4096, 2048), "heads 1 (2048 -> 1024)": (2048, 1024), "heads 2 (1024 -> 13)": (1024, 13)}
		mods = {k: torch.nn.Linear(i, o).cuda().to(torch.bfloat16) for k, (i, o) in shapes.items()}
		gem = {}
		for n in (12, 256, 1344, 3072, 12000):
			xs = {k: torch.randn(n, i, device="cuda", dtype=torch.bfloat16) for k, (i, o) in shapes.items()}
			with torch.no_grad():
				gem[str(n)] = {k + ", heuristic": timed(lambda: mods[k](xs[k])) for k in shapes}
		import torch.cuda.tunable as tn
		tn.enable(True)
		tn.tuning_enable(True)
		tn.set_max_tuning_duration(15)
		tn.set_max_tuning_iterations(20)
		try:
			tn.write_file_on_exit(False)
		except Exception:
			pass
		import time
		for n in (12, 256, 1344, 3072, 12000):
			xs = {k: torch.randn(n, i, device="cuda", dtype=torch.bfloat16) for k, (i, o) in shapes.items()}
			with torch.no_grad():
				for k in shapes:
					t0 = time.perf_counter()
					mods[k](xs[k])
					torch.cuda.synchronize()
					gem[str(n)][k + ", tuning seconds"] = time.perf_counter() - t0
					gem[str(n)][k + ", tunable"] = timed(lambda: mods[k](xs[k]))
			print(n, gem[str(n)], file=sys.stderr, flush=True)
		rec["gemms_us"] = gem
	print(json.dumps(rec))


if __name__ == "__main__":
	main()
