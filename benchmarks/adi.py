"""
ADI data generation (row f2) at the reference's training size (configs/main_train.ini:4-5: 7 500 games x depth 30 = 225 000
states -> 2.7 M children per rollout; librubiks/train.py:256-339) on one MI355X with the random-init fc_small net:
seconds per `adi_traindata` call, split into the cube kernels and the value net, for the net as it is (one-hot rows +
torch), with the first layer reading the 20-byte states, and with its ELU + BatchNorm in the kernel epilogue and the other
BatchNorm layers folded.  The reference runs the same function on CPU NumPy + a CUDA/CPU torch net.

    python benchmarks/adi.py > profiles/r02_adi.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks.nets import FcSmall  # noqa: E402
from librubiks_amd import cube  # noqa: E402
from librubiks_amd.adi import adi_traindata  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--games", type=int, default=7500)
ap.add_argument("--depth", type=int, default=30)
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()

MODES = {False: "one-hot rows + torch net", True: "first layer fused (rk_ohl)",
         "folded": "first layer fused with ELU + BatchNorm epilogue, other BatchNorm layers folded"}
for dtype, name in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
	net = FcSmall().cuda().eval().to(dtype)
	for mode, note in MODES.items():
		ff = 8 if dtype == torch.float32 and mode is False else 2            # the (rows, 480) float32 one-hot batch is the memory hog
		np.random.seed(0)
		adi_traindata(net, 200, args.depth, 0.5, "lapanfix", ff_batches=ff, fused_first_layer=mode)   # warm-up
		torch.cuda.synchronize()
		times = []
		for rep in range(args.reps):
			np.random.seed(rep)
			t0 = time.perf_counter()
			oh, pol, val, w = adi_traindata(net, args.games, args.depth, 0.5, "lapanfix", ff_batches=ff, fused_first_layer=mode)
			torch.cuda.synchronize()
			times.append(time.perf_counter() - t0)
		n = args.games * args.depth
		# the cube part alone: scramble walks, fan-out + goal test, one-hot of the scrambled states
		acts = torch.randint(0, 12, (args.depth, args.games), device="cuda", dtype=torch.uint8)
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		for _ in range(10):
			states = cube.device.apply_sequences(acts, True, False)
			cube.device.as_oh(states)
			cube.device.multi_is_solved(states)
			cube.device.expand12(states)
		torch.cuda.synchronize()
		cube_s = (time.perf_counter() - t0) / 10
		best = min(times)
		print(json.dumps({"bench": "adi_traindata", "games": args.games, "depth": args.depth, "states": n, "children": 12 * n,
		                  "net": f"fc_small random init {name}", "mode": note, "ff_batches": ff, "seconds_best": best,
		                  "seconds_all": times, "children_per_s": 12 * n / best, "cube_kernels_seconds": cube_s,
		                  "cube_share": cube_s / best}), flush=True)
		del oh, pol, val, w
	del net
	torch.cuda.empty_cache()
