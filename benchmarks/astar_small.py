"""
A* iteration floor on one MI355X: microseconds per iteration at the reference's N = 10 ... 700 with the exact stub net (engine
cost only) and the random-init fc_small net; eager launches (host polls every 4 iterations) against the same iteration replayed
as a hipGraph.  Round 1 (about 16 launches + one host sync per iteration): 115-150 us per iteration at N = 100 with the stub.

    python benchmarks/astar_small.py [--net-ends-ab] > profiles/r02_astar_small.json
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks.nets import FastStub, FcSmall  # noqa: E402
from librubiks_amd import cube  # noqa: E402
from librubiks_amd.solving.agents import AStar  # noqa: E402

from librubiks_amd import oh_linear  # noqa: E402

net = FcSmall().cuda().eval()
bf16 = FcSmall().cuda().eval().to(torch.bfloat16)
FOLDED = "fc_small bf16, first layer fused with epilogue, BatchNorm folded"
CASES = (("stub", FastStub(), False, None), ("fc_small fp32", net, False, None), ("fc_small bf16", bf16, False, None), (FOLDED, bf16, "folded", None))
if "--net-ends-ab" in sys.argv:
	# round 5: the net's two ends as they were (first layer always in its LDS-tiled form, the heads' last layer as torch's activation
	# kernel + narrow GEMM) against what ships (direct form for these batches, rk_tail_linear), same box, same searches
	CASES = ((FOLDED + " [first layer tiled, torch tail]", bf16, "folded", ("mfma_tiled", False)), (FOLDED + " [direct form, rk_tail_linear]", bf16, "folded", (None, True)))
for name, nn, fused, ends in CASES:
	oh_linear.MFMA_FORM, oh_linear.FUSE_TAIL = ends if ends is not None else (None, True)
	for N in (10, 27, 100, 700):
		for graph in (False, True):
			agent = AStar(nn, 0.2, N, poll=16, use_hipgraph=graph, fused_first_layer=fused)
			np.random.seed(3)
			state, _, _ = cube.scramble(16, True)
			agent.search(state, None, 3000 + 12 * N)   # warm
			torch.cuda.synchronize()
			t0 = time.perf_counter()
			agent.search(state, None, 200_000)
			torch.cuda.synchronize()
			dt = time.perf_counter() - t0
			print(json.dumps({"net": name, "N": N, "hipgraph": graph, "iterations": agent.iterations, "states": len(agent),
			                  "us_per_iteration": dt / max(agent.iterations, 1) * 1e6, "states_per_s": len(agent) / dt}), flush=True)
