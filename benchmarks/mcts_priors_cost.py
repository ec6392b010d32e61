"""
What the drop-in `MCTS`'s default priors cost: `priors="reference"` computes softmax(logits) where the reference does -- the root's on
the device, every other node's on the HOST (`p.cpu().softmax(dim=1)`, agents.py:551-552) -- which is one device-to-host wait per
simulation; `priors="kernel"` (softmax inside the backup kernel, bitwise torch.softmax on the device) has none and can be replayed
as a hipGraph.  One tree, random-init fc_small, depth-14 scramble, 30 000 states.  One JSON object per mode.
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks.nets import FcSmall  # noqa: E402
from librubiks_amd import cube  # noqa: E402
from librubiks_amd.solving.agents import MCTS  # noqa: E402

for dtype in (torch.float32, torch.bfloat16):
	net = FcSmall().cuda().eval().to(dtype)
	np.random.seed(5)
	state, _, _ = cube.scramble(14, True)
	for priors, graph in (("reference", False), ("torch", False), ("kernel", False), ("kernel", True)):
		agent = MCTS(net, 0.6, False, capacity=40_000, use_hipgraph=graph, priors=priors)
		agent.search(state, None, 2_000)
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		agent.search(state, None, 30_000)
		torch.cuda.synchronize()
		dt = time.perf_counter() - t0
		sims = int(agent._batch.status[0, 3])
		print(json.dumps({"bench": "mcts_priors_cost", "net": f"fc_small {str(dtype).split('.')[-1]}", "priors": priors, "hipgraph": graph, "simulations": sims,
		                  "states": len(agent), "us_per_simulation": dt / max(sims, 1) * 1e6, "simulations_per_s": sims / dt}), flush=True)
