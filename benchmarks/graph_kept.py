"""
What a search costs before its first iteration when the step is replayed as a hipGraph: the first search of an agent captures
(side stream, warm iteration, capture, instantiation), later searches on the unchanged engine and net replay the kept graph.
Short searches (easy scrambles, the `/solve` of an almost solved cube) are where that shows.

    python benchmarks/graph_kept.py > profiles/r04_graph_kept.json
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
from librubiks_amd.solving.agents import AStar, MCTS  # noqa: E402

bf16 = FcSmall().cuda().eval().to(torch.bfloat16)
for name, make in (("AStar N=10 stub, hipGraph", lambda: AStar(FastStub(), 0.2, 10, capacity=100_000, use_hipgraph=True)),
                   ("AStar N=10 fc_small bf16 fused + folded, hipGraph", lambda: AStar(bf16, 0.2, 10, capacity=100_000, use_hipgraph=True, fused_first_layer="folded")),
                   ("AStar N=10 stub, eager", lambda: AStar(FastStub(), 0.2, 10, capacity=100_000)),
                   ("MCTS stub, hipGraph (priors in the kernel)", lambda: MCTS(FastStub(), 0.6, False, capacity=20_000, use_hipgraph=True)),
                   ("MCTS stub, eager (priors in the kernel)", lambda: MCTS(FastStub(), 0.6, False, capacity=20_000, priors="kernel"))):
	agent = make()
	np.random.seed(1)
	agent.search(cube.scramble(3, True)[0], None, 500)              # library handles, allocator
	agent = make()
	ms, its = [], []
	for i in range(12):
		np.random.seed(100 + i)
		start = cube.scramble(5, True)[0]
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		agent.search(start, None, 3_000)
		torch.cuda.synchronize()
		ms.append((time.perf_counter() - t0) * 1e3)
		its.append(int(getattr(agent, "iterations", 0)) or int(agent._batch.simulations))
	caps = agent.captures if hasattr(agent, "captures") else agent._batch.captures
	print(json.dumps({"agent": name, "searches": len(ms), "captures": caps, "first_search_ms": ms[0], "later_searches_ms_median": float(np.median(ms[1:])),
	                  "ms": [round(x, 3) for x in ms], "steps": its}), flush=True)
