"""One A* configuration for `rocprofv3 --kernel-trace --stats`: stub net (engine cost only) or fc_small, eager or hipGraph.

    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 benchmarks/astar_profile.py --expansions 100 --net stub
"""
import argparse
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

ap = argparse.ArgumentParser()
ap.add_argument("--expansions", type=int, default=100)
ap.add_argument("--net", default="stub")
ap.add_argument("--graph", type=int, default=0)
ap.add_argument("--max-states", type=int, default=200_000)
ap.add_argument("--depth", type=int, default=16)
a = ap.parse_args()
net = FastStub() if a.net == "stub" else (FcSmall().cuda().eval().to(torch.bfloat16) if a.net == "bf16" else FcSmall().cuda().eval())
agent = AStar(net, 0.2, a.expansions, poll=16, use_hipgraph=bool(a.graph))
np.random.seed(3)
state, _, _ = cube.scramble(a.depth, True)
agent.search(state, None, 3000 + 12 * a.expansions)
torch.cuda.synchronize()
t0 = time.perf_counter()
agent.search(state, None, a.max_states)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"net": a.net, "N": a.expansions, "hipgraph": bool(a.graph), "iterations": agent.iterations, "states": len(agent),
                  "us_per_iteration": dt / max(agent.iterations, 1) * 1e6}))
