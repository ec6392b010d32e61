import sys, time, json, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librubiks_amd import cube
from librubiks_amd.solving.agents import AStar
from oracle.search_oracle import StubNet
from benchmarks.nets import FcSmall
net = FcSmall().cuda().eval()
for name, nn in (("stub", StubNet()), ("fc_small fp32", net)):
    for N in (10, 27, 100, 700):
        agent = AStar(nn, 0.2, N)
        np.random.seed(3); state, _, _ = cube.scramble(16, True)
        agent.search(state, None, 3000)   # warm
        torch.cuda.synchronize(); t0 = time.perf_counter()
        agent.search(state, None, 200_000)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(json.dumps({"net": name, "N": N, "iterations": agent.iterations, "states": len(agent), "us_per_iteration": dt / agent.iterations * 1e6, "states_per_s": len(agent) / dt}))
