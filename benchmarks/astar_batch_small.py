import os, sys, time, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librubiks_amd import cube
from librubiks_amd.solving.agents import AStar, AStarBatch
from benchmarks.nets import FastStub as StubNet      # the exact stub heuristic, without touching oracle/
from benchmarks.nets import FcSmall
net = FcSmall().cuda().eval()
np.random.seed(3); state, _, _ = cube.scramble(16, True)
for name, nn in (("stub", StubNet()), ("fc_small fp32", net)):
    for N in (10, 100):
        a = AStar(nn, 0.2, N); a.search(state, None, 3000)
        torch.cuda.synchronize(); t0 = time.perf_counter(); a.search(state, None, 100_000); torch.cuda.synchronize(); t_single = time.perf_counter() - t0
        row = {"net": name, "N": N, "AStar us/iter": t_single / a.iterations * 1e6, "iters": a.iterations}
        for graph in (False, True):
            b = AStarBatch(nn, 0.2, N, 1, capacity=100_000); b.search(state[None], max_states=3000, use_graph=graph, poll=32)
            torch.cuda.synchronize(); t0 = time.perf_counter(); b.search(state[None], max_states=100_000, use_graph=graph, poll=64); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            row[f"AStarBatch(1){' graph' if graph else ''} us/iter"] = dt / int(b.status[0, 3]) * 1e6
        print(json.dumps(row), flush=True)
