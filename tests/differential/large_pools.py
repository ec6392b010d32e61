"""
Pools at the sizes the agents allow by default, once: an A* search from the default 4 M-state pool to a 32 M-state budget (three
growths in place, N = 10 000), and one MCTS tree from the default 200 000 nodes to 1.5 M (three growths).  Checked on the whole
pools, not on samples: no state stored twice, every parent link (A*) / neighbour link (MCTS) is a real move, G is consistent, the
open queue is sorted and holds each node once.  (The oracle cannot replay searches of this size; these are the structural
invariants of tests/test_agents.py:49-94, :122-134 of the reference.)

    python tests/differential/large_pools.py > profiles/r04_large_pools.json
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.nets import FastStub  # noqa: E402
from librubiks_amd.solving.agents import AStar, MCTS  # noqa: E402
from oracle import c_oracle, cube_oracle as orc  # noqa: E402


def n_unique(states: np.ndarray) -> int:
	"""distinct rows of an (n, 20) int8 array: sort three 64-bit words per row (the last one padded)"""
	n = len(states)
	padded = np.zeros((n, 24), np.uint8)
	padded[:, :20] = states.view(np.uint8)
	w = padded.view(np.uint64)
	order = np.lexsort((w[:, 2], w[:, 1], w[:, 0]))
	s = w[order]
	return int(1 + (np.any(s[1:] != s[:-1], axis=1)).sum()) if n else 0


np.random.seed(30)
start, _, _ = orc.scramble(40, True)
budget = 32_000_000
agent = AStar(FastStub(), 0.3, 10_000)                       # default pool: 4 M states
torch.cuda.synchronize()
t0 = time.perf_counter()
solved = agent.search(start, None, budget)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n = len(agent)
print(f"[large] A*: {n} states, {agent.iterations} iterations, grown {agent.grown}x, {dt:.2f} s", file=sys.stderr, flush=True)
st, G, par, act = agent.states, agent.G, agent.parents, agent.parent_actions
assert not solved and agent.grown == 3 and budget - 120_000 < n <= budget
assert n_unique(st[1:n + 1]) == n, "a state is stored twice"
print("[large] A*: states distinct", file=sys.stderr, flush=True)
for lo in range(2, n + 1, 4_000_000):
	pick = np.arange(lo, min(n + 1, lo + 4_000_000))
	moved = c_oracle.multi_rotate(st[par[pick]], act[pick].astype(np.uint8), threads=8)
	assert (moved == st[pick]).all(), "a parent link is not a move"
	assert (G[pick] >= G[par[pick]] + 1).all() and (par[pick] >= 1).all() and (par[pick] <= n).all()
assert G[1] == 0
print("[large] A*: links and G checked", file=sys.stderr, flush=True)
from librubiks_amd import _ffi  # noqa: E402
n_open = int(_ffi.lib().rk_astar_open_size(agent._h))
costs, idx = np.zeros(n_open), np.zeros(n_open, np.int64)
q = range(_ffi.lib().rk_astar_export_open(agent._h, costs.ctypes.data, idx.ctypes.data, n_open, _ffi.stream_ptr()))
costs, idx = costs[:len(q)], idx[:len(q)]
assert len(q) == n_open and (np.diff(costs) >= 0).all() and len(np.unique(idx)) == len(idx) and idx.min() >= 1 and idx.max() <= n
out = {"bench": "large_pools", "astar": {"N": 10_000, "budget": budget, "states": n, "iterations": agent.iterations, "grown_in_place": agent.grown,
                                          "seconds": dt, "states_per_s": n / dt, "open_queue": len(q),
                                          "checked": "all states distinct; every parent link a move; G consistent; open queue sorted, each node once"}}
del st, G, par, act, q, costs, idx, agent
torch.cuda.empty_cache()

print("[large] A*: open queue checked", file=sys.stderr, flush=True)
np.random.seed(31)
start, _, _ = orc.scramble(40, True)
budget = 1_500_000
tree = MCTS(FastStub(), 0.6, False, priors="kernel")          # default pool: 200 000 nodes
torch.cuda.synchronize()
t0 = time.perf_counter()
solved = tree.search(start, None, budget)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n = len(tree)
print(f"[large] MCTS: {n} nodes, {tree._batch.simulations} simulations, grown {tree.grown}x, {dt:.2f} s", file=sys.stderr, flush=True)
a = tree._export()
assert not solved and tree.grown == 3 and budget - 12 < n <= budget
stt, nb, leaves = a["states"], a["neighbors"], a["leaves"]
assert n_unique(stt[1:n + 1]) == n, "a node is stored twice"
src, actn = np.nonzero(nb[1:n + 1] > 0)
src += 1
for lo in range(0, len(src), 4_000_000):
	s_, a_ = src[lo:lo + 4_000_000], actn[lo:lo + 4_000_000]
	assert (c_oracle.multi_rotate(stt[s_], a_.astype(np.uint8), threads=8) == stt[nb[s_, a_]]).all(), "a neighbour link is not a move"
assert (nb[1:n + 1].all(axis=1) != leaves[1:n + 1]).all()                                    # tests/test_agents.py:88-89
assert int(a["N"][1:n + 1].sum()) > 0 and (a["N"][1:n + 1] >= 0).all()
out["mcts"] = {"budget": budget, "nodes": n, "simulations": int(tree._batch.simulations), "grown_in_place": tree.grown, "seconds": dt,
               "links_checked": int(len(src)), "checked": "all nodes distinct; every neighbour link a move; leaves = nodes with a missing neighbour"}
print(json.dumps(out), flush=True)
