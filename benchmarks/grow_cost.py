"""
What growing a pool in place costs ("a growth at 4 M nodes costs < 5 ms"): rk_astar_grow 4 M -> 8 M states and
rk_mcts_grow 200 k -> 400 k nodes of one tree, each on a pool that is about as full as it is when an agent grows it, timed
around the call (it synchronises), median of a few engines.  Prints one JSON object per engine kind.
"""
import ctypes as C
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks.nets import FastStub  # noqa: E402
from librubiks_amd import _ffi, cube  # noqa: E402
from librubiks_amd.solving.agents import AStar, MCTSBatch  # noqa: E402

lib = _ffi.lib()
_ffi.check(lib.rk_init(0))
np.random.seed(3)
start, _, _ = cube.scramble(40, True)

times, filled = [], []
for rep in range(3):
	agent = AStar(FastStub(), 0.3, 1000, capacity=4_000_000)
	agent.max_capacity = 4_000_000
	import warnings
	with warnings.catch_warnings():
		warnings.simplefilter("ignore")
		agent.search(start, time_limit=60)                                   # fills the pool: stops at the loop guard, max_capacity reached
	filled.append(len(agent))
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	_ffi.check(lib.rk_astar_grow(agent._h, 8_000_000, _ffi.stream_ptr()))
	times.append((time.perf_counter() - t0) * 1e3)
	agent._h_cap = 8_000_000
	del agent
print(json.dumps({"bench": "rk_astar_grow", "from": 4_000_000, "to": 8_000_000, "states_stored": filled, "ms": times, "ms_median": statistics.median(times),
                  "what": "new arrays (hipMalloc), 136 MB of device-to-device copies, hash table cleared and rebuilt by one kernel, old arrays freed, stream synchronised"}), flush=True)

times, filled = [], []
for rep in range(3):
	trees = MCTSBatch(FastStub(), 1.0, 1, capacity=200_000)
	trees.search(start[None], max_states=200_000, poll=64)
	filled.append(int(trees.status[0, 2]))
	budgets = np.array([400_000], np.int64)
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	_ffi.check(lib.rk_mcts_grow(trees._h, 400_000, trees.max_path, budgets.ctypes.data, _ffi.stream_ptr()))
	times.append((time.perf_counter() - t0) * 1e3)
	del trees
print(json.dumps({"bench": "rk_mcts_grow", "trees": 1, "from": 200_000, "to": 400_000, "nodes_stored": filled, "ms": times, "ms_median": statistics.median(times),
                  "what": "new node records (512 B each), states and hash table; copies; rehash kernel; budgets; old arrays freed; stream synchronised"}), flush=True)
