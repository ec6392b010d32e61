"""Engines fail loudly: capacity, call-order and argument errors surface as negative codes / RubiksHipError."""
import ctypes as C

import numpy as np
import pytest
import torch

from librubiks_amd import _ffi, cube
from librubiks_amd.solving.agents import AStar, MCTSBatch
from oracle import cube_oracle as orc
from oracle.search_oracle import StubNet

pytestmark = pytest.mark.gpu


def test_astar_engine_state_machine_and_capacity():
	lib, st = _ffi.lib(), None
	h = C.c_void_p()
	assert lib.rk_astar_create(C.byref(h), 1, 10) == -1            # capacity too small
	_ffi.check(lib.rk_astar_create(C.byref(h), 60, 4))
	info = (C.c_longlong * 5)()
	assert lib.rk_astar_expand(h, 4, info, st) == -4               # not reset yet
	start = orc.rotate(orc.rotate(orc.SOLVED, 1, 1), 4, 0)
	_ffi.check(lib.rk_astar_reset(h, start.ctypes.data, 0.5, st))
	assert lib.rk_astar_expand(h, 5, info, st) == -1               # more than max_expansions
	_ffi.check(lib.rk_astar_expand(h, 4, info, st))
	assert list(info)[:2] == [1, 12] and lib.rk_astar_size(h) == 13
	assert lib.rk_astar_expand(h, 4, info, st) == -4               # pending expansion not committed
	vals = torch.zeros(12, dtype=torch.float32, device="cuda")
	_ffi.check(lib.rk_astar_commit(h, vals.data_ptr(), st))
	assert lib.rk_astar_commit(h, vals.data_ptr(), st) == -4       # nothing pending
	assert lib.rk_astar_expand(h, 4, info, st) == -3 and b"capacity" in lib.rk_last_error()   # 13 + 48 > 60
	assert lib.rk_astar_path(h, 0, None, 0, st) == -1
	_ffi.check(lib.rk_astar_destroy(h))


def test_astar_agent_budget_is_respected():
	np.random.seed(1)
	start, _, _ = orc.scramble(12, True)
	for budget, n in ((500, 7), (1300, 100), (5000, 64)):
		agent = AStar(StubNet(), 0.3, n)
		solved = agent.search(start, None, budget)
		assert len(agent) <= budget and (solved or len(agent) + 12 * n > budget)


def test_mcts_path_overflow_is_reported():
	np.random.seed(3)
	start, _, _ = orc.scramble(15, True)
	agent = MCTSBatch(StubNet(), 0.1, 2, capacity=3000, max_path=6)    # far too short for the descents
	with pytest.raises(_ffi.RubiksHipError, match="max_path"):
		agent.search(np.array([start, start]), max_states=3000, poll=4)
	lib = _ffi.lib()
	h = C.c_void_p()
	assert lib.rk_mcts_create(C.byref(h), 0, 100, 100) == -1
	assert lib.rk_mcts_create(C.byref(h), 1, 5, 100) == -1
	_ffi.check(lib.rk_mcts_create(C.byref(h), 1, 100, 100))
	assert lib.rk_mcts_expand(h, None) == -4                        # not reset
	_ffi.check(lib.rk_mcts_destroy(h))


def test_mcts_budget_and_done_trees_are_frozen():
	"""A tree that ran out of budget or solved must not change while the others keep going."""
	np.random.seed(8)
	starts = np.array([orc.scramble(d, True)[0] for d in (1, 9, 9)])
	agent = MCTSBatch(StubNet(), 5.0, 3, capacity=2000)
	solved = agent.search(starts, max_states=np.array([2000, 100, 2000]), poll=8)
	st = agent.status
	assert solved[0] and st[0, 3] == 1                              # one simulation solves a depth-1 scramble
	assert not solved[1] and st[1, 2] <= 100 and st[1, 2] + 12 > 100
	snap = agent.tree_arrays(1)
	agent2 = MCTSBatch(StubNet(), 5.0, 1, capacity=2000)
	agent2.search(starts[1:2], max_states=100)
	alone = agent2.tree_arrays(0)
	for k in ("states", "neighbors", "N", "W", "L", "leaves"):
		assert (snap[k] == alone[k]).all(), k


def test_astar_batch_argument_and_state_errors():
	lib = _ffi.lib()
	h = C.c_void_p()
	assert lib.rk_astarb_create(C.byref(h), 0, 1000, 10) == -1
	assert lib.rk_astarb_create(C.byref(h), 2, 100, 10) == -1 and b"12 * expansions" in lib.rk_last_error()
	_ffi.check(lib.rk_astarb_create(C.byref(h), 2, 5000, 10))
	oh = torch.empty((2 * 120, 480), dtype=torch.float32, device="cuda")
	assert lib.rk_astarb_step_expand(h, oh.data_ptr(), 0, None) == -4              # not reset
	np.random.seed(77)
	starts = np.array([orc.scramble(15, True)[0] for _ in range(2)])
	_ffi.check(lib.rk_astarb_reset(h, starts.ctypes.data, None, 0.5, None))
	vals = torch.zeros(240, dtype=torch.float32, device="cuda")
	assert lib.rk_astarb_step_commit(h, vals.data_ptr(), None) == -4               # nothing pending
	for it in range(8):
		_ffi.check(lib.rk_astarb_step_expand(h, oh.data_ptr(), 0, None))
		assert lib.rk_astarb_step_expand(h, oh.data_ptr(), 0, None) == -4          # pending
		_ffi.check(lib.rk_astarb_step_commit(h, vals.data_ptr(), None))
	st = np.zeros((2, 7), np.int64)
	_ffi.check(lib.rk_astarb_status(h, st.ctypes.data, None))
	assert (st[:, 3] == 8).all() and (st[:, 2] > 8 * 60).all() and not st[:, 6].any(), st     # eight iterations each, no error
	assert lib.rk_astarb_set_values_dtype(h, 1, None) == -1 and lib.rk_astarb_export(h, 5, 1, 1, None, None, None, None, None) == -1
	_ffi.check(lib.rk_astarb_destroy(h))


def test_time_limited_search_grows_its_pool_or_says_so():
	"""A search limited only by time must not stop silently on its default pool (the reference grows its arrays): the pool
	doubles IN PLACE while time is left (rk_astar_grow / rk_mcts_grow; tests/test_growth_gpu.py holds the grown search to the
	arrays of one that started large); at max_capacity the agent warns and flags it."""
	from librubiks_amd.solving.agents import CapacityExhausted, MCTS
	np.random.seed(6)
	start, _, _ = orc.scramble(30, True)
	agent = AStar(StubNet(), 0.3, 50, capacity=3000)
	agent.max_capacity = 12_000
	with pytest.warns(CapacityExhausted):
		assert agent.search(start, time_limit=20) is False
	assert agent.capacity_exhausted and 6_000 < len(agent) <= 12_000 and agent.grown == 2      # grew 3000 -> 6000 -> 12000, then gave up
	agent.max_capacity = 10_000_000
	assert agent.search(start, time_limit=None, max_states=5_000) is False and not agent.capacity_exhausted   # a budget is not a full pool
	assert len(agent) <= 5_000
	tree = MCTS(StubNet(), 1.0, False, capacity=1000)
	tree.max_capacity = 4000
	with pytest.warns(CapacityExhausted):
		assert tree.search(start, time_limit=20) is False
	assert tree.capacity_exhausted and 2000 < len(tree) <= 4000 and tree.grown == 2
