"""
Search agents whose expand-children loops run on the MI355X (reference: librubiks/solving/agents.py).

`AStar` and `MCTS` keep the reference's constructor arguments, `search(state, time_limit, max_states) -> bool`
protocol, `action_queue`, `len(agent)` and inspection arrays (`states`, `G`, `parents`, `neighbors`, ...), but the
node pool, the state -> index map, the open queue and all per-iteration work live in HBM inside librubiks_hip.so
(engines `rk_astar_*` / `rk_mcts_*` of include/rubiks_hip.h).  The value/policy network stays a PyTorch module:
the engine writes the one-hot batch straight into a torch tensor and reads the net's output from one.

`net` is anything with `.eval()` and `__call__(x, policy=True, value=True)` returning logits (B, 12) and/or a value
(B, 1) for a float32 (B, 480) one-hot batch on `librubiks_amd.gpu` (reference: model.py:131-141).
"""
from __future__ import annotations

import ctypes as C
import time
from collections import deque

import numpy as np
import torch

from librubiks_amd import gpu, no_grad, _ffi
from librubiks_amd import cube


class Agent:
	"""Protocol of the reference's agents (agents.py:14-64)."""
	eps = np.finfo("float").eps
	_explored_states = 0

	def __init__(self):
		self.action_queue = deque()

	def search(self, state: np.ndarray, time_limit: float = None, max_states: int = None) -> bool:
		raise NotImplementedError

	def reset(self, time_limit: float, max_states: int):
		self._explored_states = 0
		self.action_queue = deque()
		if hasattr(self, "net"):
			self.net.eval()
		assert time_limit or max_states
		return time_limit or 1e10, max_states or int(1e10)

	def __len__(self):
		return self._explored_states


class DeepAgent(Agent):
	def __init__(self, net):
		super().__init__()
		self.net = net


def _value_f32(out) -> torch.Tensor:
	"""The net's value head as a contiguous float32 vector on the GPU."""
	if isinstance(out, (list, tuple)):
		out = out[-1]
	return out.detach().to(device=gpu, dtype=torch.float32).reshape(-1).contiguous()


class AStar(DeepAgent):
	"""
	Batch weighted A* (agents.py:171-413): expands the `expansions` cheapest open nodes per iteration,
	cost = lambda_ * G + (-value).  Same results as the reference (index numbering, G, parents, action_queue)
	whenever the net returns the same values.

	`capacity` bounds the number of stored states when a search is limited only by time (the reference grows its
	arrays without bound); with `max_states` given, exactly that budget is used.
	"""
	default_capacity = 4_000_000

	def __init__(self, net, lambda_: float, expansions: int, capacity: int = None):
		super().__init__(net)
		self.lambda_ = lambda_
		self.expansions = int(expansions)
		self.capacity = capacity
		self._h = None
		self._h_cap = 0
		self._n = 0
		self._root = None
		self._cache = None
		self.iterations = 0
		self.record_pops = False      # debugging aid: keep the popped indices of every iteration in self.pops
		self.pops = []

	# -- engine lifetime ------------------------------------------------------------------------------------
	def _engine(self, capacity: int):
		if self._h is not None and self._h_cap >= capacity:
			return self._h
		self._free()
		h = C.c_void_p()
		_ffi.check(_ffi.lib().rk_astar_create(C.byref(h), capacity, self.expansions))
		self._h, self._h_cap = h, capacity
		return h

	def _free(self):
		if getattr(self, "_h", None) is not None:
			_ffi.lib().rk_astar_destroy(self._h)
			self._h = None

	def __del__(self):
		try:
			self._free()
		except Exception:
			pass

	# -- search ---------------------------------------------------------------------------------------------
	def reset(self, time_limit: float, max_states: int):
		time_limit, max_states = super().reset(time_limit, max_states)
		self._n = 0
		self._cache = None
		self._root = None
		self.iterations = 0
		self.pops = []
		return time_limit, max_states

	@no_grad
	def search(self, state: np.ndarray, time_limit: float = None, max_states: int = None) -> bool:
		_ffi.require_gpu()
		t0 = time.perf_counter()
		time_limit, max_states = self.reset(time_limit, max_states)
		state = np.ascontiguousarray(state, dtype=np.int8)
		if cube.is_solved(state):
			return True
		cap = int(min(max_states, self.capacity or self.default_capacity))
		cap = max(cap, 12 * self.expansions + 2)
		h = self._engine(cap)
		lib, st = _ffi.lib(), _ffi.stream_ptr()
		_ffi.check(lib.rk_astar_reset(h, state.ctypes.data, float(self.lambda_), st))
		self._root, self._n = state.copy(), 1
		oh = torch.empty((12 * self.expansions, 480), dtype=torch.float32, device=gpu)
		info = (C.c_longlong * 5)()
		budget = min(max_states, cap)
		while time.perf_counter() - t0 < time_limit and self._n + self.expansions * cube.action_dim <= budget:
			if lib.rk_astar_open_size(h) == 0:
				break
			if self.record_pops:
				head = np.zeros(self.expansions, np.int64)
				got = lib.rk_astar_export_open(h, None, head.ctypes.data, self.expansions, st)
				self.pops.append(head[:got].copy())
			_ffi.check(lib.rk_astar_expand(h, self.expansions, info, st))
			n_new, won, solved_idx, self._n = int(info[1]), bool(info[2]), int(info[3]), int(info[4])
			self.iterations += 1
			if won:
				path = (C.c_longlong * 4096)()
				n = lib.rk_astar_path(h, solved_idx, path, 4096, st)
				if n < 0:
					_ffi.check(int(n))
				self.action_queue = deque(int(a) for a in path[:n])
				return True
			values = None
			if n_new:
				_ffi.check(lib.rk_astar_new_states_oh(h, oh.data_ptr(), _ffi.OH_F32, st))
				values = _value_f32(self.net(oh[:n_new], policy=False, value=True))
				assert values.numel() == n_new
			_ffi.check(lib.rk_astar_commit(h, values.data_ptr() if values is not None else None, _ffi.stream_ptr()))
		return False

	# -- inspection (what the reference's tests look at: tests/test_agents.py:96-145) --------------------------
	def _export(self):
		if self._cache is None:
			n = self._n
			states = np.zeros((n + 1, 20), np.int8)
			G = np.zeros(n + 1, np.float64)
			parents = np.zeros(n + 1, np.int64)
			pact = np.zeros(n + 1, np.int64)
			if n and self._h is not None:
				_ffi.check(_ffi.lib().rk_astar_export(
					self._h, 1, n, states[1:].ctypes.data, G[1:].ctypes.data, parents[1:].ctypes.data, pact[1:].ctypes.data,
					_ffi.stream_ptr()))
			self._cache = (states, G, parents, pact)
		return self._cache

	@property
	def states(self) -> np.ndarray:
		return self._export()[0]

	@property
	def G(self) -> np.ndarray:
		return self._export()[1]

	@property
	def parents(self) -> np.ndarray:
		return self._export()[2]

	@property
	def parent_actions(self) -> np.ndarray:
		return self._export()[3]

	@property
	def indices(self) -> dict:
		"""state bytes -> index, rebuilt on the host from the exported pool (the reference's dict, agents.py:201)."""
		states = self.states
		return {states[i].tobytes(): i for i in range(1, self._n + 1)}

	@property
	def open_queue(self) -> list:
		"""The open set as (cost, index) pairs in pop order."""
		if self._h is None or self._n == 0:
			return []
		n = int(_ffi.lib().rk_astar_open_size(self._h))
		costs, idx = np.zeros(n, np.float64), np.zeros(n, np.int64)
		got = _ffi.lib().rk_astar_export_open(self._h, costs.ctypes.data, idx.ctypes.data, n, _ffi.stream_ptr())
		return list(zip(costs[:got].tolist(), idx[:got].tolist()))

	def index_of(self, state: np.ndarray) -> int:
		state = np.ascontiguousarray(state, dtype=np.int8)
		return int(_ffi.lib().rk_astar_lookup(self._h, state.ctypes.data, _ffi.stream_ptr()))

	@no_grad
	def cost(self, states: np.ndarray, indeces: np.ndarray) -> np.ndarray:
		"""lambda * G + (-value) for given states (agents.py:369-383)."""
		H = -self.net(cube.as_oh(states), value=True, policy=False)
		H = H.cpu().squeeze().detach().numpy()
		return self.lambda_ * self.G[indeces] + H

	@classmethod
	def from_saved(cls, loc: str, use_best: bool, lambda_: float, expansions: int):
		raise NotImplementedError("model loading belongs to the reference's librubiks.model; pass a loaded net to AStar(...)")

	def __len__(self) -> int:
		return self._n

	def __str__(self) -> str:
		return f'AStar (lambda={self.lambda_}, N={self.expansions})'
