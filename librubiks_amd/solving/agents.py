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
	"""Protocol of the reference's agents (agents.py:14-64): one-step agents implement `_step`."""
	eps = np.finfo("float").eps
	_explored_states = 0

	def __init__(self):
		self.action_queue = deque()

	@no_grad
	def search(self, state: np.ndarray, time_limit: float = None, max_states: int = None) -> bool:
		time_limit, max_states = self.reset(time_limit, max_states)
		t0 = time.perf_counter()
		if cube.is_solved(state):
			return True
		solved = False
		while not solved and time.perf_counter() - t0 < time_limit and len(self) < max_states:
			action, state, solved = self._step(state)
			self.action_queue.append(action)
			self._explored_states = len(self.action_queue)
		return solved

	def _step(self, state: np.ndarray):
		"""-> (action index, new state, is solved)"""
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


def _net_signature(net):
	"""Changes whenever the net is swapped or its parameters / buffers are modified in place (optimizer steps,
	load_state_dict, BatchNorm statistics): module identity + training flag + the tensors' storage and version counters.
	Writes through `.data` (`p.data.add_(...)`, the idiom of hand-written update loops) do not bump a tensor's version counter, so
	the signature also carries a checksum of the first floating-point parameter -- the layer the fused form copies -- read at the
	start of a search (one small reduction and one device-to-host wait per search); an update that leaves that one tensor's
	sum unchanged bit for bit and bypasses the version counters of all others is the remaining blind spot."""
	sig = [id(net), bool(getattr(net, "training", False))]
	first = None
	for get in ("parameters", "buffers"):
		it = getattr(net, get, None)
		if callable(it):
			for t in it():
				sig.append((t.data_ptr(), t._version))
				if first is None and t.is_floating_point():
					first = t
	if first is not None:
		sig.append(float(first.detach().double().sum()))
	return tuple(sig)


#: How every search step is captured: errors of the capture are judged per THREAD.  In a process with a torch.distributed process group
#: the NCCL (RCCL) watchdog thread polls events while the main thread captures; under the default "global" mode such a call from another
#: thread invalidates the capture -- a sporadic failure of the first search of a rank (seen once in the round-5 test runs).
CAPTURE = {"capture_error_mode": "thread_local"}


def _capture_key(net, fs) -> tuple:
	"""What a captured search step holds of the net: the module, its mode, and the storage of every parameter and buffer (their
	VALUES are read at replay time, so in-place training between searches keeps a captured step valid; the fused copy `fs` is
	rebuilt -- a new object -- whenever the values change)."""
	ptrs = []
	for get in ("parameters", "buffers"):
		it = getattr(net, get, None)
		if callable(it):
			ptrs += [(t.data_ptr(), t.dtype) for t in it()]
	return (id(net), bool(getattr(net, "training", False)), tuple(ptrs), id(fs))


class DeepAgent(Agent):
	"""
	An agent with a value/policy net.  `fused_first_layer` (False, True, "epilogue", "folded"; librubiks_amd.oh_linear)
	makes the net's first Linear(480, H) read the engine's 20-byte states; that form COPIES weights (and, folded, BatchNorm
	statistics and the merged heads), while the reference trains its net in place and reassigns `agent.net` during
	training (train.py:134, :214).  So the copy is never trusted blindly: `_from_states` -- read at the start of every
	search -- rebuilds it whenever the net's signature (module identity, training flag, every parameter's and buffer's
	storage and in-place version counter) differs from the one the copy was taken at.
	"""
	def __init__(self, net, fused_first_layer=False):
		super().__init__()
		if fused_first_layer not in (False, None, True, "epilogue", "folded"):
			raise ValueError('fused_first_layer is False, True, "epilogue" or "folded"')
		self._fused_mode = fused_first_layer or False
		self._fused, self._fused_sig = None, None
		self._fs = None                    # `_from_states` as checked at the start of the running search (hot loops use this)
		self.net = net

	@property
	def net(self):
		return self._net

	@net.setter
	def net(self, net):
		self._net = net                    # the signature holds the module's identity: a swapped net is re-copied at the next search

	@property
	def _from_states(self):
		"""callable(states (n, 20) int8, policy, value) equal to net(as_oh(states), policy, value), or None when not fused."""
		if not self._fused_mode:
			return None
		self._net.eval()
		sig = _net_signature(self._net)
		if self._fused is None or sig != self._fused_sig:
			from librubiks_amd.oh_linear import fused_net
			self._fused, self._fused_sig = fused_net(self._net, self._fused_mode), sig
		return self._fused

	@classmethod
	def from_saved(cls, loc: str, use_best: bool, loader=None):
		"""agents.py:72-76."""
		return cls(_load_net(loc, use_best, loader))


def _host_softmax(logits_cpu: torch.Tensor) -> torch.Tensor:
	"""`logits.softmax(dim=1)` of a small CPU tensor on ONE intra-op thread (same kernel, same bits: rows are independent).  torch
	forks its whole thread pool even for twelve rows; on a host whose CPU share is a fraction of its thread count the pool's threads
	spin after every call and starve the thread that launches GPU work (measured: 3.3 ms per MCTS simulation instead of 0.29)."""
	threads = torch.get_num_threads()
	if threads != 1:
		torch.set_num_threads(1)
	try:
		return logits_cpu.softmax(dim=1)
	finally:
		if threads != 1:
			torch.set_num_threads(threads)


class RandomSearch(Agent):
	"""Random walk (agents.py:82-89)."""
	def _step(self, state):
		action = np.random.randint(cube.action_dim)
		state = cube.rotate(state, *cube.action_space[action])
		return action, state, cube.is_solved(state)

	def __str__(self):
		return "Random depth-first search"


class BFS(Agent):
	"""Breadth-first search over the 12-move graph (agents.py:92-129).  The reference pops one state at a time and moves it
	twelve times; here the children of up to `chunk` queued states come from ONE fan-out launch (with the goal test), and the
	reference's loop -- same order, same checks before every pop -- then runs over them, so `states`, the action queue and
	len(agent) are the reference's."""
	states = dict()
	chunk = 16_384                          # queued states expanded per launch (their 12 x chunk children are held on the host)

	def search(self, state: np.ndarray, time_limit: float = None, max_states: int = None) -> bool:
		time_limit, max_states = self.reset(time_limit, max_states)
		t0 = time.perf_counter()
		self.states = {}
		if cube.is_solved(state):
			return True
		self.states = {state.tobytes(): (None, None)}          # state -> (predecessor key, action)
		queue = deque([np.ascontiguousarray(state, dtype=np.int8)])
		while queue and time.perf_counter() - t0 < time_limit and len(self) < max_states:
			# the states that will be popped next, in order; a first-in-first-out queue only ever grows at the far end, so the
			# children of its head do not depend on what the loop below appends
			head = [queue[i] for i in range(min(len(queue), self.chunk, max(1, (max_states - len(self)) // 11 + 1)))]
			children, solved = cube.expand(np.array(head), return_solved=True)
			for j in range(len(head)):
				if not (time.perf_counter() - t0 < time_limit and len(self) < max_states):      # agents.py:105, before every pop
					return False
				pkey = queue.popleft().tobytes()
				for a in range(cube.action_dim):
					child = children[12 * j + a]
					ckey = child.tobytes()
					if ckey in self.states:
						continue
					if solved[12 * j + a]:
						self.action_queue.appendleft(a)
						while self.states[pkey][0] is not None:
							pkey, act = self.states[pkey]
							self.action_queue.appendleft(act)
						return True
					self.states[ckey] = (pkey, a)
					queue.append(child.copy())                      # (a view would keep the whole launch's children alive)
		return False

	def __str__(self):
		return "Breadth-first search"

	def __len__(self):
		return len(self.states)


class PolicySearch(DeepAgent):
	"""Follow (or sample from) the policy head (agents.py:132-151)."""
	def __init__(self, net, sample_policy=False):
		super().__init__(net)
		self.sample_policy = sample_policy

	def _step(self, state):
		logits = self.net(cube.as_oh(state), value=False)
		policy = _host_softmax(logits.float().cpu()).numpy().squeeze()
		action = int(np.random.choice(cube.action_dim, p=policy)) if self.sample_policy else int(policy.argmax())
		state = cube.rotate(state, *cube.action_space[action])
		return action, state, cube.is_solved(state)

	@classmethod
	def from_saved(cls, loc: str, use_best: bool, sample_policy=False, loader=None):
		return cls(_load_net(loc, use_best, loader), sample_policy)

	def __str__(self):
		return f"{'Sampled' if self.sample_policy else 'Greedy'} policy"


class ValueSearch(DeepAgent):
	"""Greedy one-step look-ahead on the value head (agents.py:154-169); one fan-out launch per step."""
	def _step(self, state):
		children, solved = cube.expand(np.asarray(state)[None], return_solved=True)
		if solved.any():
			action = int(np.flatnonzero(solved)[0])
			return action, children[action], True
		v = self.net(cube.as_oh(children), policy=False).float().squeeze().cpu().numpy()
		action = int(np.argmax(v))
		return action, children[action], False

	def __str__(self):
		return "Greedy value"


class EGVM(DeepAgent):
	"""Epsilon-greedy value maximisation (agents.py:649-726): `workers` walkers of `depth` moves, restart from the best."""
	def __init__(self, net, epsilon: float, workers: int, depth: int):
		super().__init__(net)
		self.epsilon, self.workers, self.depth = epsilon, workers, depth

	@no_grad
	def search(self, state: np.ndarray, time_limit: float = None, max_states: int = None) -> bool:
		time_limit, max_states = self.reset(time_limit, max_states)
		t0 = time.perf_counter()
		if cube.is_solved(state):
			return True
		while time.perf_counter() - t0 < time_limit and len(self) + self.workers * self.depth <= max_states:
			paths, states, states_oh, solved = self.expand(state)
			if solved != (-1, -1):
				self.action_queue += deque(int(a) for a in paths[solved[0], :solved[1]])
				return True
			best = int(self.net(states_oh, policy=False).float().cpu().squeeze().argmax())
			state = states[best]
			worker, d = divmod(best, self.depth)
			self.action_queue += deque(int(a) for a in paths[worker, :d + 1])
		return False

	def expand(self, state: np.ndarray):
		"""Walks all workers `depth` moves on the device; row w*depth + d = worker w after d+1 moves."""
		cur = torch.from_numpy(cube.repeat_state(np.asarray(state), self.workers)).to(gpu)
		paths = np.empty((self.workers, self.depth), dtype=int)
		visited = torch.empty((self.workers, self.depth, 20), dtype=torch.int8, device=gpu)
		flags = torch.empty(self.workers, dtype=torch.uint8, device=gpu)
		stats0 = torch.tensor([0, _ffi.INT64_MAX], dtype=torch.int64, device=gpu)
		stats = torch.empty_like(stats0)
		for d in range(self.depth):
			use_random = np.random.choice(2, self.workers, p=[1 - self.epsilon, self.epsilon]).astype(bool)
			actions = np.empty(self.workers, dtype=int)
			actions[use_random] = np.random.randint(0, cube.action_dim, use_random.sum())
			if (~use_random).any():
				p = self.net(cube.device.as_oh(cur[torch.from_numpy(~use_random).to(gpu)].contiguous()), value=False).float().cpu().numpy()
				actions[~use_random] = p.argmax(axis=1)
			paths[:, d] = actions
			# move + goal test in one launch (agents.py:696-703); the host reads the two statistics words, not the flags
			stats.copy_(stats0)
			cur, _ = cube.device.multi_rotate_solved(cur, torch.from_numpy(actions.astype(np.uint8)).to(gpu), flags=flags, stats=stats)
			hit = stats.cpu()
			if int(hit[0]):
				self._explored_states += (d + 1) * self.workers
				return paths, None, None, (int(hit[1]), d + 1)
			visited[:, d] = cur
		self._explored_states += self.workers * self.depth
		flat = visited.reshape(self.workers * self.depth, 20)
		return paths, flat.cpu().numpy(), cube.device.as_oh(flat), (-1, -1)

	@classmethod
	def from_saved(cls, loc: str, use_best: bool, epsilon: float, workers: int, depth: int, loader=None):
		return cls(_load_net(loc, use_best, loader), epsilon=epsilon, workers=workers, depth=depth)

	def __str__(self):
		return f"EGVM (e={self.epsilon}, w={self.workers}, d={self.depth})"


_OH_CODES = {torch.float32: _ffi.OH_F32, torch.float16: _ffi.OH_F16, torch.bfloat16: _ffi.OH_BF16}


def _load_net(loc: str, use_best: bool, loader=None):
	"""The value/policy net of a saved model folder (reference: DeepAgent.from_saved, agents.py:72-75)."""
	if loader is not None:
		return loader(loc, use_best)
	try:
		from librubiks.model import Model          # the reference package, when installed beside this one
	except ImportError as e:
		raise ImportError("from_saved needs the reference's librubiks.model.Model (or pass loader=callable(loc, use_best) -> net); "
		                  "librubiks_amd replaces the cube and search engines, not the network") from e
	return Model.load(loc, load_best=use_best).to(gpu)


def _oh_dtype(net) -> torch.dtype:
	"""
	The dtype the net wants its one-hot input in: that of its first floating-point parameter (a bf16/fp16 net gets a
	bf16/fp16 one-hot straight from the kernel -- 0/1 are exact, and no cast kernel or float32 copy is needed);
	float32 for anything that is not a torch module.
	"""
	params = getattr(net, "parameters", None)
	if callable(params):
		for prm in params():
			if prm.dtype in _OH_CODES:
				return prm.dtype
	return torch.float32


def _has_f32_weights(net) -> bool:
	"""True for a torch module whose first floating-point parameter is float32 (a forward pass on thousands of rows then
	costs milliseconds); False for low-precision nets and for parameter-free heuristics."""
	params = getattr(net, "parameters", None)
	if callable(params):
		for prm in params():
			if prm.dtype in _OH_CODES:
				return prm.dtype == torch.float32
	return False


def _value_f32(out) -> torch.Tensor:
	"""The net's value head as a contiguous float32 vector on the GPU."""
	if isinstance(out, (list, tuple)):
		out = out[-1]
	return out.detach().to(device=gpu, dtype=torch.float32).reshape(-1).contiguous()


def _values_for_engine(h, out) -> torch.Tensor:
	"""
	The net's value head as the A* engine takes it: a bfloat16 net's values go in as they are (the engine widens them,
	exactly; rk_astar_set_values_dtype), anything else as a contiguous float32 vector.  Tells the engine which it is.
	"""
	v = out[-1] if isinstance(out, (list, tuple)) else out
	if isinstance(v, torch.Tensor) and v.is_cuda and v.dtype == torch.bfloat16 and v.is_contiguous():
		v, code = v.detach().reshape(-1), _ffi.OH_BF16
	else:
		v, code = _value_f32(v), _ffi.OH_F32
	_ffi.check(_ffi.lib().rk_astar_set_values_dtype(h, code))
	return v


#: Rows per net forward.  A forward on B rows streams B x 4096 activations per layer through every elementwise kernel
#: of the torch module (Linear, ELU, BatchNorm ...); while a slice's activations fit the 256 MiB Infinity Cache those
#: kernels run out of it, beyond that every one of them goes to HBM: 64 searches x 12 000 rows in ONE forward ran four
#: times slower PER ROW than 12 000-row forwards (profiles/r02_astar_batch.json: batch 0.22x of sequential at N = 1000).
#: 16 384 rows x 4 096 x 2 B = 134 MB (bf16).  The reference slices its own large forwards the same way (train.py:301-311).
NET_SLICE_ROWS = 16_384


def _value_of(out):
	return out[-1] if isinstance(out, (list, tuple)) else out


def _sliced_value_forward(forward, rows: torch.Tensor, max_rows: int = None):
	"""The value head of `forward` on `rows`, evaluated in slices of at most `max_rows` rows (see NET_SLICE_ROWS)."""
	max_rows = max_rows or NET_SLICE_ROWS
	n = len(rows)
	if n <= max_rows:
		return _value_of(forward(rows, policy=False, value=True))
	return torch.cat([_value_of(forward(rows[i:i + max_rows], policy=False, value=True)).reshape(-1) for i in range(0, n, max_rows)])


class CapacityExhausted(RuntimeWarning):
	"""A search that was limited only by time stopped because its node pool was full (the reference grows its arrays)."""


class AStar(DeepAgent):
	"""
	Batch weighted A* (agents.py:171-413): expands the `expansions` cheapest open nodes per iteration,
	cost = lambda_ * G + (-value).  Same results as the reference (index numbering, G, parents, action_queue)
	whenever the net returns the same values.

	Two ways to drive an iteration, chosen by the batch size K = 12 * expansions:
	  * K < 2048 (latency regime, the reference's N = 27 ... 170): `rk_astar_step_expand` -> net forward on the fixed
	    (K, 480) one-hot batch -> `rk_astar_step_commit`: six small launches around the net, no host synchronisation; the
	    host polls the engine's status every few iterations (never past the state budget; steps after a win are no-ops
	    on the device).  `use_hipgraph=True` captures the iteration once (the net must be capturable) and replays it.
	  * `exact_batch` (default: K >= 2048 with a float32 net, i.e. when a forward pass costs milliseconds):
	    `rk_astar_expand` synchronises once per iteration so that the net runs on exactly the new states instead of the
	    padded batch (measured with fc_small at N = 700 ... 1000: 15 % faster in float32, but 20 % slower than running
	    ahead of the GPU with the three times faster bf16 net).

	`capacity` is the size the node pool starts with.  A search whose state budget is larger (or that is limited only by
	time) GROWS the pool in place when it fills up, as the reference's increase_stack_size does (agents.py:396-402):
	`rk_astar_grow` doubles it -- device-to-device copies and one rehash kernel, the open queue stays as it is -- and the
	search continues where it stood, so it ends with the arrays of a search that had started in the large pool.  At
	`max_capacity` a search that still fills the pool warns with `CapacityExhausted` and sets `self.capacity_exhausted`.
	`self.grown` counts the growths of the last search.
	"""
	default_capacity = 4_000_000
	max_capacity = 64_000_000

	def __init__(self, net, lambda_: float, expansions: int, capacity: int = None, poll: int = 4, use_hipgraph: bool = False,
	             fused_first_layer=False, exact_batch: bool = None):
		# fused_first_layer: the engine hands the net the new nodes' 20-byte states and the net's first Linear(480, H)
		# reads them directly (librubiks_amd.oh_linear) -- no one-hot batch exists at all (DeepAgent keeps the copy fresh)
		super().__init__(net, fused_first_layer)
		self.exact_batch = exact_batch
		self.lambda_ = lambda_
		self.expansions = int(expansions)
		self.capacity = capacity
		self.poll = max(1, int(poll))
		# use_hipgraph: True / False, or "auto" = replay the iteration as a hipGraph only where that wins.  Measured eager vs
		# graph, us per iteration (profiles/r03_astar_small.json): with a real net the graph wins while the HOST is the
		# bottleneck (fc_small bf16: N = 10 207 -> 124, N = 27 196 -> 140), ties at N = 100 and LOSES above (N = 700: 471 -> 589;
		# float32 1580 -> 1878), although the kernels and their order are the same and under rocprofv3 (which serialises both)
		# both modes take the same time kernel by kernel (profiles/r03_astar_graph_gaps.json).
		self.use_hipgraph = (self.expansions <= 50) if use_hipgraph == "auto" else bool(use_hipgraph)
		self._h = None
		self._h_cap = 0
		self._h_expansions = 0
		self._n = 0
		self._root = None
		self._cache = None
		self.iterations = 0
		self.capacity_exhausted = False
		self.grown = 0                # times the pool grew in place during the last search
		self.profile_events = None    # a list: every run-ahead iteration appends four HIP events (before expand, before the net, after it, after commit)
		self._graph_cache = None      # (key, hipGraph of one iteration, its batch buffer): kept from search to search, see search()
		self.captures = 0             # hipGraph captures so far (a search on an unchanged engine and net replays the previous search's graph)
		self.record_pops = False      # debugging aid: keep the popped indices of every iteration in self.pops
		self.pops = []

	# -- engine lifetime ------------------------------------------------------------------------------------
	def _engine(self, capacity: int):
		if self._h is not None and self._h_cap >= capacity and self._h_expansions == self.expansions:
			return self._h
		self._free()
		h = C.c_void_p()
		_ffi.check(_ffi.lib().rk_astar_create(C.byref(h), capacity, self.expansions))
		self._h, self._h_cap, self._h_expansions = h, capacity, self.expansions      # (an engine is built for one batch size)
		return h

	def _free(self):
		self._graph_cache = None              # it holds the engine's addresses
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
		self.capacity_exhausted = False
		return time_limit, max_states

	def _iteration(self, h, oh, code):
		lib = _ffi.lib()
		ev = self.profile_events           # measurement aid (bench.py): HIP events between the three parts of an iteration
		if ev is not None:
			marks = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
			marks[0].record()
		_ffi.check(lib.rk_astar_step_expand(h, oh.data_ptr(), code, _ffi.stream_ptr()))
		if ev is not None:
			marks[1].record()
		values = _values_for_engine(h, _sliced_value_forward(self._fs or self.net, oh))
		self._keep = values                # the commit kernels read it after this call returns
		if ev is not None:
			marks[2].record()
		_ffi.check(lib.rk_astar_step_commit(h, values.data_ptr(), _ffi.stream_ptr()))
		if ev is not None:
			marks[3].record()
			ev.append(marks)

	@no_grad
	def search(self, state: np.ndarray, time_limit: float = None, max_states: int = None) -> bool:
		_ffi.require_gpu()
		t0 = time.perf_counter()
		time_limit, max_states = self.reset(time_limit, max_states)
		state = np.ascontiguousarray(state, dtype=np.int8)
		if cube.is_solved(state):
			return True
		K = 12 * self.expansions
		cap = max(int(min(max_states, self.capacity or self.default_capacity)), K + 2)
		lib = _ffi.lib()
		self._fs = self._from_states           # re-copied here if the net changed since the last search
		code = _ffi.OH_STATES if self._fs is not None else _OH_CODES[_oh_dtype(self.net)]
		cached = self._graph_cache
		if cached is not None and self.use_hipgraph and cached[2][1] == code and len(cached[2][0]) == K:
			oh = cached[2][0]                  # the buffer the kept graph was captured on (rows are rewritten before they are read)
		elif self._fs is not None:
			oh = torch.from_numpy(cube.repeat_state(cube.get_solved(), K)).to(gpu)      # (K, 20) int8: valid codes everywhere
		else:
			oh = torch.zeros((K, 480), dtype=_oh_dtype(self.net), device=gpu)
		status = (C.c_longlong * 8)()
		h = self._engine(cap)
		cap = self._h_cap
		_ffi.check(lib.rk_astar_reset(h, state.ctypes.data, float(self.lambda_), _ffi.stream_ptr()))
		_ffi.check(lib.rk_astar_set_budget(h, int(min(max_states, cap)), _ffi.stream_ptr()))
		self._root, self._n, self._cache = state.copy(), 1, None
		self.iterations, self.pops, self.grown = 0, [], 0
		exact = self.exact_batch if self.exact_batch is not None else (K >= 2048 and _has_f32_weights(self.net))
		exact = exact and not self.use_hipgraph and not self.record_pops
		info = (C.c_longlong * 5)()
		while True:                                                  # one round per pool size: the pool grows in place below
			graph = None
			if self.use_hipgraph and not self.record_pops:
				# The captured iteration holds addresses (engine pools, batch buffer, the net's tensors) and two scalars passed by
				# value (lambda, the values' dtype) -- nothing of the search itself, which lives in device memory that rk_astar_reset
				# rewrites.  So the graph is KEPT from search to search and captured again only when one of those changes (a grown
				# or new engine, another net or fused copy, another lambda): capturing cost every search about 7 ms
				# (profiles/NOTES.md section 7), as much as a hundred iterations.
				key = (h.value, cap, float(self.lambda_), code, oh.data_ptr(), _capture_key(self.net, self._fs))
				if self._graph_cache is not None and self._graph_cache[0] == key:
					graph = self._graph_cache[1]
				else:
					self._graph_cache = None
					side = torch.cuda.Stream()
					side.wait_stream(torch.cuda.current_stream())
					with torch.cuda.stream(side):
						self._iteration(h, oh, code)               # a real iteration; also warms the allocator
					torch.cuda.current_stream().wait_stream(side)
					graph = torch.cuda.CUDAGraph()                 # (captured again after a growth: it holds the pool's addresses)
					with torch.cuda.graph(graph, **CAPTURE):
						self._iteration(h, oh, code)
					self._graph_cache = (key, graph, (oh, code), (self.net, self._fs))     # the net stays alive with the graph that holds its addresses
					self.captures += 1
			budget = int(min(max_states, cap))
			done = won = err = solved_idx = 0
			while exact:
				# one synchronisation per iteration, the net sees exactly the new states (agents.py:315, :369-383)
				_ffi.check(lib.rk_astar_expand(h, self.expansions, info, _ffi.stream_ptr()))
				n_pop, n_new, won, solved_idx, self._n = (int(x) for x in info)
				if n_pop == 0:
					done = 1
					break
				self.iterations += 1
				if won:
					break
				values = None
				if n_new:
					_ffi.check(lib.rk_astar_new_states_oh(h, oh.data_ptr(), code, _ffi.stream_ptr()))
					values = _values_for_engine(h, _sliced_value_forward(self._fs or self.net, oh[:n_new]))
				_ffi.check(lib.rk_astar_commit(h, values.data_ptr() if values is not None else None, _ffi.stream_ptr()))
				if time.perf_counter() - t0 >= time_limit:
					break
			while not exact:
				# never run past the state budget: a search grows by at most K states per iteration
				poll = 1 if self.record_pops else max(1, min(self.poll, (budget - self._n) // K))
				for _ in range(poll):
					if self.record_pops:
						head = np.zeros(self.expansions, np.int64)
						got = lib.rk_astar_next_pops(h, head.ctypes.data, self.expansions, _ffi.stream_ptr())
						if got > 0:
							self.pops.append(head[:got].copy())
					if graph is not None:
						graph.replay()
					else:
						self._iteration(h, oh, code)
				_ffi.check(lib.rk_astar_status(h, status, _ffi.stream_ptr()))
				done, won, self._n, self.iterations, solved_idx, err = (int(status[i]) for i in (0, 1, 2, 3, 5, 6))
				if err:
					raise _ffi.RubiksHipError(f"A* engine error code {err}")
				if won or done or time.perf_counter() - t0 >= time_limit:
					break
			self._cache = None
			if won:
				path = (C.c_longlong * 4096)()
				n = lib.rk_astar_path(h, solved_idx, path, 4096, _ffi.stream_ptr())
				if n < 0:
					_ffi.check(int(n))
				self.action_queue = deque(int(a) for a in path[:n])
				return True
			# out of budget, out of time, or nothing left to expand
			pool_full = done and self._n + K > cap and lib.rk_astar_open_size(h) > 0
			if not (pool_full and max_states > cap and time.perf_counter() - t0 < time_limit):
				return False
			# the pool, not the caller's budget, ended the search: grow it like the reference's increase_stack_size
			# (agents.py:396-402) -- in place, the search goes on where it stood -- or say so
			if cap >= self.max_capacity:
				self.capacity_exhausted = True
				import warnings
				warnings.warn(f"{self}: node pool of {cap} states is full with time left; raise max_capacity", CapacityExhausted)
				return False
			cap = min(2 * cap, self.max_capacity)
			_ffi.check(lib.rk_astar_grow(h, cap, _ffi.stream_ptr()))
			self._h_cap = cap
			_ffi.check(lib.rk_astar_set_budget(h, int(min(max_states, cap)), _ffi.stream_ptr()))
			self.grown += 1

	# -- inspection (what the reference's tests look at: tests/test_agents.py:96-145) --------------------------
	def _export(self):
		if self._cache is None:
			n = self._n
			rows = max(n + 1, 1000)            # the reference's arrays start with 1000 rows (agents.py:385-394); its tests write G of a reset agent
			states = np.zeros((rows, 20), np.int8)
			G = np.zeros(rows, np.float64)
			parents = np.zeros(rows, np.int64)
			pact = np.zeros(rows, np.int64)
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
	def from_saved(cls, loc: str, use_best: bool, lambda_: float, expansions: int, loader=None):
		"""agents.py:405-407.  The net itself is the reference's `librubiks.model.Model` (out of this package's scope):
		it is loaded with `loader(loc, use_best)` if given, else with the reference's `Model.load` when that package is
		importable next to this one (the drop-in situation)."""
		return cls(_load_net(loc, use_best, loader), lambda_=lambda_, expansions=expansions)

	def __len__(self) -> int:
		return self._n

	def __str__(self) -> str:
		return f'AStar (lambda={self.lambda_}, N={self.expansions})'


def _policy_value_f32(out):
	"""(softmaxed policy (B,12), value (B,)) as contiguous float32 GPU tensors from the net's [logits, value]."""
	p, v = out
	p = p.detach().to(device=gpu, dtype=torch.float32).softmax(dim=1).contiguous()
	v = v.detach().to(device=gpu, dtype=torch.float32).reshape(-1).contiguous()
	return p, v


PRIORS = ("kernel", "torch", "reference")


class MCTSBatch(DeepAgent):
	"""
	T independent Monte Carlo tree searches advanced in lock-step on the GPU (engine rk_mcts_*).  Each tree follows
	the reference's MCTS exactly (agents.py:415-645): same node numbering, neighbors, P, V, N, W, L and action queue
	as running the reference on that start state alone, whenever the net returns the same numbers.

	One simulation of all trees = expand kernel, one-hot kernel, net forward on the fixed (12 T, 480) batch,
	backup+select kernel.  Nothing in a step synchronises, so `use_graph=True` captures the step in a hipGraph
	(through torch.cuda.CUDAGraph) and replays it; the host only polls every `poll` simulations.

	priors -- where softmax(logits) (agents.py:472, :551-552) is computed:
	  "kernel"     inside the backup kernel from the net's raw float32 / bfloat16 logits: exp(x - max) / sum in float32, the sum
	               taken in the order torch.softmax's own kernel adds on this device (rk_mcts.hip, child_policy) -- no softmax,
	               conversion or copy kernels in the step; capturable.  The default of this class (the throughput form).
	  "torch"      torch.softmax on the device, float32; capturable.
	  "reference"  the reference's own two lines: the root's priors by `p.softmax(dim=1)` on the device (agents.py:472), every
	               other node's by `p.cpu().softmax(dim=1)` (agents.py:551-552: on the HOST), bit for bit whatever torch's CPU
	               kernel does.  One device-to-host wait per simulation, so not capturable.  The default of `MCTS`.
	`search_graph`: after the search, every solved tree gets the reference's graph completion and breadth-first shortening
	(agents.py:597-633) on the device (rk_mcts_search_graph); `action_queue_of` then returns the shortened queue.
	`max_capacity`: pools that fill up while a tree's own budget is larger grow in place (rk_mcts_grow; agents.py:450-460),
	doubling up to this many states per tree.
	"""

	def __init__(self, net, c: float, n_trees: int, capacity: int = 50_000, max_path: int = None, nu: float = 100.0,
	             fused_first_layer=False, torch_softmax: bool = False, priors: str = None, search_graph: bool = False,
	             max_capacity: int = None, overlap_halves: bool = False):
		# fused_first_layer: the net's first Linear(480, H) reads the children's 20-byte states (librubiks_amd.oh_linear)
		super().__init__(net, fused_first_layer)
		priors = priors or ("torch" if torch_softmax else "kernel")
		if priors not in PRIORS:
			raise ValueError(f"priors is one of {PRIORS}")
		self.priors = priors
		self.search_graph = bool(search_graph)
		# overlap_halves (hipGraph mode, fused first layer, priors in the kernel): the captured step advances the batch as two halves
		# on two streams, skewed by half a step, so that one half's latency-bound backup + descent runs under the other half's net
		# forward (see _capture_halves).  Trees are independent: every tree's arrays are what they would be without it.
		self.overlap_halves = bool(overlap_halves)
		self.c, self.nu, self.n_trees = float(c), float(nu), int(n_trees)
		self.capacity = int(capacity)
		self.max_capacity = int(max_capacity) if max_capacity else None
		self._auto_path = max_path is None
		self.max_path = int(max_path or max(4096, 2 * self.capacity))
		self._h = None
		self._shape = None
		self.status = None
		self.simulations = 0
		self.grown = 0
		self.profile_events = None        # a list: every eager simulation step appends a HIP event pair around its backup + select launch
		self.on_poll = None               # callable(status): called with every status the search reads (solving/evaluation.py times games with it)
		self._graph_cache = None          # (key, hipGraph of one step, its batch buffer): kept from search to search, see _capture
		self.captures = 0                 # hipGraph captures so far

	@property
	def torch_softmax(self) -> bool:
		return self.priors == "torch"

	def _engine(self):
		shape = (self.n_trees, self.capacity, self.max_path)
		if self._h is None or self._shape != shape:
			self._free()
			h = C.c_void_p()
			_ffi.check(_ffi.lib().rk_mcts_create(C.byref(h), *shape))
			self._h, self._shape = h, shape
		return self._h

	def _free(self):
		self._graph_cache = None              # it holds the engine's addresses
		if getattr(self, "_h", None) is not None:
			_ffi.lib().rk_mcts_destroy(self._h)
			self._h = None

	def __del__(self):
		try:
			self._free()
		except Exception:
			pass

	_ERRORS = {1: "path longer than max_path", 2: "broken neighbour link", 3: "backup without a pending expansion (rk_mcts_expand missing from the step)"}

	def _poll(self):
		st = np.zeros((self.n_trees, 6), np.int64)
		_ffi.check(_ffi.lib().rk_mcts_status(self._h, st.ctypes.data, _ffi.stream_ptr()))
		if (st[:, 5] != 0).any():
			codes = sorted(set(int(x) for x in st[:, 5] if x))
			raise _ffi.RubiksHipError(f"MCTS engine error codes {codes}: " + "; ".join(self._ERRORS.get(k, "?") for k in codes)
			                          + f" (max_path={self.max_path})")
		self.status = st
		if self.on_poll is not None:
			self.on_poll(st)
		return st

	def _kept_buffer(self, code: int):
		"""The batch buffer of the kept graph, if it fits this search (the step rewrites its rows before the net reads them)."""
		kept = self._graph_cache
		if kept is None:
			return None
		oh = kept[2]
		want = torch.int8 if code == _ffi.OH_STATES else {v: k for k, v in _OH_CODES.items()}[code]
		return oh if len(oh) == 12 * self.n_trees and oh.dtype == want else None

	def _step(self, oh, h, expand: bool = True):
		lib = _ffi.lib()
		if expand:
			_ffi.check(lib.rk_mcts_expand(h, _ffi.stream_ptr()))
		if self._fs is not None:
			# the first layer reads the children where the engine keeps them; no one-hot, no copy
			x = self._fs.first.from_pointer(lib.rk_mcts_children(h), 12 * self.n_trees)
			p, v = self._fs.tail(x)
		else:
			_ffi.check(lib.rk_mcts_children_oh(h, oh.data_ptr(), _OH_CODES[oh.dtype], _ffi.stream_ptr()))
			p, v = self.net(oh)
		ev = self.profile_events           # measurement aid (bench.py): a HIP event pair around the backup + select launch
		if self.priors == "kernel" and isinstance(p, torch.Tensor) and p.is_cuda and p.dtype == v.dtype and p.dtype in (torch.float32, torch.bfloat16) \
		   and p.dim() == 2 and p.stride(1) == 1 and p.stride(0) >= 12 and v.numel() == len(p) and v.reshape(len(p), -1).stride(0) >= 1:
			# raw logits and values in the net's dtype (rows may be views into one tensor of merged heads): the softmax
			# (agents.py:551) runs inside the backup kernel
			self._keep = (p, v)        # the kernels read these after this call returns
			if ev is not None:
				pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
				pair[0].record()
			_ffi.check(lib.rk_mcts_backup_select_logits(h, p.data_ptr(), p.stride(0), v.data_ptr(), v.reshape(len(p), -1).stride(0),
			                                            _OH_CODES[p.dtype], _ffi.stream_ptr()))
			if ev is not None:
				pair[1].record()
				ev.append(pair)
			return
		if self.priors == "reference":
			# agents.py:551-552 to the letter: `p.cpu().softmax(dim=1)` -- the HOST's softmax of the logits, the same CPU kernel on
			# the same numbers; only the plumbing differs: the logits travel through page-locked buffers, and the 144 numbers are
			# softmaxed by ONE thread.  torch forks its whole intra-op pool even for twelve rows, and a pool of 128 threads spinning
			# after every simulation on a host whose CPU share is a fraction of that starves the launching thread: 3.3 ms per
			# simulation instead of 0.25 (profiles/r04_mcts_priors_cost.json).  Rows are independent: the bits do not depend on it.
			v = v.detach().to(device=gpu, dtype=torch.float32).reshape(-1).contiguous()
			buf = getattr(self, "_ref_buf", None)
			if buf is None or buf[0].shape != p.shape or buf[0].dtype != p.dtype:
				buf = self._ref_buf = (torch.empty(p.shape, dtype=p.dtype).pin_memory(), torch.empty(p.shape, dtype=torch.float32).pin_memory(),
				                       torch.empty(p.shape, dtype=torch.float32, device=gpu))
			h_in, h_out, d_out = buf
			h_in.copy_(p.detach(), non_blocking=True)
			torch.cuda.current_stream().synchronize()                    # the one device-to-host wait of this mode
			h_out.copy_(_host_softmax(h_in))
			d_out.copy_(h_out, non_blocking=True)
			p = d_out
		else:
			p, v = _policy_value_f32((p, v))
		self._keep = (p, v)
		_ffi.check(lib.rk_mcts_backup_select(h, p.data_ptr(), v.data_ptr(), _ffi.stream_ptr()))

	@no_grad
	def search(self, states: np.ndarray, time_limit: float = None, max_states=None, max_sims: int = None,
	           use_graph: bool = False, poll: int = 16) -> np.ndarray:
		"""Runs all trees until each is solved or out of budget; returns the bool vector `solved` (T,)."""
		t0 = time.perf_counter()
		assert time_limit or max_states is not None or max_sims
		time_limit = time_limit or 1e10
		self._begin(states, max_states, max_sims, use_graph)
		while time.perf_counter() - t0 < time_limit and (max_sims is None or self.simulations < max_sims):
			self._advance(poll if max_sims is None else min(poll, max_sims - self.simulations))
			if self._poll()[:, 0].all() and not self._grow():
				break
		return self._finish()

	# The three stages of a search.  Everything is enqueued on the CURRENT stream, so several engines can be advanced in turn on
	# streams of their own (tried in round 3 with 2 and 4 engines sharing the 256 trees, so that one engine's descent would
	# overlap the others' net forwards: 0.166 and 0.242 ms per 256-tree step against 0.159 with one engine,
	# profiles/r03_mcts_variants.json -- the small GEMMs of the halves do not overlap each other; dropped).
	@no_grad
	def _begin(self, states: np.ndarray, max_states, max_sims, use_graph: bool):
		_ffi.require_gpu()
		self.net.eval()
		states = np.ascontiguousarray(states, dtype=np.int8).reshape(self.n_trees, 20)
		if max_states is None:
			max_states = self.capacity
		self._budget = np.broadcast_to(np.asarray(max_states, dtype=np.int64), (self.n_trees,)).copy()      # what the caller asked for
		ms = np.minimum(self._budget, self.capacity)
		h, lib = self._engine(), _ffi.lib()
		_ffi.check(lib.rk_mcts_reset(h, states.ctypes.data, ms.ctypes.data, self.c, self.nu, _ffi.stream_ptr()))
		self._fs = self._from_states           # re-copied here if the net changed since the last search
		if self._fs is not None:
			root_oh = torch.empty((self.n_trees, 20), dtype=torch.int8, device=gpu)
			_ffi.check(lib.rk_mcts_roots_oh(h, root_oh.data_ptr(), _ffi.OH_STATES, _ffi.stream_ptr()))
			p, v = _policy_value_f32(self._fs(root_oh))
			oh = self._kept_buffer(_ffi.OH_STATES)
			if oh is None:
				oh = torch.from_numpy(cube.repeat_state(cube.get_solved(), 12 * self.n_trees)).to(gpu)
		else:
			oh_dtype = _oh_dtype(self.net)
			root_oh = torch.empty((self.n_trees, 480), dtype=oh_dtype, device=gpu)
			_ffi.check(lib.rk_mcts_roots_oh(h, root_oh.data_ptr(), _OH_CODES[oh_dtype], _ffi.stream_ptr()))
			p, v = _policy_value_f32(self.net(root_oh))                  # agents.py:470-473: the root's softmax runs on the device
			oh = self._kept_buffer(_OH_CODES[oh_dtype])
			if oh is None:
				oh = torch.empty((12 * self.n_trees, 480), dtype=oh_dtype, device=gpu)
		_ffi.check(lib.rk_mcts_set_root_pv(h, p.data_ptr(), v.data_ptr(), _ffi.stream_ptr()))
		# every backup + select launch also expands the leaf it found, while another simulation is to follow
		_ffi.check(lib.rk_mcts_set_expand_ahead(h, int(max_sims) if max_sims is not None else -1))
		self.simulations, self.grown = 0, 0
		self._max_sims, self._oh, self._graph = max_sims, oh, None
		self._use_graph = bool(use_graph) and self.priors != "reference"       # the host softmax waits for the device: nothing to capture
		if self._use_graph and (max_sims is None or max_sims > 2):
			self._capture(2)

	def _capture(self, warm: int):
		"""`warm` real simulations, then the step as a hipGraph.  After an eager step with expand-ahead on, every tree's next
		expansion is always done by the backup launch before it, so the captured step leaves rk_mcts_expand out (one launch less
		per replay); were that ever untrue, the backup kernel stops the tree with error 3 instead of using stale children."""
		h, oh = self._h, self._oh
		if self.overlap_halves and self._fs is not None and self.priors == "kernel" and self.n_trees >= 2:
			return self._capture_halves(warm)
		# The captured step holds addresses (pools, batch buffer, the net's tensors) and scalars passed by value (c, nu, sizes, the
		# simulation limit of the expand-ahead) --
		# nothing of the trees, which live in device memory that rk_mcts_reset rewrites.  So the graph is KEPT from search to
		# search and captured again only when one of those changes (grown or new engine, another net or fused copy).
		key = (h.value, self._shape, self.c, self.nu, self.priors, self._max_sims, oh.data_ptr(), oh.dtype, _capture_key(self.net, self._fs))
		hit = self._graph_cache is not None and self._graph_cache[0] == key
		if hit:
			for _ in range(warm):                                      # real simulations: after them every tree is expanded ahead
				self._step(oh, h)
				self.simulations += 1
			self._graph = self._graph_cache[1]
			return
		self._graph_cache = None
		side = torch.cuda.Stream()
		side.wait_stream(torch.cuda.current_stream())
		with torch.cuda.stream(side):
			for _ in range(warm):                                      # real simulations; they also warm the allocator
				self._step(oh, h)
				self.simulations += 1
		torch.cuda.current_stream().wait_stream(side)
		self._graph = torch.cuda.CUDAGraph()
		with torch.cuda.graph(self._graph, **CAPTURE):
			self._step(oh, h, expand=False)
		self._graph_cache = (key, self._graph, oh, (self.net, self._fs))      # the net stays alive with the graph that holds its addresses
		self.captures += 1

	def _net_half(self, first: int, count: int):
		"""Raw logits and values of the trees first ... first + count - 1: the fused first layer reads their children where the engine keeps them."""
		lib = _ffi.lib()
		x = self._fs.first.from_pointer(lib.rk_mcts_children(self._h) + first * 12 * 20, 12 * count)
		return self._fs.tail(x)

	def _backup_half(self, first: int, count: int, p, v):
		_ffi.check(_ffi.lib().rk_mcts_backup_select_logits_range(self._h, first, count, p.data_ptr(), p.stride(0), v.data_ptr(), v.reshape(len(p), -1).stride(0),
		                                                          _OH_CODES[p.dtype], _ffi.stream_ptr()))

	def _capture_halves(self, warm: int):
		"""
		The step as a hipGraph that advances the batch in TWO HALVES, A = trees [0, T/2) and B = the rest, skewed by half a step
		(VERDICT r4 #3).  A step of one half is net forward -> backup + select (+ expand ahead); the second part is one wave per
		tree chasing pointers for ~45 us with < 2 % of the chip's wave slots occupied, the first is throughput-bound GEMMs.
		One replay =
		    phase 1:   backup+select A (simulation k)      ||   net B (simulation k)
		    phase 2:   net A (simulation k + 1)            ||   backup+select B (simulation k)
		on two streams forked and joined inside the capture, so each half's descent runs under the other half's GEMMs.  Before the
		first replay A's net outputs of simulation 1 are computed eagerly (`_prime_halves`); A's net of the simulation after the
		last one is computed in vain.  Every tree still sees select, expand, net, backup in the reference's order.
		"""
		h = self._h
		T = self.n_trees
		na = T // 2
		key = ("halves", h.value, self._shape, self.c, self.nu, self._max_sims, _capture_key(self.net, self._fs))
		hit = self._graph_cache is not None and self._graph_cache[0] == key
		side = torch.cuda.Stream() if not hit else None
		if hit:
			for _ in range(warm):
				self._step(self._oh, h)
				self.simulations += 1
			self._graph, self._half_bufs = self._graph_cache[1], self._graph_cache[4]
			self._prime_halves()
			return
		self._graph_cache = None
		main = torch.cuda.Stream()
		main.wait_stream(torch.cuda.current_stream())
		with torch.cuda.stream(main):
			for _ in range(warm):                                      # real simulations; they also warm the allocator
				self._step(self._oh, h)
				self.simulations += 1
			pa, va = self._net_half(0, na)                             # shapes / dtypes of a half's outputs (and a warm-up of the half-size GEMMs)
			pb, vb = self._net_half(na, T - na)
			del pb, vb
		torch.cuda.current_stream().wait_stream(main)
		# A's outputs cross from one replay to the next (and from the eager priming into the first replay): they live in buffers
		# of their own, copied into at the end of phase 2 (two copies of a few KB)
		self._half_bufs = (torch.empty_like(pa), torch.empty_like(va))
		del pa, va
		XA, VA = self._half_bufs
		self._graph = torch.cuda.CUDAGraph()
		with torch.cuda.graph(self._graph, **CAPTURE):
			cur = torch.cuda.current_stream()
			side.wait_stream(cur)                                      # fork
			self._backup_half(0, na, XA, VA)                           # phase 1, this stream: backup + select A
			with torch.cuda.stream(side):
				pb, vb = self._net_half(na, T - na)                    # phase 1, side stream: net B
			cur.wait_stream(side)                                      # both streams meet between the phases
			side.wait_stream(cur)
			pa, va = self._net_half(0, na)                             # phase 2, this stream: net A of the NEXT simulation
			XA.copy_(pa)
			VA.copy_(va)
			with torch.cuda.stream(side):
				self._backup_half(na, T - na, pb, vb)                  # phase 2, side stream: backup + select B
			cur.wait_stream(side)                                      # join
		self._graph_cache = (key, self._graph, self._oh, (self.net, self._fs), self._half_bufs)
		self.captures += 1
		self._prime_halves()

	def _prime_halves(self):
		"""A's net outputs of the coming simulation into the buffers the graph's first phase reads (eager, once per run of replays)."""
		pa, va = self._net_half(0, self.n_trees // 2)
		self._half_bufs[0].copy_(pa)
		self._half_bufs[1].copy_(va)

	@no_grad
	def _advance(self, n: int):
		for _ in range(n):
			if self._graph is not None:
				self._graph.replay()
			else:
				self._step(self._oh, self._h)
		self.simulations += n

	def _grow(self) -> bool:
		"""All trees are done.  If some stopped only because the POOL was their budget (the caller's is larger), grow every
		pool in place (agents.py:450-460, :503-504) and let those trees go on.  True if the search continues."""
		st = self.status
		if not self.max_capacity or self.capacity >= self.max_capacity:
			return False
		want = (st[:, 1] == 0) & (st[:, 2] + 12 > np.minimum(self._budget, self.capacity)) & (self._budget > self.capacity)
		if not want.any():
			return False
		cap = min(2 * self.capacity, self.max_capacity)
		path = max(self.max_path, 2 * cap) if self._auto_path else self.max_path
		ms = np.minimum(self._budget, cap)
		_ffi.check(_ffi.lib().rk_mcts_grow(self._h, cap, path, ms.ctypes.data, _ffi.stream_ptr()))
		self.capacity, self.max_path = cap, path
		self._shape = (self.n_trees, cap, path)
		self.grown += 1
		if self._graph is not None:                                   # the captured step holds the old pools' addresses
			self._graph = None
			if self._max_sims is None or self._max_sims - self.simulations > 1:
				self._capture(1)
		return True

	@no_grad
	def _finish(self) -> np.ndarray:
		_ffi.check(_ffi.lib().rk_mcts_set_expand_ahead(self._h, 0))
		if (self._max_sims is None or self.simulations < self._max_sims) and not self._poll()[:, 0].all():
			# stopped early (time) with leaves expanded ahead: one more step (net + backup, no further expansion) completes
			# them, so that every tree is the reference's tree after a whole number of simulations
			self._step(self._oh, self._h)
			self.simulations += 1
		st = self._poll()
		self._graph_done = False
		if self.search_graph and (st[:, 1] == 1).any():                 # agents.py:483-486, on the device
			_ffi.check(_ffi.lib().rk_mcts_search_graph(self._h, _ffi.stream_ptr()))
			self._graph_done = True
		return st[:, 1] != 0

	# -- results ------------------------------------------------------------------------------------------------
	def action_queue_of(self, tree: int) -> deque:
		buf = (C.c_longlong * self.max_path)()
		if getattr(self, "_graph_done", False) and self.status[tree, 1] == 1:
			n = _ffi.lib().rk_mcts_graph_path(self._h, tree, buf, self.max_path, _ffi.stream_ptr())
			if n >= 0:
				return deque(int(a) for a in buf[:n])
		n = _ffi.lib().rk_mcts_path(self._h, tree, buf, None, self.max_path, _ffi.stream_ptr())
		if n < 0:
			_ffi.check(int(n))
		return deque(int(a) for a in buf[:n])

	def path_nodes_of(self, tree: int) -> list:
		acts = (C.c_longlong * self.max_path)()
		nodes = (C.c_longlong * self.max_path)()
		_ffi.lib().rk_mcts_path(self._h, tree, acts, nodes, self.max_path, _ffi.stream_ptr())
		return [int(x) for x in nodes[:int(self.status[tree, 4])]]

	def tree_arrays(self, tree: int) -> dict:
		"""The reference's arrays of one tree, rows 0..n (row 0 unused), in the reference's dtypes."""
		n = int(self._poll()[tree, 2])
		out = dict(
			states=np.zeros((n + 1, 20), np.int8), neighbors=np.zeros((n + 1, 12), np.int64), leaves=np.ones(n + 1, np.uint8),
			P=np.zeros((n + 1, 12)), V=np.zeros(n + 1), N=np.zeros((n + 1, 12), np.int64), W=np.zeros((n + 1, 12)), L=np.zeros((n + 1, 12)))
		_ffi.check(_ffi.lib().rk_mcts_export(
			self._h, tree, 1, n, out["states"][1:].ctypes.data, out["neighbors"][1:].ctypes.data, out["leaves"][1:].ctypes.data,
			out["P"][1:].ctypes.data, out["V"][1:].ctypes.data, out["N"][1:].ctypes.data, out["W"][1:].ctypes.data,
			out["L"][1:].ctypes.data, _ffi.stream_ptr()))
		out["leaves"] = out["leaves"].astype(bool)
		out["n"] = n
		return out

	def __len__(self):
		return int(self.status[:, 2].sum()) if self.status is not None else 0

	def __str__(self):
		return f"Batched MCTS x{self.n_trees} (c={self.c})"


class MCTS(DeepAgent):
	"""
	The reference's single-tree agent (agents.py:415-645) on the device engine: `MCTS(net, c, search_graph)`,
	`search(state, time_limit, max_states)`, `action_queue`, `len(agent)`, and the arrays `states, neighbors, leaves,
	P, V, N, W, L, indices` for inspection (tests/test_agents.py:49-94).

	With default arguments the priors are computed exactly where and how the reference computes them (`priors="reference"`,
	see MCTSBatch: the root's softmax on the device, every other node's on the host, agents.py:472 and :551-552), so P equals
	the reference's bit for bit whenever the logits do; that costs one device-to-host wait per simulation.  `priors="kernel"`
	(softmax inside the backup kernel, no wait, hipGraph-capturable) is the fast form; `use_hipgraph=True` implies it unless
	`priors` is given.  The pool starts at `capacity` (default 200 000 nodes) and grows in place while the state budget allows
	(rk_mcts_grow, agents.py:450-460) up to `max_capacity`; graph completion and path shortening of a solved search
	(`search_graph`, agents.py:597-633) run on the device.
	"""
	default_capacity = 200_000
	max_capacity = 25_000_000           # 461 B per node: 11.5 GB

	def __init__(self, net, c: float, search_graph: bool, capacity: int = None, use_hipgraph: bool = False, torch_softmax: bool = False,
	             priors: str = None):
		super().__init__(net)
		self.priors = priors or ("torch" if torch_softmax else ("kernel" if use_hipgraph else "reference"))
		if self.priors not in PRIORS:
			raise ValueError(f"priors is one of {PRIORS}")
		self.c = c
		self.search_graph = search_graph
		self.nu = 100
		self.capacity = capacity
		self.use_hipgraph = use_hipgraph          # replay the simulation step as a captured hipGraph (the net must be capturable)
		self._batch = None
		self._arrays = None
		self._n = 0
		self.capacity_exhausted = False

	@property
	def grown(self) -> int:
		return self._batch.grown if self._batch is not None else 0

	def reset(self, time_limit: float, max_states: int):
		time_limit, max_states = super().reset(time_limit, max_states)
		self._arrays, self._n = None, 0
		return time_limit, max_states

	@no_grad
	def search(self, state: np.ndarray, time_limit: float = None, max_states: int = None) -> bool:
		time_limit, max_states = self.reset(time_limit, max_states)
		self.capacity_exhausted = False
		cap = max(int(min(max_states, self.capacity or self.default_capacity)), 13)      # a pool holds at least the root and its 12 children
		b = self._batch
		if b is None or b.capacity < cap or b.c != float(self.c) or b.priors != self.priors or b.search_graph != bool(self.search_graph) \
		   or b.max_capacity != self.max_capacity:
			b = self._batch = MCTSBatch(self.net, self.c, 1, capacity=cap, nu=self.nu, priors=self.priors, search_graph=self.search_graph,
			                            max_capacity=self.max_capacity)
		b.net = self.net
		# the pool grows in place inside the batch engine while the budget is larger (agents.py:496-503)
		solved = bool(b.search(np.asarray(state)[None], time_limit=time_limit, max_states=int(min(max_states, 2 ** 62)), poll=8,
		                       use_graph=self.use_hipgraph)[0])
		self._n = int(b.status[0, 2])
		if not solved and self._n + 12 > b.capacity and max_states > b.capacity and b.capacity >= self.max_capacity:
			self.capacity_exhausted = True
			import warnings
			warnings.warn(f"{self}: node pool of {b.capacity} states is full with time left; raise max_capacity", CapacityExhausted)
		self.action_queue = b.action_queue_of(0)       # with search_graph: completed and shortened on the device (agents.py:483-486)
		return solved

	# -- inspection -------------------------------------------------------------------------------------------
	def _export(self) -> dict:
		if self._arrays is None:
			self._arrays = self._batch.tree_arrays(0) if self._batch is not None and self._n else dict(
				states=np.zeros((1, 20), np.int8), neighbors=np.zeros((1, 12), np.int64), leaves=np.ones(1, bool),
				P=np.zeros((1, 12)), V=np.zeros(1), N=np.zeros((1, 12), np.int64), W=np.zeros((1, 12)), L=np.zeros((1, 12)), n=0)
		return self._arrays

	states = property(lambda self: self._export()["states"])
	neighbors = property(lambda self: self._export()["neighbors"])
	leaves = property(lambda self: self._export()["leaves"])
	P = property(lambda self: self._export()["P"])
	V = property(lambda self: self._export()["V"])
	N = property(lambda self: self._export()["N"])
	W = property(lambda self: self._export()["W"])
	L = property(lambda self: self._export()["L"])

	@property
	def indices(self) -> dict:
		"""state bytes -> index (the reference's dict, agents.py:419), rebuilt on the host from the exported states: inspection only"""
		st = self.states
		return {st[i].tobytes(): i for i in range(1, self._n + 1)}

	@classmethod
	def from_saved(cls, loc: str, use_best: bool, c: float, search_graph: bool, loader=None):
		"""agents.py:635-639."""
		return cls(_load_net(loc, use_best, loader), c=c, search_graph=search_graph)

	def __str__(self):
		return ("BFS" if self.search_graph else "Naive") + f" MCTS (c={self.c})"

	def __len__(self):
		return self._n


class AStarBatch(DeepAgent):
	"""
	S independent batch weighted A* searches advanced in lock-step on the GPU (engine rk_astarb_*).  Every search
	follows the reference's AStar exactly (agents.py:171-413): same node numbering, G, parents and action queue as
	running the reference on that start state alone, whenever the net returns the same numbers.

	One iteration of all searches = one fixed sequence of launches around one net forward on the padded
	(S * 12 N, 480) batch; nothing synchronises, so `use_graph=True` captures the iteration in a hipGraph and replays
	it.  The host polls the per-search status every `poll` iterations (searches that are solved or out of budget are
	skipped on the device in between).  This is the throughput form for evaluating many scrambles -- the reference's
	Evaluator runs its games one after the other.
	"""

	def __init__(self, net, lambda_: float, expansions: int, n_searches: int, capacity: int = 200_000, fused_first_layer=False):
		super().__init__(net, fused_first_layer)           # fused: the net's first Linear reads the new nodes' 20-byte states
		self.net_slice_rows = None                         # rows per net forward (None: NET_SLICE_ROWS)
		self.lambda_, self.expansions, self.n_searches = float(lambda_), int(expansions), int(n_searches)
		self.capacity = max(int(capacity), 12 * self.expansions + 2)
		self._h = None
		self.status = None
		self.iterations = 0
		self.on_poll = None                                # callable(status): called with every status the search reads

	def _engine(self):
		if self._h is None:
			h = C.c_void_p()
			_ffi.check(_ffi.lib().rk_astarb_create(C.byref(h), self.n_searches, self.capacity, self.expansions))
			self._h = h
		return self._h

	def __del__(self):
		try:
			if getattr(self, "_h", None) is not None:
				_ffi.lib().rk_astarb_destroy(self._h)
				self._h = None
		except Exception:
			pass

	def _poll(self) -> np.ndarray:
		st = np.zeros((self.n_searches, 7), np.int64)
		_ffi.check(_ffi.lib().rk_astarb_status(self._h, st.ctypes.data, _ffi.stream_ptr()))
		if st[:, 6].any():
			raise _ffi.RubiksHipError(f"batched A* engine error codes {st[:, 6].tolist()}")
		self.status = st
		if self.on_poll is not None:
			self.on_poll(st)
		return st

	def _step_exact(self, oh, code):
		"""One iteration with the net's batch compacted to the searches' NEW rows (one host wait for their count)."""
		lib, h = _ffi.lib(), self._h
		_ffi.check(lib.rk_astarb_step_expand_compact(h, oh.data_ptr(), code, self._n_rows.data_ptr(), _ffi.stream_ptr()))
		self._counted.record()
		self._counted.synchronize()
		rows = int(self._n_rows[0])
		self.net_rows_total += rows
		if rows == 0:
			values = self._no_values
		else:
			v = _sliced_value_forward(self._fs or self.net, oh[:min(len(oh), -(-rows // 64) * 64)], self.net_slice_rows or 12_288)
			if isinstance(v, torch.Tensor) and v.is_cuda and v.dtype == torch.bfloat16 and v.is_contiguous():
				values, vcode = v.detach().reshape(-1), _ffi.OH_BF16
			else:
				values, vcode = _value_f32(v), _ffi.OH_F32
			if vcode != self._vcode:
				_ffi.check(lib.rk_astarb_set_values_dtype(h, vcode, _ffi.stream_ptr()))
				self._vcode = vcode
		self._keep = values
		_ffi.check(lib.rk_astarb_step_commit(h, values.data_ptr(), _ffi.stream_ptr()))

	def _step(self, oh, code):
		lib, h = _ffi.lib(), self._h
		_ffi.check(lib.rk_astarb_step_expand(h, oh.data_ptr(), code, _ffi.stream_ptr()))
		self.net_rows_total += len(oh)
		# rows per forward: whole searches, as many as fit NET_SLICE_ROWS -- and exactly ONE search per forward once a search's
		# own batch is that large: then every forward has the shape a sequential AStar would give the net, so the batch can
		# not lose to it through the library's kernel choice (measured at N = 1000, 64 searches, bf16 fc_small: hipBLASLt and
		# torch's BatchNorm pick kernels for 16 384-row forwards that cost three times as much per row as for 12 000 rows)
		K = 12 * self.expansions
		rows = self.net_slice_rows or (K if K >= NET_SLICE_ROWS // 4 else (NET_SLICE_ROWS // K) * K)
		v = _sliced_value_forward(self._fs or self.net, oh, rows)
		if isinstance(v, torch.Tensor) and v.is_cuda and v.dtype == torch.bfloat16 and v.is_contiguous():
			values, vcode = v.detach().reshape(-1), _ffi.OH_BF16        # a bf16 net's values go in as they are
		else:
			values, vcode = _value_f32(v), _ffi.OH_F32
		if vcode != self._vcode:
			_ffi.check(lib.rk_astarb_set_values_dtype(h, vcode, _ffi.stream_ptr()))
			self._vcode = vcode
		self._keep = values
		_ffi.check(lib.rk_astarb_step_commit(h, values.data_ptr(), _ffi.stream_ptr()))

	@no_grad
	def search(self, states: np.ndarray, time_limit: float = None, max_states=None, use_graph: bool = False, poll: int = 8,
	           exact_batch: bool = None) -> np.ndarray:
		"""Runs all searches until each is solved or out of budget / time; returns the bool vector `solved` (S,).
		exact_batch: evaluate the net on exactly the new states of all searches (compacted on the device, one host wait per
		iteration) instead of on the padded (S * 12 N)-row batch.  Default: whenever the net is a torch module with parameters
		(a forward then costs far more than the wait) and the iteration is not replayed as a hipGraph."""
		_ffi.require_gpu()
		t0 = time.perf_counter()
		assert time_limit or max_states is not None
		self.net.eval()
		time_limit = time_limit or 1e10
		S, K = self.n_searches, 12 * self.expansions
		states = np.ascontiguousarray(states, dtype=np.int8).reshape(S, 20)
		budget = np.minimum(np.broadcast_to(np.asarray(self.capacity if max_states is None else max_states, dtype=np.int64), (S,)),
		                    self.capacity).copy()
		h, lib = self._engine(), _ffi.lib()
		_ffi.check(lib.rk_astarb_reset(h, states.ctypes.data, budget.ctypes.data, self.lambda_, _ffi.stream_ptr()))
		self._vcode = _ffi.OH_F32
		_ffi.check(lib.rk_astarb_set_values_dtype(h, self._vcode, _ffi.stream_ptr()))
		self._fs = self._from_states           # re-copied here if the net changed since the last search
		if self._fs is not None:
			oh, code = torch.from_numpy(cube.repeat_state(cube.get_solved(), S * K)).to(gpu), _ffi.OH_STATES
		else:
			oh_dtype = _oh_dtype(self.net)
			oh = torch.zeros((S * K, 480), dtype=oh_dtype, device=gpu)
			code = _OH_CODES[oh_dtype]
		self.iterations = 0
		self.net_rows_total = 0
		if exact_batch is None:
			exact_batch = not use_graph and callable(getattr(self.net, "parameters", None)) and any(True for _ in self.net.parameters())
		exact_batch = bool(exact_batch) and not use_graph
		if exact_batch:
			self._n_rows = torch.zeros(1, dtype=torch.int32).pin_memory()
			self._counted = torch.cuda.Event()
			self._no_values = torch.zeros(4, dtype=torch.float32, device=gpu)
		step = self._step_exact if exact_batch else self._step
		graph = None
		if use_graph:
			side = torch.cuda.Stream()
			side.wait_stream(torch.cuda.current_stream())
			with torch.cuda.stream(side):
				self._step(oh, code)                                    # a real iteration; also warms the allocator and fixes the values' dtype
			torch.cuda.current_stream().wait_stream(side)
			self.iterations += 1
			graph = torch.cuda.CUDAGraph()
			with torch.cuda.graph(graph, **CAPTURE):
				self._step(oh, code)
		# Steps after a search is done are no-ops on the device but still run the net on the padded batch, so the host must not
		# poll too rarely: a search grows by at most K states per iteration, so no live search can run out of budget in fewer
		# than (budget - states) // K iterations -- poll after that many (at most `poll`).  Round 2's harness polled every 64
		# iterations of a search that needs 12: that, not a kernel, was the 0.22x of profiles/r02_astar_batch.json at N = 1000.
		n_states = np.ones(S, np.int64)
		live = np.ones(S, bool)
		while time.perf_counter() - t0 < time_limit:
			safe = int(((budget[live] - n_states[live]) // K).min()) if live.any() else 1
			burst = max(1, min(poll, safe))
			for _ in range(burst):
				if graph is not None:
					graph.replay()
				else:
					step(oh, code)
			self.iterations += burst
			st = self._poll()
			live, n_states = st[:, 0] == 0, st[:, 2]
			if not live.any():
				break
		return self._poll()[:, 1] != 0

	def action_queue_of(self, search: int) -> deque:
		st = self.status[search]
		if st[1] != 1:                                                   # unsolved, or the start was already solved
			return deque()
		buf = (C.c_longlong * 4096)()
		n = _ffi.lib().rk_astarb_path(self._h, search, int(st[5]), buf, 4096, _ffi.stream_ptr())
		if n < 0:
			_ffi.check(int(n))
		return deque(int(a) for a in buf[:n])

	def arrays_of(self, search: int):
		"""(states, G, parents, parent_actions) of one search, rows 0..n (row 0 unused), in the reference's dtypes."""
		n = int(self._poll()[search, 2])
		states, G = np.zeros((n + 1, 20), np.int8), np.zeros(n + 1)
		parents, pact = np.zeros(n + 1, np.int64), np.zeros(n + 1, np.int64)
		_ffi.check(_ffi.lib().rk_astarb_export(self._h, search, 1, n, states[1:].ctypes.data, G[1:].ctypes.data, parents[1:].ctypes.data,
		                                       pact[1:].ctypes.data, _ffi.stream_ptr()))
		return states, G, parents, pact

	def __len__(self):
		return int(self.status[:, 2].sum()) if self.status is not None else 0

	def __str__(self):
		return f"Batched AStar x{self.n_searches} (lambda={self.lambda_}, N={self.expansions})"
