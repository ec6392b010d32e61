"""
Hash-sharded batch weighted A* across the GPUs of one node (BASELINE config 5).  No counterpart in the reference,
which is single-process; the per-rank semantics are those of `AStar` (reference: librubiks/solving/agents.py:171-413).

One process per GPU.  Every rank owns the states whose `rk_shard_owner(state, world)` is its rank: its own node pool,
hash table and open queue (engine `rk_astar_*` in sharded mode).  One iteration, all ranks in lock-step, is TWO
collectives on fixed-size device buffers and no host synchronisation (the host polls the stop decision every `poll`
iterations; steps after the decision are no-ops on the device):

  1. all-gather of 8 + N doubles per rank, written by the engine at the end of the previous iteration: pool size, win
     flag, solved index, error flag, elapsed time of rank 0, and the rank's N cheapest open costs;
  2. `rk_astar_shard_select`: every rank derives the same stop decision (won / state budget / a rank's pool could
     overflow / time / error / nothing open) and the same global top-N by (cost, rank, position), expands its share and
     buckets the 12 n children by owner as 32-byte records {state, parent index, g, action, parent rank};
  3. frontier exchange: ONE `all_to_all_single` with equal splits -- per peer a block of {header with the record and
     offer counts, <= 12 N records, <= 12 N shortcut offers}.  RCCL over xGMI with the "nccl" backend: an all-to-all
     sends a different block to each peer, so all 7 xGMI links of a GPU carry traffic at once.  The blocks are padded to
     their upper bound (12 N * 48 B per peer: 0.4 MB at N = 700), which is what removes the count exchange and every
     host round trip; at these sizes the exchange is latency-, not bandwidth-bound;
  4. `rk_astar_shard_insert`: first the shortcut offers that arrived (relaxation case 2 of the PREVIOUS iteration, on
     the parents' owner), then membership, first-occurrence de-duplication in arrival order, append, goal test,
     relaxation case 1, one-hot of the new states;
  5. value net on the NEW states only, `rk_astar_shard_push_rows`: cost, push into the local queue, this iteration's shortcut
     offers into the send blocks (they ride on the next all-to-all, so relaxation case 2 costs no collective of its
     own), next candidates, next all-gather contribution.  All ranks together pop at most N nodes, so a rank can receive
     at most 12 N children and appends at most 12 N new states, whatever the world size: the net batch is bounded by
     12 N rows, not world * 12 N.  A rank EXPECTS 12 N / world of them (owner = hash), so the net runs on a FIXED number of
     rows a little above that (`net_rows`: mean + 6 sigma of the multinomial share + 64, rounded up to 64) -- no count travels
     to the host and nothing waits (round 4 read the count back every iteration and blocked on it).  Should an iteration ever
     bring a rank more new states than rows, the engine says so in its all-gather contribution (error 3), every rank stops
     at the next decision, and the driver repeats the search with the full 12 N-row batch: the search is a pure function of
     its inputs, so the repeat gives what a run without the shortfall would have given.

The whole iteration -- all-gather, select, all-to-all, insert, net, push -- is a fixed sequence of launches on fixed buffers with
nothing written by the host (rank 0's clock for the time limit is the device's, started by the engine at the reset), so with device
collectives (`nccl`, `rk_comm`) or world = 1 it is captured ONCE as a hipGraph (`use_hipgraph=True`) and replayed: one host launch
per iteration, one 64-byte read of the decision every `poll` iterations.  The graph is kept from search to search (it holds
addresses and the by-value scalars lambda / limits / row count, nothing of a search).

The action queue is rebuilt by walking (rank, index) parent references with one small broadcast per hop.

With `world == 1` (no process group needed; the send buffer is the receive buffer) the result equals `AStar` bit for
bit -- that is how the sharded code path is pinned to the reference on a single GPU.  With the "gloo" backend the
buffers are staged through the host, which lets several ranks share one GPU in tests; with "nccl" they stay on the
device.
"""
from __future__ import annotations

import ctypes as C
import time
from collections import deque

import numpy as np
import torch
import torch.distributed as dist

from librubiks_amd import gpu, no_grad, _ffi, cube
from librubiks_amd.solving.agents import CAPTURE, DeepAgent, _capture_key, _values_for_engine, _sliced_value_forward, _oh_dtype, _OH_CODES

STOP_REASONS = {0: "running", 1: "won", 2: "budget", 3: "capacity", 4: "time", 5: "nothing open", 6: "engine error"}


class Transport:
	"""The collectives the search needs, over torch.distributed (or nothing when world == 1)."""

	def __init__(self, group=None, force_collectives: bool = False):
		self.group = group
		self.active = dist.is_available() and dist.is_initialized()
		self.world = dist.get_world_size(group) if self.active else 1
		self.rank = dist.get_rank(group) if self.active else 0
		self.backend = dist.get_backend(group) if self.active else "local"
		self.on_device = self.backend == "nccl"
		# world == 1 normally short-circuits every collective; `force_collectives` runs them anyway, which is how the
		# nccl (RCCL) code path -- device tensors, all_gather_into_tensor, all_to_all_single -- is exercised on a one-GPU box
		self.shortcut = self.world == 1 and not (force_collectives and self.active)
		self.collectives = 0

	def all_gather(self, mine: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
		"""(world, len(mine)) device tensor of every rank's vector; into `out` when given (the search loop's preallocated buffer)."""
		if self.shortcut:
			return mine.view(1, -1)
		self.collectives += 1
		if out is None:
			out = torch.empty((self.world, mine.numel()), dtype=mine.dtype, device=mine.device)
		if self.on_device:
			dist.all_gather_into_tensor(out, mine, group=self.group)
			return out
		host = mine.cpu()
		parts = [torch.empty_like(host) for _ in range(self.world)]
		dist.all_gather(parts, host, group=self.group)
		out.copy_(torch.stack(parts))
		return out

	def all_to_all(self, send: torch.Tensor, recv: torch.Tensor) -> torch.Tensor:
		"""send, recv: (world, block) uint8 device tensors; row p of `send` goes to rank p, row q of `recv` comes from rank q."""
		if self.shortcut:
			return send
		self.collectives += 1
		if self.on_device:
			dist.all_to_all_single(recv, send, group=self.group)
			return recv
		src = send.cpu()
		dst = torch.empty_like(src)
		dist.all_to_all_single(dst, src, group=self.group)
		recv.copy_(dst)
		return recv

	def broadcast_vec(self, vec: np.ndarray, src: int) -> np.ndarray:
		if self.shortcut:
			return vec
		t = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.int64)).to(gpu if self.on_device else "cpu")
		dist.broadcast(t, src=dist.get_global_rank(self.group, src) if self.group is not None else src, group=self.group)
		return t.cpu().numpy()


class RcclTransport:
	"""
	The same three transfers through the C ABI's own RCCL layer (`rk_comm_*`, include/rubiks_hip.h) instead of
	torch.distributed: what a caller of the shared library without torch.distributed uses (route B).
	Rank 0 obtains the 128-byte id with `RcclTransport.unique_id()` and hands it to the other ranks by any means (a file, a
	socket, MPI); every rank then constructs `RcclTransport(id, rank, world)` -- a collective call.  Device buffers only,
	on the current stream, nothing synchronises.  `ShardedAStar(..., transport=...)` takes it.
	"""
	on_device, backend, active = True, "rccl (rk_comm)", True

	@staticmethod
	def unique_id() -> bytes:
		buf = C.create_string_buffer(128)
		_ffi.check(_ffi.lib().rk_comm_unique_id(buf))
		return buf.raw

	def __init__(self, unique_id: bytes, rank: int, world: int):
		if len(unique_id) != 128:
			raise ValueError("the RCCL unique id is 128 bytes")
		_ffi.require_gpu()
		h = C.c_void_p()
		_ffi.check(_ffi.lib().rk_comm_create(C.byref(h), unique_id, int(rank), int(world)))
		self._h, self.rank, self.world = h, int(rank), int(world)
		self.shortcut = False                      # the collectives always run, also with one rank: that is how a one-GPU box tests them
		self.collectives = 0

	def __del__(self):
		try:
			if getattr(self, "_h", None) is not None:
				_ffi.lib().rk_comm_destroy(self._h)
				self._h = None
		except Exception:
			pass

	def all_gather(self, mine: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
		self.collectives += 1
		if out is None:
			out = torch.empty((self.world, mine.numel()), dtype=mine.dtype, device=mine.device)
		_ffi.check(_ffi.lib().rk_comm_all_gather(self._h, mine.data_ptr(), out.data_ptr(), mine.numel() * mine.element_size(), _ffi.stream_ptr()))
		return out

	def all_to_all(self, send: torch.Tensor, recv: torch.Tensor) -> torch.Tensor:
		self.collectives += 1
		_ffi.check(_ffi.lib().rk_comm_all_to_all(self._h, send.data_ptr(), recv.data_ptr(), send.shape[1] * send.element_size(), _ffi.stream_ptr()))
		return recv

	def broadcast_vec(self, vec: np.ndarray, src: int) -> np.ndarray:
		t = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.int64)).to(gpu)
		_ffi.check(_ffi.lib().rk_comm_broadcast(self._h, t.data_ptr(), t.numel() * 8, int(src), _ffi.stream_ptr()))
		return t.cpu().numpy()


def net_rows(K: int, world: int) -> int:
	"""
	Rows of a rank's net batch, FIXED for the whole search: all K = 12 N with one rank; else the rank's expected share of the at
	most K new states of an iteration (owner = hash: multinomial, mean K / world) plus six standard deviations plus 64, rounded
	up to 64, never more than K.  N = 700 on 8 ranks: 1 344 of 8 400 rows; a shortfall has a probability of about 1e-9 per rank
	and iteration, is detected on the device (rk_astar_shard_push_rows) and costs a repeated search, never a wrong one.
	"""
	if world <= 1:
		return K
	mu = K / world
	rows = mu + 6.0 * (mu * (1.0 - 1.0 / world)) ** 0.5 + 64.0
	return min(K, -(-int(rows + 0.999999) // 64) * 64)


def select_pops(heads: np.ndarray, n: int) -> np.ndarray:
	"""
	Host statement of the selection rule the device kernel `k_shard_decide` implements (used by the tests as its spec):
	heads is (world, n) float64, row r = the n cheapest open costs of rank r in ascending order, padded with +inf.
	Returns how many head entries each rank pops so that together they are the n globally cheapest by
	(cost, rank, position).
	"""
	world = heads.shape[0]
	cost = heads.ravel()
	rank = np.repeat(np.arange(world), heads.shape[1])
	pos = np.tile(np.arange(heads.shape[1]), world)
	order = np.lexsort((pos, rank, cost))
	take = order[:n]
	take = take[np.isfinite(cost[take])]
	return np.bincount(rank[take], minlength=world).astype(np.int64)


class ShardedAStar(DeepAgent):
	"""Collective agent: every rank constructs it and calls `search` with the same arguments."""

	def __init__(self, net, lambda_: float, expansions: int, capacity: int = 2_000_000, group=None, force_collectives: bool = False,
	             poll: int = 1, profile: bool = False, fused_first_layer=False, transport=None, use_hipgraph: bool = False, full_rows: bool = False):
		# fused_first_layer (True / "epilogue" / "folded"): the net's first Linear reads the new nodes' 20-byte states
		super().__init__(net, fused_first_layer)
		self.lambda_, self.expansions, self.capacity = lambda_, int(expansions), int(capacity)
		self.tp = transport if transport is not None else Transport(group, force_collectives)      # torch.distributed unless told otherwise
		self.poll = max(1, int(poll))
		self.profile = profile                 # record device-time per phase (HIP events); read `self.phase_ms` afterwards (eager iterations)
		self.phase_ms = {}
		# use_hipgraph: replay the iteration as ONE hipGraph launch (device collectives or world = 1 only; host-staged gloo cannot be
		# captured).  A capture that fails (a collective the stack cannot capture) is reported in `graph_error` and the search runs eagerly.
		self.use_hipgraph = bool(use_hipgraph)
		self.graph_error = None
		self.captures = 0
		self.host_launches = 0                 # launches the host issued during the last search's loop (graph replays or eager calls into the library / torch)
		self.full_rows = bool(full_rows)       # evaluate the net on all 12 N rows (what a search repeated after a row shortfall does)
		self.repeated = 0                      # searches repeated with the full-width batch after a row shortfall
		self.rows_override = None              # tests: a row count small enough for the shortfall to happen
		self._h = None
		self._bufs = None
		self._graph_cache = None
		self.iterations = 0
		self.total_states = 0
		self.stop_reason = "running"
		self._n = 0
		self.net_rows_max = 0                  # rows pushed through the net: per iteration (fixed) / sum over the last search
		self.net_rows_total = 0

	def _engine(self):
		if self._h is None:
			h = C.c_void_p()
			_ffi.check(_ffi.lib().rk_astar_create_sharded(C.byref(h), self.capacity, self.expansions, self.tp.rank, self.tp.world))
			self._h = h
		return self._h

	def __del__(self):
		try:
			self._graph_cache = None
			if self._h is not None:
				_ffi.lib().rk_astar_destroy(self._h)
				self._h = None
		except Exception:
			pass

	def _buffers(self, h, code, oh_dtype):
		"""The search's device buffers, allocated once per engine and row type: a kept hipGraph holds their addresses."""
		lib, tp, K = _ffi.lib(), self.tp, 12 * self.expansions
		if self._bufs is not None and self._bufs["code"] == code:
			return self._bufs
		self._graph_cache = None
		block, glen = int(lib.rk_astar_shard_block_bytes(h)), int(lib.rk_astar_shard_gather_len(h))
		send = torch.zeros((tp.world, block), dtype=torch.uint8, device=gpu)
		mine = torch.zeros(glen, dtype=torch.float64, device=gpu)
		if code == _ffi.OH_STATES:
			oh = torch.from_numpy(cube.repeat_state(cube.get_solved(), K)).to(gpu)                  # rows = states: valid codes everywhere
		else:
			oh = torch.zeros((K, 480), dtype=oh_dtype, device=gpu)
		self._bufs = {"code": code, "send": send, "recv": send if tp.shortcut else torch.zeros_like(send), "mine": mine,
		              "gathered": mine.view(1, -1) if tp.shortcut else torch.zeros((tp.world, glen), dtype=torch.float64, device=gpu), "oh": oh}
		_ffi.check(lib.rk_astar_shard_bind(h, mine.data_ptr()))
		return self._bufs

	def _iteration(self, h, b, forward, rows, time_limit, max_states, marks=None):
		"""One iteration: a fixed sequence of launches on fixed buffers, nothing from the host but the launches themselves."""
		lib, tp, st = _ffi.lib(), self.tp, _ffi.stream_ptr

		def mark():
			if marks is not None:
				e = torch.cuda.Event(enable_timing=True)
				e.record()
				marks.append(e)

		mark()
		gathered = tp.all_gather(b["mine"], b["gathered"])               # collective 1: heads + status
		mark()
		_ffi.check(lib.rk_astar_shard_select(h, gathered.data_ptr(), float(time_limit), float(max_states), b["send"].data_ptr(), st()))
		mark()
		got = tp.all_to_all(b["send"], b["recv"])                        # collective 2: records (+ last iteration's offers)
		mark()
		_ffi.check(lib.rk_astar_shard_insert(h, got.data_ptr(), b["send"].data_ptr(), b["oh"].data_ptr(), b["code"], st()))
		mark()
		values = _values_for_engine(h, _sliced_value_forward(forward, b["oh"][:rows]))
		self._keep = values                                              # the push kernels read it after this call returns
		mark()
		_ffi.check(lib.rk_astar_shard_push_rows(h, values.data_ptr(), rows, got.data_ptr(), b["send"].data_ptr(), st()))
		mark()

	@no_grad
	def search(self, state: np.ndarray, time_limit: float = None, max_states: int = None) -> bool:
		_ffi.require_gpu()
		limits = (time_limit, max_states)
		time_limit, max_states = self.reset(time_limit, max_states)
		self.iterations, self.stop_reason = 0, "running"
		state = np.ascontiguousarray(state, dtype=np.int8)
		if cube.is_solved(state):
			self.stop_reason = "won"
			return True
		lib, tp, N = _ffi.lib(), self.tp, self.expansions
		h = self._engine()
		st = _ffi.stream_ptr
		K = 12 * N                                                       # upper bound of a rank's new states per iteration
		rows = K if self.full_rows else min(K, self.rows_override or net_rows(K, tp.world))   # rows the net evaluates every iteration
		self._fs = self._from_states                                     # re-copied here if the net changed since the last search
		if self._fs is not None:
			code, oh_dtype, forward = _ffi.OH_STATES, None, self._fs
		else:
			oh_dtype = _oh_dtype(self.net)
			code, forward = _OH_CODES[oh_dtype], self.net
		b = self._buffers(h, code, oh_dtype)
		_ffi.check(lib.rk_astar_shard_reset(h, state.ctypes.data, float(self.lambda_), b["send"].data_ptr(), st()))
		root_owner = lib.rk_shard_owner(state.ctypes.data, tp.world)
		self.net_rows_max, self.net_rows_total, self.host_launches = rows, 0, 0
		decision = (C.c_longlong * 8)()
		marks = []                                                       # per eager iteration: events between the phases

		graph = None
		if self.use_hipgraph and not self.profile and (tp.shortcut or tp.on_device):
			# the captured iteration holds addresses (engine, buffers, the net's tensors) and scalars passed by value (lambda, the two
			# limits, the row count, the values' dtype) -- nothing of the search, which lives in device memory the reset rewrites
			key = (h.value, rows, code, float(self.lambda_), float(time_limit), float(max_states), b["oh"].data_ptr(), _capture_key(self.net, self._fs))
			if self._graph_cache is not None and self._graph_cache[0] == key:
				graph = self._graph_cache[1]
			else:
				self._graph_cache = None
				try:
					side = torch.cuda.Stream()
					side.wait_stream(torch.cuda.current_stream())
					with torch.cuda.stream(side):
						self._iteration(h, b, forward, rows, time_limit, max_states)      # a real iteration; also warms the allocator and the collectives
					torch.cuda.current_stream().wait_stream(side)
					graph = torch.cuda.CUDAGraph()
					with torch.cuda.graph(graph, **CAPTURE):
						self._iteration(h, b, forward, rows, time_limit, max_states)
					self._graph_cache = (key, graph, (self.net, self._fs))
					self.captures += 1
					self.net_rows_total += rows
				except Exception as e:                                      # e.g. a collective this stack cannot capture: say so, run eagerly
					self.graph_error = f"{type(e).__name__}: {e}"[:300]
					self.use_hipgraph, graph, self._graph_cache = False, None, None
					torch.cuda.synchronize()
					return self.search(state, *limits)                        # the engine may be mid-iteration: start over, eagerly

		stop, it = 0, 0
		while True:
			for _ in range(self.poll):
				if graph is not None:
					graph.replay()
					self.host_launches += 1
				else:
					row = [] if self.profile else None
					self._iteration(h, b, forward, rows, time_limit, max_states, row)
					if row is not None and len(marks) < 4096:
						marks.append(row)
				it += 1
				self.net_rows_total += rows
			_ffi.check(lib.rk_astar_shard_decision(h, decision, st()))       # the only host synchronisation: every `poll` iterations
			stop = int(decision[0])
			if stop:
				break                                                    # (iterations after the decision were no-ops on the device)
		self.stop_reason = STOP_REASONS.get(stop, str(stop))
		self.total_states, self.iterations, self._n = int(decision[3]), int(decision[5]), int(decision[6])
		if self.profile and marks:
			torch.cuda.synchronize()
			names = ("all_gather", "select+expand", "all_to_all", "insert", "net", "push")
			tot = np.zeros(len(names))
			for row in marks:
				tot += [row[i].elapsed_time(row[i + 1]) for i in range(len(names))]
			self.phase_ms = {n: float(t / len(marks)) for n, t in zip(names, tot)}
			self.phase_ms["iterations_timed"] = len(marks)
		if stop == 6 and int(decision[7]) == 3 and not self.full_rows:
			# a rank received more new states than the net evaluated rows for: repeat with the full-width batch (same result as a
			# run without the shortfall: the search is deterministic); the agent keeps the full width from now on
			self.full_rows, self._graph_cache = True, None
			self.repeated += 1
			return self.search(state, *limits)
		if stop == 6:
			raise _ffi.RubiksHipError(f"rank {tp.rank}: a rank reported engine error {int(decision[7])}; every rank stops together")
		if stop == 1:
			self._walk(int(decision[1]), int(decision[2]), root_owner)
			return True
		# no win: shortcut offers may still sit in the send blocks; deliver and apply them
		got = tp.all_to_all(b["send"], b["recv"])
		_ffi.check(lib.rk_astar_shard_flush(h, got.data_ptr(), st()))
		_ffi.check(lib.rk_astar_shard_clear_send(h, b["send"].data_ptr(), 1, 1, st()))
		return False

	def _walk(self, rank: int, idx: int, root_owner: int):
		"""Action queue from (rank, idx) back to the root; one 3-int broadcast per hop (agents.py:244-251)."""
		tp, lib = self.tp, _ffi.lib()
		queue = deque()
		hop = np.zeros(3, np.int64)
		for _ in range(10_000):
			if rank == root_owner and idx == 1:
				break
			if tp.rank == rank:
				_ffi.check(lib.rk_astar_shard_parent(self._h, idx, hop.ctypes.data, _ffi.stream_ptr()))
			hop = tp.broadcast_vec(hop, rank).copy()
			queue.appendleft(int(hop[2]))
			rank, idx = int(hop[0]), int(hop[1])
		else:
			raise _ffi.RubiksHipError("parent chain does not reach the root")
		self.action_queue = queue

	# -- inspection of this rank's shard ------------------------------------------------------------------------
	def local_arrays(self):
		"""(states, G, parents, parent_actions) of the states this rank owns, rows 1..n."""
		n = self._n
		states, G = np.zeros((n + 1, 20), np.int8), np.zeros(n + 1)
		parents, pact = np.zeros(n + 1, np.int64), np.zeros(n + 1, np.int64)
		if n:
			_ffi.check(_ffi.lib().rk_astar_export(self._h, 1, n, states[1:].ctypes.data, G[1:].ctypes.data, parents[1:].ctypes.data,
			                                      pact[1:].ctypes.data, _ffi.stream_ptr()))
		return states, G, parents, pact

	def local_parent_ranks(self) -> np.ndarray:
		"""Owner rank of every local node's parent, rows 1..n (row 0 unused): with `local_arrays` the whole shard."""
		n = self._n
		out = np.zeros(n + 1, np.int64)
		if n:
			_ffi.check(_ffi.lib().rk_astar_shard_export_ranks(self._h, 1, n, out[1:].ctypes.data, _ffi.stream_ptr()))
		return out

	def __len__(self):
		return self._n

	def __str__(self):
		return f"Sharded AStar x{self.tp.world} (lambda={self.lambda_}, N={self.expansions})"


class PartitionedMCTS:
	"""
	`MCTSBatch` across the GPUs of one node (BASELINE config 4 at N > 1): the trees are independent searches, so they are
	PARTITIONED, not sharded -- rank r owns the trees r, r + world, r + 2 world, ... of the batch and runs them with its
	own `MCTSBatch` engine; no collective in the search loop.  One all-gather of fixed size at the end hands every rank
	the whole batch's results (solved flag, explored states, simulations and the action queue of every tree), so the
	caller sees what one `MCTSBatch` over all trees would have returned; every tree is the reference's MCTS run on that
	start state alone (agents.py:415-645), whichever rank ran it.
	"""

	def __init__(self, net, c: float, n_trees: int, capacity: int = 50_000, max_path: int = None, group=None, force_collectives: bool = False, **kw):
		from librubiks_amd.solving.agents import MCTSBatch
		self.tr = Transport(group, force_collectives)       # force_collectives: run the final all-gather also with one rank (rehearsals on one GPU)
		self.n_trees = int(n_trees)
		self.mine = np.arange(self.tr.rank, self.n_trees, self.tr.world)
		self.max_path = int(max_path or 4096)
		# the engine's path limit IS the row width of the final all-gather: a tree cannot report more moves than fit a row
		self.local = MCTSBatch(net, c, max(len(self.mine), 1), capacity=capacity, max_path=self.max_path, **kw)
		self.solved = self.states = self.sims = None
		self._queues = None

	def search(self, states: np.ndarray, time_limit: float = None, max_states=None, max_sims: int = None, **kw) -> np.ndarray:
		states = np.ascontiguousarray(states, np.int8).reshape(self.n_trees, 20)
		per = -(-self.n_trees // self.tr.world)                              # trees per rank, rounded up: fixed-size rows
		row = 4 + self.max_path                                             # solved, states, sims, path length, actions
		out = np.full((per, row), -1, np.int32)
		if len(self.mine):
			solved = self.local.search(states[self.mine], time_limit, max_states, max_sims, **kw)
			st = self.local.status
			for k in range(len(self.mine)):
				q = list(self.local.action_queue_of(k)) if solved[k] else []
				if len(q) > self.max_path:
					# never raise before the collective (the other ranks would wait in it for ever): the row carries the error
					out[k, :4] = (-2, int(st[k, 2]), int(st[k, 3]), len(q))
					continue
				out[k, :4] = (int(solved[k]), int(st[k, 2]), int(st[k, 3]), len(q))
				out[k, 4:4 + len(q)] = q
		every = self.tr.all_gather(torch.from_numpy(out).reshape(-1).to(gpu if self.tr.on_device else "cpu"))
		every = every.cpu().numpy().reshape(self.tr.world, per, row)
		too_long = [(r, k, int(every[r, k, 3])) for r in range(self.tr.world) for k in range(per) if every[r, k, 0] == -2]
		if too_long:                                                     # the same rows on every rank: all of them raise
			raise RuntimeError(f"solutions longer than max_path = {self.max_path} (rank, local tree, moves): {too_long}")
		self.solved, self.states, self.sims = (np.zeros(self.n_trees, t) for t in (bool, np.int64, np.int64))
		self._queues = [deque() for _ in range(self.n_trees)]
		for r in range(self.tr.world):
			for k, tree in enumerate(range(r, self.n_trees, self.tr.world)):
				rec = every[r, k]
				self.solved[tree], self.states[tree], self.sims[tree] = bool(rec[0]), rec[1], rec[2]
				self._queues[tree] = deque(int(a) for a in rec[4:4 + rec[3]])
		return self.solved

	def action_queue_of(self, tree: int) -> deque:
		return self._queues[tree]

	def __len__(self):
		return int(self.states.sum()) if self.states is not None else 0

	def __str__(self):
		return f"MCTS x{self.n_trees} partitioned over {self.tr.world} ranks"
