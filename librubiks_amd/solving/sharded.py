"""
Hash-sharded batch weighted A* across the GPUs of one node (BASELINE config 5).  No counterpart in the reference,
which is single-process; the per-rank semantics are those of `AStar` (reference: librubiks/solving/agents.py:171-413).

One process per GPU.  Every rank owns the states whose `rk_shard_owner(state, world)` is its rank: its own node pool,
hash table and sorted open queue (engine `rk_astar_*` in sharded mode).  One iteration, all ranks in lock-step:

  1. all-gather of the N cheapest open costs of every rank (+ pool sizes) -> every rank computes the same global
     top-N by (cost, rank, position) and knows how many of its own head entries are in it;
  2. `rk_astar_shard_pop`: expand those, bucket the 12 n children by owner as 32-byte records
     {state, parent index, g, action, parent rank};
  3. frontier exchange: counts, then records, `all_to_all_single` -- RCCL over xGMI with the "nccl" backend.  An
     all-to-all sends a different slice to each peer, so all 7 xGMI links of a GPU carry traffic at once; at N = 700
     a rank ships ~270 KB per iteration, i.e. the exchange is latency-, not bandwidth-bound;
  4. `rk_astar_shard_insert`: membership, first-occurrence de-duplication in arrival order, append, goal test,
     relaxation case 1 on the owner;
  5. value net on the new states of this rank, `rk_astar_shard_push` into the local open queue;
  6. relaxation case 2 (a seen child offers its parent a shortcut): 16-byte offers travel back to the parent's owner
     in a second, small all-to-all, `rk_astar_shard_apply_shortcuts`;
  7. all-gather of {won, solved index}: the rank that inserted the solved state reports it; the action queue is
     rebuilt by walking (rank, index) parent references with one small broadcast per hop.

With `world == 1` (no process group needed) the result equals `AStar` bit for bit -- that is how the sharded code
path is pinned to the reference on a single GPU.  With the "gloo" backend the buffers are staged through the host,
which lets several ranks share one GPU in tests; with "nccl" they stay on the device.
"""
from __future__ import annotations

import ctypes as C
import time
from collections import deque

import numpy as np
import torch
import torch.distributed as dist

from librubiks_amd import gpu, no_grad, _ffi, cube
from librubiks_amd.solving.agents import DeepAgent, _value_f32, _oh_dtype, _OH_CODES

REC_BYTES = 32          # child record
OFFER_BYTES = 16        # shortcut offer


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
		# nccl (RCCL) code path -- device tensors, variable-size all_to_all_single -- is exercised on a one-GPU box
		self.shortcut = self.world == 1 and not (force_collectives and self.active)

	def _dev(self):
		return gpu if self.on_device else torch.device("cpu")

	def all_gather_vec(self, vec: np.ndarray) -> np.ndarray:
		"""(world, len(vec)) array of every rank's float64 vector."""
		if self.shortcut:
			return vec[None].copy()
		mine = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.float64)).to(self._dev())
		if self.on_device:
			out = torch.empty((self.world, len(vec)), dtype=torch.float64, device=gpu)
			dist.all_gather_into_tensor(out, mine, group=self.group)
			return out.cpu().numpy()
		parts = [torch.empty_like(mine) for _ in range(self.world)]
		dist.all_gather(parts, mine, group=self.group)
		return torch.stack(parts).numpy()

	def exchange_counts(self, send_counts: np.ndarray) -> np.ndarray:
		if self.shortcut:
			return send_counts.copy()
		s = torch.from_numpy(np.ascontiguousarray(send_counts, dtype=np.int64)).to(self._dev())
		r = torch.empty_like(s)
		dist.all_to_all_single(r, s, group=self.group)
		return r.cpu().numpy()

	def exchange_records(self, send: torch.Tensor, send_counts: np.ndarray, recv_counts: np.ndarray) -> torch.Tensor:
		"""send: (n, width) uint8 on the GPU grouped by destination; returns (m, width) uint8 on the GPU grouped by source."""
		if self.shortcut:
			return send[:int(send_counts[0])]
		width = send.shape[1]
		n_out, n_in = int(send_counts.sum()), int(recv_counts.sum())
		src = send[:n_out]
		if not self.on_device:
			src = src.cpu()
		dst = torch.empty((n_in, width), dtype=torch.uint8, device=src.device)
		dist.all_to_all_single(dst, src.contiguous(), [int(x) for x in recv_counts], [int(x) for x in send_counts], group=self.group)
		return dst if self.on_device else dst.to(gpu)

	def broadcast_vec(self, vec: np.ndarray, src: int) -> np.ndarray:
		if self.shortcut:
			return vec
		t = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.int64)).to(self._dev())
		dist.broadcast(t, src=dist.get_global_rank(self.group, src) if self.group is not None else src, group=self.group)
		return t.cpu().numpy()


def select_pops(heads: np.ndarray, n: int) -> np.ndarray:
	"""
	heads: (world, n) float64, row r = the n cheapest open costs of rank r in ascending order, padded with +inf.
	Returns how many head entries each rank pops so that together they are the n globally cheapest by
	(cost, rank, position).  Deterministic and identical on every rank.
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

	def __init__(self, net, lambda_: float, expansions: int, capacity: int = 2_000_000, group=None, force_collectives: bool = False):
		super().__init__(net)
		self.lambda_, self.expansions, self.capacity = lambda_, int(expansions), int(capacity)
		self.tp = Transport(group, force_collectives)
		self._h = None
		self.iterations = 0
		self.total_states = 0
		self._n = 0

	def _engine(self):
		if self._h is None:
			h = C.c_void_p()
			_ffi.check(_ffi.lib().rk_astar_create_sharded(C.byref(h), self.capacity, self.expansions, self.tp.rank, self.tp.world))
			self._h = h
		return self._h

	def __del__(self):
		try:
			if self._h is not None:
				_ffi.lib().rk_astar_destroy(self._h)
				self._h = None
		except Exception:
			pass

	@no_grad
	def search(self, state: np.ndarray, time_limit: float = None, max_states: int = None) -> bool:
		_ffi.require_gpu()
		t0 = time.perf_counter()
		time_limit, max_states = self.reset(time_limit, max_states)
		self.iterations = 0
		state = np.ascontiguousarray(state, dtype=np.int8)
		if cube.is_solved(state):
			return True
		lib, tp, N = _ffi.lib(), self.tp, self.expansions
		h = self._engine()
		st = _ffi.stream_ptr
		_ffi.check(lib.rk_astar_shard_reset(h, state.ctypes.data, float(self.lambda_), st()))
		root_owner = lib.rk_shard_owner(state.ctypes.data, tp.world)
		k_out, k_in = 12 * N, 12 * N * tp.world
		send = torch.empty((k_out, REC_BYTES), dtype=torch.uint8, device=gpu)
		offers_out = torch.empty((k_in, OFFER_BYTES), dtype=torch.uint8, device=gpu)
		oh_dtype = _oh_dtype(self.net)
		oh = torch.empty((min(k_in, self.capacity), 480), dtype=oh_dtype, device=gpu)
		send_counts = np.zeros(tp.world, np.int64)
		offer_counts = np.zeros(tp.world, np.int64)
		info = np.zeros(5, np.int64)
		heads = np.empty(N + 2, np.float64)
		self._n = int(lib.rk_astar_size(h))

		while True:
			# 1. global selection (also carries pool sizes and the clock of rank 0)
			heads[:] = np.inf
			got = lib.rk_astar_export_open(h, heads[2:].ctypes.data, None, N, st())
			if got < 0:
				_ffi.check(int(got))
			heads[0] = self._n
			heads[1] = time.perf_counter() - t0
			allh = tp.all_gather_vec(heads)
			self.total_states = int(allh[:, 0].sum())
			if allh[0, 1] >= time_limit or self.total_states + N * 12 * tp.world > max_states:
				return False
			pops = select_pops(allh[:, 2:], N)
			if pops.sum() == 0:
				return False
			# 2. expand and bucket
			_ffi.check(lib.rk_astar_shard_pop(h, int(pops[tp.rank]), send.data_ptr(), send_counts.ctypes.data, st()))
			# 3. frontier exchange
			recv_counts = tp.exchange_counts(send_counts)
			recv = tp.exchange_records(send, send_counts, recv_counts)
			n_recv = int(recv_counts.sum())
			if n_recv > k_in:
				raise _ffi.RubiksHipError(f"rank {tp.rank} received {n_recv} records, scratch holds {k_in}")
			# 4. insert on the owner
			_ffi.check(lib.rk_astar_shard_insert(h, recv.data_ptr() if n_recv else None, n_recv, offers_out.data_ptr(),
			                                     offer_counts.ctypes.data, info.ctypes.data, st()))
			n_new, won, solved_idx, self._n = int(info[1]), int(info[2]), int(info[3]), int(info[4])
			self.iterations += 1
			# 7a. has anybody inserted the solved state?
			flags = tp.all_gather_vec(np.array([won, solved_idx], dtype=np.float64))
			winners = np.flatnonzero(flags[:, 0])
			if len(winners):
				self._walk(int(winners[0]), int(flags[winners[0], 1]), root_owner)
				self.total_states = int(tp.all_gather_vec(np.array([self._n], dtype=np.float64)).sum())
				return True
			# 5. value net on this rank's new states, push
			values = None
			if n_new:
				_ffi.check(lib.rk_astar_new_states_oh(h, oh.data_ptr(), _OH_CODES[oh_dtype], st()))
				values = _value_f32(self.net(oh[:n_new], policy=False, value=True))
			_ffi.check(lib.rk_astar_shard_push(h, values.data_ptr() if values is not None else None, st()))
			# 6. shortcut offers back to the parents' owners
			offers_in_counts = tp.exchange_counts(offer_counts)
			offers = tp.exchange_records(offers_out, offer_counts, offers_in_counts)
			n_off = int(offers_in_counts.sum())
			_ffi.check(lib.rk_astar_shard_apply_shortcuts(h, offers.data_ptr() if n_off else None, n_off, st()))

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

	def __len__(self):
		return self._n

	def __str__(self):
		return f"Sharded AStar x{self.tp.world} (lambda={self.lambda_}, N={self.expansions})"
