"""
CPU oracle of the HASH-SHARDED batch weighted A* -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (rules: see cube_oracle.py).

The reference (librubiks/solving/agents.py:171-413) is a single process; sharding its open and closed set over the GPUs of
a node is this project's own protocol (include/rubiks_hip.h "hash-sharded A*", librubiks_amd/csrc/rk_astar.hip section
"Hash-sharded search", librubiks_amd/solving/sharded.py).  This file restates that PROTOCOL in plain Python, all `world`
ranks in one process, one explicit loop per step, so that the device engines can be compared with something at world > 1:
per-rank states / G / parents / parent ranks / actions, the nodes every rank pops in every iteration, and the stop decision.

Per rank the semantics are the reference's (`AStarOracle` in search_oracle.py is the same code for one rank):
  * children in pop-major, action-minor order (agents.py:277-282); only the FIRST occurrence of a state in a batch counts
    (np.unique(..., return_index=True), agents.py:291-295) -- here the batch of a rank is what ARRIVES at it: the records of
    rank 0, then rank 1, ..., each group in its sender's order;
  * unseen first occurrences are appended in that order with G / parent / action (agents.py:299-313), goal test on the new
    states (:321-323);
  * relaxation case 1 (agents.py:354-359) on the owner of the seen state, evaluated in full before it is applied;
  * relaxation case 2 (agents.py:362-367: the seen child offers its expanded PARENT a shorter way) crosses ranks: the child's
    owner sends a 16-byte OFFER to the parent's owner; offers ride on the NEXT iteration's exchange and are applied there
    first, all evaluated against G as it stands, the LAST hit per parent in arrival order winning (NumPy's fancy assignment
    with repeated indices); a search that ends without a win delivers the pending ones with one more exchange.
What is new with several ranks, restated here from the kernels:
  * owner(state) = ((hash(state) * 0x9E3779B1 mod 2^32) * world) >> 32, hash = the 64-bit mix of rk_search_dev.h:hash_state;
  * every iteration each rank publishes its pool size, win flag and the min(N, |open|) cheapest open costs; every rank
    derives the same STOP decision (won / budget: total + 12 N > max_states / capacity: largest pool + 12 N > capacity /
    nothing open) and the same global top-N by (cost, rank, position in the rank's list): that many of its own candidates a
    rank pops (k_shard_decide);
  * a rank that has won stops relaxing (it returns like the reference, agents.py:321-323); the others do not know yet.

Parity status: PINNED AT WORLD = 1 (tests/test_sharded_oracle_cpu.py: equal to AStarOracle, which is pinned to the
unmodified reference's traces, array for array and pop for pop).  For world > 1 the reference has nothing to compare with
("parity unpinned" by reference tests): this restatement IS the specification the device engines are held to.
"""
from __future__ import annotations

import heapq
from collections import deque

import numpy as np

from oracle import cube_oracle as orc

M64 = (1 << 64) - 1
STOP_NO, STOP_WON, STOP_BUDGET, STOP_CAPACITY, STOP_TIME, STOP_EMPTY, STOP_ERROR = range(7)


def hash_state(state: np.ndarray) -> int:
	"""rk_search_dev.h:hash_state over the five little-endian dwords of the 20-byte state."""
	h = 0x9E3779B97F4A7C15
	for w in np.frombuffer(np.ascontiguousarray(state, np.int8).tobytes(), dtype="<u4"):
		h ^= int(w)
		h = (h * 0xFF51AFD7ED558CCD) & M64
		h ^= h >> 29
	return (h ^ (h >> 32)) & 0xFFFFFFFF


def owner_of(state: np.ndarray, world: int) -> int:
	"""rk_search_dev.h:owner_of: a remix of the hash, scaled to 0..world-1."""
	return (((hash_state(state) * 0x9E3779B1) & 0xFFFFFFFF) * world) >> 32


def _values(net, states: np.ndarray) -> np.ndarray:
	v = net(orc.as_oh(states), policy=False, value=True)
	return np.asarray(v, dtype=np.float32).reshape(-1)


class _Rank:
	"""One rank's shard: node arrays 1-based like the reference's, an open heap of (cost, idx), the offers it will send."""

	def __init__(self, rank: int, world: int):
		self.rank, self.world = rank, world
		self.index = {}
		self.states = [None]
		self.G = [0]
		self.parents = [0]
		self.prank = [0]
		self.pact = [0]
		self.open = []
		self.won = False
		self.solved_idx = 0
		self.offers_out = [[] for _ in range(world)]      # built by this iteration's push, delivered with the next exchange

	def __len__(self):
		return len(self.states) - 1

	def add(self, state, g, parent, prank, action) -> int:
		idx = len(self.states)
		self.states.append(np.array(state, np.int8))
		self.G.append(int(g))
		self.parents.append(int(parent))
		self.prank.append(int(prank))
		self.pact.append(int(action))
		self.index[self.states[idx].tobytes()] = idx
		return idx

	def candidates(self, n: int):
		"""The min(n, |open|) cheapest open records in heappop order, without popping them."""
		return heapq.nsmallest(n, self.open)


class ShardedAStarOracle:
	def __init__(self, net, lambda_: float, expansions: int, world: int):
		self.net, self.lambda_, self.N, self.world = net, lambda_, int(expansions), int(world)

	def search(self, start: np.ndarray, max_states: int, capacity: int = None) -> int:
		"""Runs until a stop decision; returns the stop reason (STOP_*).  `capacity` = states a rank's pool holds (default: no limit)."""
		W, N = self.world, self.N
		start = np.asarray(start, np.int8)
		self.ranks = [_Rank(r, W) for r in range(W)]
		self.root_owner = owner_of(start, W)
		root = self.ranks[self.root_owner]
		root.add(start, 0, 0, self.root_owner, 0)
		heapq.heappush(root.open, (0.0, 1))                                 # heappush(open_queue, (0, 1)), agents.py:234
		self.pops = []              # per iteration: list over ranks of the node indices popped, in pop order
		self.new_counts = []        # per iteration: list over ranks of the number of states appended
		self.iterations = 0
		self.winner = None
		self.action_queue = deque()
		capacity = capacity if capacity is not None else 1 << 60
		while True:
			# ---- all-gather + k_shard_decide -------------------------------------------------------------------
			cands = [rk.candidates(N) for rk in self.ranks]
			total = sum(len(rk) for rk in self.ranks)
			biggest = max(len(rk) for rk in self.ranks)
			winner = next((r for r, rk in enumerate(self.ranks) if rk.won), None)
			if winner is not None:
				self.stop = STOP_WON
				self.winner = (winner, self.ranks[winner].solved_idx)
				break
			if total + 12 * N > max_states:                                  # the reference's own guard, agents.py:236
				self.stop = STOP_BUDGET
				break
			if biggest + 12 * N > capacity:
				self.stop = STOP_CAPACITY
				break
			if not any(cands):
				self.stop = STOP_EMPTY
				break
			merged = sorted((cost + 0.0, r, pos) for r in range(W) for pos, (cost, _) in enumerate(cands[r]))[:N]
			n_pop = [sum(1 for _, r, _ in merged if r == q) for q in range(W)]
			# ---- k_shard_expand: pop, 12 children each, records bucketed by owner (stable) ------------------------
			send = [[[] for _ in range(W)] for _ in range(W)]                # send[src][dst] = records
			popped_now = []
			for r, rk in enumerate(self.ranks):
				popped = [heapq.heappop(rk.open)[1] for _ in range(n_pop[r])]
				popped_now.append(popped)
				if not popped:
					continue
				children = orc.expand12(np.array([rk.states[i] for i in popped]))
				for c, child in enumerate(children):
					p, a = popped[c // 12], c % 12
					send[r][owner_of(child, W)].append((child, p, rk.G[p] + 1, a, r))
			self.pops.append(popped_now)
			# ---- all-to-all: records of this iteration, offers of the previous one --------------------------------
			offers_in = [[o for src in range(W) for o in self.ranks[src].offers_out[dst]] for dst in range(W)]
			recs_in = [[rec for src in range(W) for rec in send[src][dst]] for dst in range(W)]
			for rk in self.ranks:
				rk.offers_out = [[] for _ in range(W)]
			# ---- insert + push, rank by rank (the ranks do not interact inside an iteration) ------------------------
			self.new_counts.append([self._insert_push(rk, offers_in[r], recs_in[r]) for r, rk in enumerate(self.ranks)])
			self.iterations += 1
		if self.stop == STOP_WON:
			self._walk()
		else:
			# no win: the offers of the last iteration are delivered and applied by one more exchange (rk_astar_shard_flush)
			for dst, rk in enumerate(self.ranks):
				self._apply_offers(rk, [o for src in range(W) for o in self.ranks[src].offers_out[dst]])
			for rk in self.ranks:
				rk.offers_out = [[] for _ in range(W)]
		return self.stop

	# -- one rank's half of an iteration ----------------------------------------------------------------------------
	@staticmethod
	def _apply_offers(rk: _Rank, offers):
		"""k_shard_offers_in: (parent idx, new G, child idx, child rank, rev action), all evaluated against G as it stands; the
		last hit per parent in arrival order is applied (agents.py:362-367 on the parent's owner)."""
		hits = {}
		for o, (p, g_new, child, crank, act) in enumerate(offers):
			if g_new < rk.G[p]:
				hits[p] = o
		for p, o in hits.items():
			_, g_new, child, crank, act = offers[o]
			rk.G[p], rk.parents[p], rk.prank[p], rk.pact[p] = g_new, child, crank, act

	def _insert_push(self, rk: _Rank, offers, records) -> int:
		self._apply_offers(rk, offers)
		# membership + first occurrence in arrival order (k_shard_lookup + k_append; agents.py:286-295)
		first_pos = {}
		for c, rec in enumerate(records):
			first_pos.setdefault(rec[0].tobytes(), c)
		first_unseen = [c for c, rec in enumerate(records) if first_pos[rec[0].tobytes()] == c and rec[0].tobytes() not in rk.index]
		first_seen = [c for c, rec in enumerate(records) if first_pos[rec[0].tobytes()] == c and rec[0].tobytes() in rk.index]
		seen_idx = [rk.index[records[c][0].tobytes()] for c in first_seen]
		# relaxation case 1, read half, against G before anything of this batch is written (agents.py:354)
		new_way = [records[c][2] < rk.G[s] for c, s in zip(first_seen, seen_idx)]
		# append (agents.py:299-313) + goal test of the new states (:321-323)
		new_idx = []
		for c in first_unseen:
			state, p, g, a, src = records[c]
			idx = rk.add(state, g, p, src, a)
			new_idx.append(idx)
			if orc.is_solved(state):
				rk.won, rk.solved_idx = True, idx
		# relaxation case 1, write half -- a rank that has just won returns before relaxing, like the reference
		if not rk.won:
			for c, s, hit in zip(first_seen, seen_idx, new_way):
				if hit:
					_, p, g, a, src = records[c]
					rk.G[s], rk.parents[s], rk.prank[s], rk.pact[s] = g, p, src, a
		# cost and push (agents.py:315-317, :369-383): float64 lambda * G plus the float32 heuristic
		if new_idx:
			H = -_values(self.net, np.array([rk.states[i] for i in new_idx]))
			for i, h in zip(new_idx, H):
				heapq.heappush(rk.open, (self.lambda_ * float(rk.G[i]) + float(h) + 0.0, i))
		# relaxation case 2 as offers to the parents' owners (k_shard_offers), G as it stands after case 1
		if not rk.won:
			for c, s in zip(first_seen, seen_idx):
				_, p, g, a, src = records[c]
				if rk.G[s] + 1 < g - 1:                                       # g - 1 = the parent's G when it expanded
					rk.offers_out[src].append((p, rk.G[s] + 1, s, rk.rank, a ^ 1))
		return len(new_idx)

	def _walk(self):
		"""Action queue from the solved node back to the root through (rank, idx) parent references (agents.py:244-251)."""
		r, i = self.winner
		guard = 0
		while not (r == self.root_owner and i == 1):
			rk = self.ranks[r]
			self.action_queue.appendleft(rk.pact[i])
			r, i = rk.prank[i], rk.parents[i]
			guard += 1
			assert guard < 100_000, "parent chain does not reach the root"

	# -- results in the layout the device engines export ---------------------------------------------------------------
	def arrays(self, rank: int):
		"""(states (n,20) int8, G (n,), parents (n,), parent ranks (n,), parent actions (n,)) of rows 1..n of `rank`."""
		rk = self.ranks[rank]
		n = len(rk)
		return (np.array(rk.states[1:], dtype=np.int8).reshape(n, 20), np.array(rk.G[1:], dtype=np.float64),
		        np.array(rk.parents[1:], dtype=np.int64), np.array(rk.prank[1:], dtype=np.int64), np.array(rk.pact[1:], dtype=np.int64))

	def open_queue(self, rank: int):
		return sorted(self.ranks[rank].open)

	@property
	def total_states(self) -> int:
		return sum(len(rk) for rk in self.ranks)
