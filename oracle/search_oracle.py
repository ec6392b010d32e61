"""
CPU oracle for the search loops -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (rules: see cube_oracle.py).

A plain-Python/NumPy restatement of the reference's batch weighted A* (librubiks/solving/agents.py:171-413) and MCTS
(agents.py:415-645), written for clarity, one explicit loop per step of the reference's vectorised code.  Built on
oracle/cube_oracle.py for the cube arithmetic.

Parity status: PINNED.  tests/test_search_oracle.py replays the traces that oracle/gen_golden.py captured from the
UNMODIFIED reference agents (driven by the exact-integer stub net below) and requires identical node arrays,
pop order and action queues.
"""
from __future__ import annotations

import heapq
from collections import deque

import numpy as np

from oracle import cube_oracle as orc


class StubNet:
	"""
	Exact heuristic used for traces: value = -(number of cubies not on their solved code), policy logits zero.
	Works on NumPy or torch one-hot batches; returns the same container type.  All numbers are small integers in
	float32, so CPU and GPU agree bit for bit.
	"""
	def __init__(self):
		self.solved_oh = orc.as_oh(orc.SOLVED)[0]
		self._dev = {}                       # device -> solved one-hot tensor (no copies inside a graph capture)

	def eval(self):
		return self

	def __call__(self, x, policy=True, value=True):
		import torch
		is_torch = isinstance(x, torch.Tensor)
		if is_torch:
			if x.device not in self._dev:
				self._dev[x.device] = torch.from_numpy(self.solved_oh).to(x.device)
			sol = self._dev[x.device]
			v = -(20 - (x.float() * sol).sum(dim=1, keepdim=True))
			p = torch.zeros(len(x), 12, device=x.device)
		else:
			v = -(20 - (x * self.solved_oh).sum(axis=1, keepdims=True)).astype(np.float32)
			p = np.zeros((len(x), 12), np.float32)
		out = ([p] if policy else []) + ([v] if value else [])
		return out if len(out) > 1 else out[0]


class PolicyStubNet(StubNet):
	"""
	StubNet with a non-uniform policy that is exact on any hardware (the twin of gen_golden.py's PolicyStubNet): every
	logit is 0 or -inf, 8 finite ones when corner 0 has an even code and 4 when odd, so the softmax is exactly 0, 1/8
	or 1/4 whether an implementation divides by the sum or multiplies by its reciprocal.
	"""
	def __init__(self):
		super().__init__()
		t = np.zeros((24, 12), np.float32)
		for c in range(24):
			banned = [(c + 3 * j) % 12 for j in range(4)] if c % 2 == 0 else [(c + j) % 12 for j in range(8)]
			t[c, banned] = -np.inf
		self.table = t
		self._tab_dev = {}

	def __call__(self, x, policy=True, value=True):
		import torch
		out = []
		if policy:
			if isinstance(x, torch.Tensor):
				if x.device not in self._tab_dev:
					self._tab_dev[x.device] = torch.from_numpy(self.table).to(x.device)
				out.append(self._tab_dev[x.device][x[:, :24].float().argmax(dim=1)])
			else:
				out.append(self.table[x[:, :24].argmax(axis=1)])
		if value:
			out.append(super().__call__(x, policy=False, value=True))
		return out if len(out) > 1 else out[0]


class NoisyStubNet(StubNet):
	"""
	StubNet plus exact integer "noise": value = -(cubies off their solved code) + ((one-hot . w) mod 7) - 3 with a fixed vector
	of small integers w, so every number is a small integer in float32 on any hardware.  The noise misleads the search, which
	then reaches the goal over roundabout paths -- what the graph completion and the breadth-first shortening of MCTS
	(agents.py:597-633) need to have something to shorten.  Test infrastructure only.
	"""
	def __init__(self, seed: int = 0):
		super().__init__()
		self.w = np.random.RandomState(seed).randint(0, 50, 480).astype(np.float32)
		self._w_dev = {}

	def __call__(self, x, policy=True, value=True):
		import torch
		out = []
		if policy:
			out.append(torch.zeros(len(x), 12, device=x.device) if isinstance(x, torch.Tensor) else np.zeros((len(x), 12), np.float32))
		if value:
			base = super().__call__(x, policy=False, value=True)
			if isinstance(x, torch.Tensor):
				if x.device not in self._w_dev:
					self._w_dev[x.device] = torch.from_numpy(self.w).to(x.device)
				noise = torch.remainder((x.float() * self._w_dev[x.device]).sum(dim=1, keepdim=True), 7.0) - 3.0
			else:
				noise = (np.mod((x * self.w).sum(axis=1, keepdims=True), 7.0) - 3.0).astype(np.float32)
			out.append(base + noise)
		return out if len(out) > 1 else out[0]


def adi_traindata_oracle(net, games: int, depth: int, alpha: float, method: str):
	"""
	`Train.ADI_traindata` (train.py:256-339) on the CPU oracle: scramble (:277), 12-child fan-out (:285), rewards
	(:294-296), value of every child + reward (:303, :313), targets (:315-324), loss weights (:329-332).
	Pinned by tests/test_search_oracle.py against tests/golden/adi_trace.npz (captured from the unmodified reference).
	Returns (one-hot states float32 (n,480), policy int64 (n,), value float32 (n,), loss weights float32 (n,)).
	"""
	states, oh_states = orc.sequence_scrambler(games, depth, method == "lapanfix")
	solved_scrambled = orc.multi_is_solved(states)
	sub = orc.expand12(states)
	solved_sub = orc.multi_is_solved(sub)
	rewards = np.where(solved_sub, 0.0 if method == "reward0" else 1.0, -1.0).astype(np.float32)
	values = np.asarray(net(orc.as_oh(sub), policy=False, value=True), dtype=np.float32).reshape(-1) + rewards
	values = values.reshape(-1, 12)
	policy = values.argmax(axis=1)
	value = values[np.arange(len(values)), policy].copy()
	if method == "lapanfix":
		value[solved_scrambled] = 0
	elif method == "schultzfix":
		value[np.arange(0, len(states), depth)] = 0
	w = np.tile(1 / np.arange(1, depth + 1), games)
	u = np.ones_like(w)
	lw = ((1 - alpha) * w / w.sum() + alpha * u / len(u)) * (w.sum() + len(u))
	return oh_states, policy.astype(np.int64), value.astype(np.float32), lw.astype(np.float32)


def _values(net, states: np.ndarray) -> np.ndarray:
	"""float32 value vector of the net for a batch of 20-byte states."""
	v = net(orc.as_oh(states), policy=False, value=True)
	return np.asarray(v, dtype=np.float32).reshape(-1)


# ----------------------------------------------------------------------------------------------------------------
class BFSOracle:
	"""agents.py:92-129: first-in-first-out search over the 12-move graph, bounded by the number of states seen (the reference's
	time limit is not restated: traces are captured with max_states alone)."""

	def __init__(self):
		self.states, self.action_queue = {}, deque()

	def __len__(self):
		return len(self.states)                                              # agents.py:128-129

	def search(self, start: np.ndarray, max_states: int) -> bool:
		self.states, self.action_queue = {}, deque()
		if orc.is_solved(start):
			return True
		self.states = {start.tobytes(): (None, None)}                        # agents.py:103
		queue = deque([start])
		while len(self) < max_states:                                        # agents.py:105
			state = queue.popleft()
			key = state.tobytes()
			for a in range(12):
				child = orc.rotate(state, a // 2, 1 - a % 2)
				ckey = child.tobytes()
				if ckey in self.states:
					continue
				if orc.is_solved(child):                                     # agents.py:113-118: walk the predecessors back to the start
					self.action_queue.appendleft(a)
					while self.states[key][0] is not None:
						self.action_queue.appendleft(self.states[key][1])
						key = self.states[key][0]
					return True
				self.states[ckey] = (key, a)
				queue.append(child)
		return False


# ----------------------------------------------------------------------------------------------------------------
class AStarOracle:
	"""agents.py:171-413.  Node arrays are 1-based like the reference's (index 0 unused)."""

	def __init__(self, net, lambda_: float, expansions: int):
		self.net, self.lambda_, self.expansions = net, lambda_, expansions

	def reset(self):
		self.index = {}                      # state bytes -> idx                     agents.py:201
		self.states = [None]                 # agents.py:202
		self.G = [np.nan]                    # float64 in the reference; whole numbers
		self.parents = [0]
		self.parent_actions = [0]
		self.open = []                       # heap of (cost, idx)                    agents.py:185
		self.pops = []                       # pop order of every iteration (for the traces)
		self.action_queue = deque()

	def __len__(self):
		return len(self.index)

	def _add(self, state, g, parent, action):
		idx = len(self.states)
		self.states.append(state.copy())
		self.G.append(float(g))
		self.parents.append(parent)
		self.parent_actions.append(action)
		self.index[state.tobytes()] = idx
		return idx

	def search(self, start: np.ndarray, max_states: int) -> bool:
		"""agents.py:220-252 with the deterministic budget (no time limit)."""
		self.reset()
		if orc.is_solved(start):
			return True
		self._add(np.asarray(start, np.int8), 0, 0, 0)
		heapq.heappush(self.open, (0, 1))
		while len(self) + self.expansions * 12 <= max_states:
			n_pop = min(len(self.open), self.expansions)
			popped = [heapq.heappop(self.open)[1] for _ in range(n_pop)]
			self.pops.append(np.array(popped, dtype=np.int64))
			if self.expand_batch(popped):
				i = self.index[orc.SOLVED.tobytes()]
				while i != 1:
					self.action_queue.appendleft(self.parent_actions[i])
					i = self.parents[i]
				return True
		return False

	def expand_batch(self, popped) -> bool:
		"""agents.py:254-331."""
		if not popped:
			return False
		parents_of = np.repeat(np.array(popped), 12)
		actions_of = np.tile(np.arange(12), len(popped))
		children = orc.expand12(np.array([self.states[i] for i in popped]))
		keys = [c.tobytes() for c in children]

		# first occurrence of every distinct state in batch order (np.unique(..., return_index=True), :291-295)
		first_pos = {}
		for pos, k in enumerate(keys):
			first_pos.setdefault(k, pos)
		first_unseen = [pos for pos, k in enumerate(keys) if first_pos[k] == pos and k not in self.index]
		first_seen = [pos for pos, k in enumerate(keys) if first_pos[k] == pos and k in self.index]
		seen_idx = [self.index[keys[pos]] for pos in first_seen]          # before the new ones are added

		# unseen: append in batch order (:299-313)
		new_idx = []
		for pos in first_unseen:
			p = int(parents_of[pos])
			new_idx.append(self._add(children[pos], self.G[p] + 1, p, int(actions_of[pos])))
		new_states = children[first_unseen]

		# cost and push (:315-317, :369-383): float64 lambda*G plus float32 heuristic
		if len(new_idx):
			H = -_values(self.net, new_states)
			cost = self.lambda_ * np.array([self.G[i] for i in new_idx], dtype=np.float64) + H
			for c, i in zip(cost, new_idx):
				heapq.heappush(self.open, (c, i))

		# goal test on the new states only (:321-323); the reference returns before relaxing
		if len(new_idx) and orc.multi_is_solved(new_states).any():
			return True

		# relax the seen ones (:333-367): two passes, each evaluated in full before it is applied
		rel_parent = [int(parents_of[pos]) for pos in first_seen]
		rel_action = [int(actions_of[pos]) for pos in first_seen]
		new_way = [self.G[p] + 1 < self.G[s] for s, p in zip(seen_idx, rel_parent)]
		rhs = [self.G[p] + 1 for p in rel_parent]
		for s, p, a, hit, g in zip(seen_idx, rel_parent, rel_action, new_way, rhs):
			if hit:
				self.G[s], self.parent_actions[s], self.parents[s] = g, a, p
		shortcut = [self.G[s] + 1 < self.G[p] for s, p in zip(seen_idx, rel_parent)]
		rhs = [self.G[s] + 1 for s in seen_idx]
		for s, p, a, hit, g in zip(seen_idx, rel_parent, rel_action, shortcut, rhs):
			if hit:                                   # repeated p: the last assignment stays, as in NumPy
				self.G[p], self.parent_actions[p], self.parents[p] = g, a ^ 1, s
		return False

	# arrays in the layout of the golden traces
	def arrays(self):
		n = len(self)
		return (np.array(self.states[1:n + 1], dtype=np.int8), np.array(self.G[1:n + 1]),
		        np.array(self.parents[2:n + 1], dtype=np.int64), np.array(self.parent_actions[2:n + 1], dtype=np.int64))


# ----------------------------------------------------------------------------------------------------------------
class MCTSOracle:
	"""agents.py:415-645.  One tree; arrays 1-based like the reference's."""

	def __init__(self, net, c: float, search_graph: bool, nu: float = 100.0):
		self.net, self.c, self.search_graph, self.nu = net, c, search_graph, nu

	def reset(self, cap: int):
		self.index = {}
		self.states = np.zeros((cap + 1, 20), np.int8)
		self.neighbors = np.zeros((cap + 1, 12), np.int64)
		self.leaves = np.ones(cap + 1, bool)
		self.P = np.zeros((cap + 1, 12))
		self.V = np.zeros(cap + 1)
		self.N = np.zeros((cap + 1, 12), np.int64)
		self.W = np.zeros((cap + 1, 12))
		self.L = np.zeros((cap + 1, 12))
		self.action_queue = deque()
		self.sims = 0

	def __len__(self):
		return len(self.index)

	def _policy_value(self, states):
		import torch
		p, v = self.net(orc.as_oh(states))
		p = torch.as_tensor(np.asarray(p)).float().softmax(dim=1).numpy()       # float32 softmax, as agents.py:552
		return p.astype(np.float64), np.asarray(v, dtype=np.float32).reshape(-1).astype(np.float64)

	def search(self, start: np.ndarray, max_states: int, max_sims: int = None) -> bool:
		"""agents.py:461-494 with the deterministic budget (`max_sims` additionally bounds the number of simulations)."""
		self.reset(max_states + 12)
		self.index[start.tobytes()] = 1
		self.states[1] = start
		if orc.is_solved(start):
			return True
		p, v = self._policy_value(start[None])
		self.P[1], self.V[1] = p[0], v[0]
		path, actions = [1], []
		while len(self) + 12 <= max_states and (max_sims is None or self.sims < max_sims):
			leaf, action = self.expand_leaf(path, actions)
			if leaf != -1:
				self.action_queue = deque(actions + [action])
				if self.search_graph:
					self.complete_graph()
					self.shorten(leaf)
				return True
			path, actions = self.find_leaf()
		self.action_queue = deque(actions)
		return False

	def expand_leaf(self, path, actions):
		"""agents.py:496-573."""
		self.sims += 1
		leaf = path[-1]
		children = orc.expand12(self.states[leaf][None])
		keys = [c.tobytes() for c in children]
		unseen = [a for a, k in enumerate(keys) if k not in self.index]
		for a in unseen:                                                     # new indices in action order (:523-529)
			idx = len(self.index) + 1
			self.index[keys[a]] = idx
			self.states[idx] = children[a]
		child_idx = np.array([self.index[k] for k in keys])
		self.neighbors[leaf] = child_idx                                     # :534
		self.neighbors[child_idx, np.arange(12) ^ 1] = leaf                  # :535
		self.leaves[leaf] = False
		solved = np.flatnonzero(orc.multi_is_solved(children))               # :540-543, over ALL children
		solve_leaf, solve_action = (int(child_idx[solved[0]]), int(solved[0])) if solved.size else (-1, -1)

		new_idx = child_idx[unseen]
		p, v = self._policy_value(children[unseen])                          # :548-552 (the reference assumes >= 1 new)
		self.P[new_idx], self.V[new_idx] = p, v
		best = v.max()
		self.W[leaf] = self.V[self.neighbors[leaf]]                          # :560
		self.W[new_idx] = v[:, None]                                         # :561
		for node, a in zip(path[:-1], actions):                              # :562 max-backup
			self.W[node, a] = max(self.W[node, a], best)
		if actions:                                                          # :567-570
			for node, a in set(zip(path[:-1], actions)):                     # fancy `+=` counts a repeated pair once
				self.N[node, a] += 1
			for node, a in zip(path[:-1], actions):
				self.L[node, a] = 0
			for node, a in zip(path[1:], actions):
				self.L[node, a ^ 1] = 0
		return solve_leaf, solve_action

	def find_leaf(self):
		"""agents.py:575-595."""
		cur, path, actions = 1, [1], []
		while not self.leaves[cur]:
			sqrt_n = np.sqrt(self.N[cur].sum())
			U = self.c * self.P[cur] * sqrt_n / (1 + self.N[cur])
			Q = self.W[cur] - self.L[cur]
			a = int((U + Q).argmax())
			self.L[cur, a] += self.nu
			cur = int(self.neighbors[cur, a])
			self.L[cur, a ^ 1] += self.nu
			path.append(cur)
			actions.append(a)
		return path, actions

	def complete_graph(self):
		"""agents.py:597-611."""
		n = len(self)
		leaf_idx = np.flatnonzero(self.leaves[:n + 1])[1:]
		if not len(leaf_idx):
			return
		children = orc.expand12(self.states[leaf_idx])
		child_idx = np.array([self.index.get(c.tobytes(), 0) for c in children])
		rep = np.repeat(leaf_idx, 12)
		acts = np.tile(np.arange(12), len(leaf_idx))
		self.neighbors[rep, acts] = child_idx
		self.neighbors[child_idx, acts ^ 1] = rep
		self.neighbors[0] = 0

	def shorten(self, solved_index: int):
		"""Breadth-first search inside the explored graph (agents.py:613-633)."""
		if solved_index == 1:
			return
		self.action_queue = deque()
		came_from = {1: (None, None)}
		q = deque([1])
		while q:
			v = q.popleft()
			for a, n in enumerate(self.neighbors[v]):
				n = int(n)
				if not n or n in came_from:
					continue
				if n == solved_index:
					self.action_queue.appendleft(a)
					while came_from[v][0] is not None:
						self.action_queue.appendleft(came_from[v][1])
						v = came_from[v][0]
					return
				came_from[n] = (v, a)
				q.append(n)
