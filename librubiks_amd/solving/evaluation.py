"""
The caller on the far side of the search agents: the reference's `Evaluator` (librubiks/solving/evaluation.py:15-133) --
`n_games` scrambles at each of `scrambling_depths`, one `agent.search` per scramble, and the three matrices every plot and
log line of the reference is made from: `res` (moves of the solution found, -1 if none), `states` (len(agent) after the
game) and `times` (seconds).  Plots are not mirrored (UI, out of scope).

Two ways to play the games:

  * one after the other (evaluation.py:68-84 to the letter): any agent, any limit;
  * `batched`: all games of an `AStar` or `MCTS` agent advance in lock-step on the device (`AStarBatch`, `MCTSBatch`), the
    form the engines were built for -- the reference plays 100 games per depth one at a time.  The scrambles are drawn from
    the global NumPy generator in exactly the order the sequential loop draws them (A* and MCTS draw nothing while they
    search, so the interleaving of scrambling and searching does not matter), every search is the sequential search bit for
    bit, and so `res` and `states` are equal entry by entry for games bounded by `max_states`
    (tests/test_evaluation_gpu.py, against the unmodified reference's own Evaluator through tests/golden/evaluator_trace.npz).
    `times[i, j]` is then the time from the start of the game's batch to the poll that saw it finished -- games share the
    device, a per-game stopwatch does not exist.
"""
import time

import numpy as np

from librubiks_amd import cube
from librubiks_amd.solving import agents


class NullLogger:
	"""utils/logger.py:65-76: the logger that logs nothing (a real one is any callable with .section and .verbose)."""
	def __call__(self, *a, **k):
		pass

	def section(self, *a, **k):
		pass

	def verbose(self, *a, **k):
		pass


def bernoulli_error(p: float, n: int, alpha: float = 0.05) -> float:
	"""Half width of the normal-approximation confidence interval of a share (utils/__init__.py:24-30, used at evaluation.py:108)."""
	z = {0.1: 1.6448536269514722, 0.05: 1.959963984540054, 0.01: 2.5758293035489004}.get(alpha)
	if z is None:
		from scipy.stats import norm
		z = norm.ppf(1 - alpha / 2)
	return float(z * np.sqrt(p * (1 - p) / n)) if n else float("nan")


class Evaluator:
	def __init__(self, n_games, scrambling_depths, max_time=None, max_states=None, logger=None, batch_games: int = 64):
		self.n_games = n_games
		self.max_time = max_time
		self.max_states = max_states
		self.log = logger or NullLogger()
		self.batch_games = int(batch_games)          # games advanced together in batched mode (memory: a pool of max_states per game)
		# evaluation.py:30: an empty range means "deep": every game's depth is drawn uniformly from [100, 999]
		self.scrambling_depths = np.array(scrambling_depths) if scrambling_depths != range(0) else np.array([0])
		self.last_mode = None                        # "sequential" or "batched": how the last eval() played its games

	def _isdeep(self):
		return self.scrambling_depths.size == 1 and self.scrambling_depths[0] == 0

	def approximate_time(self):
		return self.max_time * self.n_games * len(self.scrambling_depths)

	def _eval_game(self, agent, depth: int):
		"""evaluation.py:46-54"""
		turns_to_complete = -1
		state, _, _ = cube.scramble(depth, True)
		t0 = time.perf_counter()
		solution_found = agent.search(state, self.max_time, self.max_states)
		dt = time.perf_counter() - t0
		if solution_found:
			turns_to_complete = len(agent.action_queue)
		return turns_to_complete, dt

	@staticmethod
	def can_batch(agent) -> bool:
		return type(agent) in (agents.AStar, agents.MCTS)

	def eval(self, agent, batched: bool = None):
		"""
		Returns (res, states, times), each len(scrambling_depths) x n_games (evaluation.py:56-96).
		batched: None = lock-step on the device when the agent can (AStar, MCTS) and the games are bounded by max_states alone.
		"""
		assert self.max_time or self.max_states
		if batched is None:
			batched = self.can_batch(agent) and self.max_states is not None and not self.max_time
		if batched and not self.can_batch(agent):
			raise TypeError(f"{agent} has no batched engine; play its games one after the other")
		if batched and self.max_states is None:
			raise ValueError("batched games need max_states: it sizes every game's node pool (max_time, if given, then limits the whole batch)")
		self.log.section(f"Evaluation of {agent}")
		D, G = len(self.scrambling_depths), self.n_games
		self.last_mode = "batched" if batched else "sequential"
		if batched:
			res, states, times = self._eval_batched(agent)
		else:
			res, states, times = [], [], []
			for d in self.scrambling_depths:
				for _ in range(G):
					if self._isdeep():
						d = np.random.randint(100, 1000)
					r, dt = self._eval_game(agent, d)
					res.append(r)
					states.append(len(agent))
					times.append(dt)
		res = np.reshape(res, (D, G))
		states = np.reshape(states, (D, G))
		times = np.reshape(times, (D, G))
		self.log("Evaluation results")
		for i, d in enumerate(self.scrambling_depths):
			self.log_this_depth(res[i], states[i], times[i], d)
		return res, states, times

	def _eval_batched(self, agent):
		# the scrambles, drawn as the sequential loop draws them: per game the deep depth (if any), then faces, then directions
		starts = []
		for d in self.scrambling_depths:
			for _ in range(self.n_games):
				if self._isdeep():
					d = np.random.randint(100, 1000)
				starts.append(cube.scramble(d, True)[0])
		starts = np.array(starts, dtype=np.int8).reshape(-1, 20)
		total = len(starts)
		res, states, times = np.full(total, -1, np.int64), np.zeros(total, np.int64), np.zeros(total)
		b, b_games = None, 0
		for lo in range(0, total, self.batch_games):
			hi = min(total, lo + self.batch_games)
			n = hi - lo
			if n != b_games:                                          # (the last group may be smaller)
				b = None                                              # at most one engine alive: its pools are n x max_states nodes
				b, b_games = self._batch_agent(agent, n), n
			t0 = time.perf_counter()
			solved, seen = self._run_batch(b, starts[lo:hi], t0)
			for i in range(n):
				if solved[i]:
					res[lo + i] = len(b.action_queue_of(i))
			states[lo:hi] = b.status[:, 2]
			times[lo:hi] = seen
		return res, states, times

	def _batch_agent(self, agent, n: int):
		cap = int(self.max_states)
		if isinstance(agent, agents.AStar):
			b = agents.AStarBatch(agent.net, agent.lambda_, agent.expansions, n, capacity=cap, fused_first_layer=agent._fused_mode)
		else:
			b = agents.MCTSBatch(agent.net, agent.c, n, capacity=max(cap, 13), nu=agent.nu, priors=agent.priors,
			                     search_graph=agent.search_graph)
		return b

	def _run_batch(self, b, starts: np.ndarray, t0: float):
		"""Runs one batch; returns (solved, seconds until each game was seen finished)."""
		seen = np.zeros(len(starts))
		def on_poll(st):
			fresh = (st[:, 0] != 0) & (seen == 0)
			seen[fresh] = time.perf_counter() - t0
		b.on_poll = on_poll
		try:
			# max_time, if given, is every game's limit on the batch's one clock (the games share the device: each gets less of it
			# than a game played alone would -- which is why eval() only chooses this form by itself for games without a time limit)
			if isinstance(b, agents.AStarBatch):
				solved = b.search(starts, self.max_time, self.max_states)
			else:
				solved = b.search(starts, self.max_time, max_states=self.max_states, use_graph=b.priors != "reference")
		finally:
			b.on_poll = None
		seen[seen == 0] = time.perf_counter() - t0
		return solved, seen

	def log_this_depth(self, res: np.ndarray, states: np.ndarray, times: np.ndarray, depth: int):
		"""evaluation.py:98-133: share solved with its 95 % interval, moves of the solutions, states per game and per second."""
		share = np.count_nonzero(res != -1) * 100 / len(res)
		won = res[res != -1]
		self.log(f"Scrambling depth {depth if depth else 'deep'}", with_timestamp=False)
		self.log(f"\tShare completed: {share:.2f} % +/- {100 * bernoulli_error(share / 100, len(res), 0.05):.2f} % (approx. 95 % CI)", with_timestamp=False)
		if won.size:
			self.log(f"\tTurns to win: {won.mean():.2f} +/- {won.std():.1f} (std.), Median: {np.median(won):.0f}", with_timestamp=False)
		safe = times != 0
		sps = states[safe] / times[safe]
		if sps.size:
			self.log(f"\tStates seen: Pr. game: {states.mean():.2f} +/- {states.std():.0f} (std.), "
			         f"Pr. sec.: {sps.mean():.2f} +/- {sps.std():.0f} (std.)", with_timestamp=False)
		self.log(f"\tTime:  {times.mean():.2f} +/- {times.std():.2f} (std.)", with_timestamp=False)
