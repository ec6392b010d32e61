"""
Pins oracle/sharded_oracle.py (the CPU restatement of the hash-sharded A* protocol) on the CPU:
  * at world = 1 it must BE the reference's batch weighted A* -- equal to AStarOracle (itself pinned to the unmodified reference's
    traces, tests/test_search_oracle.py) and to those traces directly: states, G, parents, actions, every iteration's pops;
  * its owner function equals the library's (rk_shard_owner is host code: no GPU needed);
  * at world = 2 / 3 / 8: every state on its owner exactly once, every parent link -- also across ranks -- a real move of the
    cube, G never below the walked distance bound, budget guard as the reference's.
The device engines are compared with this oracle in tests/test_sharded_gpu.py.
"""
import numpy as np
import pytest

from librubiks_amd import _ffi
from oracle import cube_oracle as orc
from oracle.search_oracle import AStarOracle, NoisyStubNet, StubNet
from oracle.sharded_oracle import STOP_BUDGET, STOP_CAPACITY, STOP_WON, ShardedAStarOracle, owner_of

NOISY_CASES = [(11, 14, 0.05, 50, 60_000), (12, 16, 0.02, 200, 40_000)]
CASES = [(7, 6, 0.5, 10, 30_000), (19, 7, 0.1, 300, 60_000), (402, 6, 1.0, 50, 30_000), (405, 8, 0.5, 30, 40_000), (104, 6, 0.05, 1000, 40_000)]


@pytest.mark.parametrize("tag", ["a", "b", "d", "e", "f"])
def test_world1_is_the_reference(golden, tag):
	"""e, f: traces of the unmodified reference with the misleading NoisyStubNet, in which relaxation cases 1 and 2 really happen:
	the protocol's offers (case 2 deferred to the next exchange, last hit per parent wins) must give the reference's arrays."""
	t = golden["astar_trace"]
	_, _, expansions, max_states = (int(x) for x in t[f"{tag}_params"])
	o = ShardedAStarOracle(NoisyStubNet() if tag in ("e", "f") else StubNet(), float(t[f"{tag}_lambda"]), expansions, 1)
	stop = o.search(t[f"{tag}_start"], max_states)
	assert (stop == STOP_WON) == bool(t[f"{tag}_solved"])
	states, G, parents, prank, pact = o.arrays(0)
	assert (states == t[f"{tag}_states"]).all() and (G == t[f"{tag}_G"]).all()
	assert (parents[1:] == t[f"{tag}_parents"]).all() and (pact[1:] == t[f"{tag}_parent_actions"]).all() and not prank.any()
	assert [len(p[0]) for p in o.pops] == t[f"{tag}_pop_lens"].tolist()
	assert (np.concatenate([p[0] for p in o.pops]) == t[f"{tag}_pops"]).all()
	assert list(o.action_queue) == t[f"{tag}_action_queue"].tolist()


@pytest.mark.parametrize("seed,depth,lam,n,budget", CASES)
def test_world1_equals_the_single_queue_oracle(seed, depth, lam, n, budget):
	np.random.seed(seed)
	start, _, _ = orc.scramble(depth, True)
	ref = AStarOracle(StubNet(), lam, n)
	solved = ref.search(start, budget)
	o = ShardedAStarOracle(StubNet(), lam, n, 1)
	stop = o.search(start, budget)
	assert (stop == STOP_WON) == solved and (solved or stop == STOP_BUDGET)
	rs, rG, rp, ra = ref.arrays()
	states, G, parents, prank, pact = o.arrays(0)
	assert (states == rs).all() and (G == rG).all() and (parents[1:] == rp).all() and (pact[1:] == ra).all()
	assert len(o.pops) == len(ref.pops) and all((np.array(a[0]) == b).all() for a, b in zip(o.pops, ref.pops))
	assert list(o.action_queue) == list(ref.action_queue)
	assert sorted(ref.open) == o.open_queue(0)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_protocol_with_a_misleading_heuristic(world):
	"""NoisyStubNet: dozens of cross-rank shortcut offers, several on one parent in one exchange.  The shards stay consistent."""
	for seed, depth, lam, n, budget in NOISY_CASES:
		np.random.seed(seed)
		start, _, _ = orc.scramble(depth, True)
		o = ShardedAStarOracle(NoisyStubNet(), lam, n, world)
		o.search(start, budget)
		assert len(check_shards(o, start)) == o.total_states


def test_owner_function_is_the_librarys():
	lib = _ffi.lib()
	np.random.seed(0)
	s = orc.repeat_state(orc.SOLVED, 600)
	for _ in range(12):
		s = orc.multi_rotate(s, np.random.randint(0, 6, len(s)), np.random.randint(0, 2, len(s)))
	for world in (1, 2, 3, 5, 8):
		assert [owner_of(x, world) for x in s] == [lib.rk_shard_owner(np.ascontiguousarray(x).ctypes.data, world) for x in s]


def check_shards(o: ShardedAStarOracle, start: np.ndarray, arrays=None):
	"""Properties any correct sharded search has; `arrays(rank)` defaults to the oracle's own (the GPU test passes the engines')."""
	arrays = arrays or o.arrays
	W = o.world
	shards = [arrays(r) for r in range(W)]
	seen = {}
	for r, (states, G, parents, prank, pact) in enumerate(shards):
		for i, x in enumerate(states):
			assert owner_of(x, W) == r
			assert x.tobytes() not in seen
			seen[x.tobytes()] = (r, i + 1)
	assert seen[np.asarray(start, np.int8).tobytes()] == (o.root_owner, 1)
	# every parent link is a real move, also across ranks; following the links reaches the root with no more steps than G, and
	# G never undercuts the true distance bound (a link's parent has G >= child's G - 1 is NOT guaranteed after shortcuts, but
	# G itself always counts a real path: G[child] >= walked length is what relaxation keeps)
	rng = np.random.RandomState(1)
	for r, (states, G, parents, prank, pact) in enumerate(shards):
		n = len(states)
		for i in ([0, n - 1] + list(rng.randint(0, n, 40)) if n else []):
			cr, ci, steps = r, i + 1, 0
			while not (cr == o.root_owner and ci == 1):
				st, Gc, par, pr, pa = shards[cr]
				pstates = shards[int(pr[ci - 1])][0]
				parent_state = pstates[int(par[ci - 1]) - 1]
				a = int(pa[ci - 1])
				assert (orc.rotate(parent_state, a // 2, 1 - a % 2) == st[ci - 1]).all()      # the link is a move of the cube
				cr, ci = int(pr[ci - 1]), int(par[ci - 1])
				steps += 1
				assert steps <= 200
			assert steps <= G[i] or steps == 0                                  # the chain is never longer than the recorded cost
	return seen


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_protocol_properties(world):
	for seed, depth, lam, n, budget in CASES[:4]:
		np.random.seed(seed)
		start, _, _ = orc.scramble(depth, True)
		o = ShardedAStarOracle(StubNet(), lam, n, world)
		stop = o.search(start, budget)
		seen = check_shards(o, start)
		assert len(seen) == o.total_states <= budget
		assert all(sum(len(p) for p in it) <= n for it in o.pops) and all(sum(c) <= 12 * n for c in o.new_counts)
		if stop == STOP_WON:
			s = start
			for a in o.action_queue:
				s = orc.rotate(s, a // 2, 1 - a % 2)
			assert orc.is_solved(s)
			ref = AStarOracle(StubNet(), lam, n)
			if ref.search(start, budget):
				assert len(o.action_queue) <= len(ref.action_queue) + 2
		else:
			assert stop == STOP_BUDGET and o.total_states + 12 * n > budget
	# a pool smaller than the budget stops every rank together
	np.random.seed(42)
	start, _, _ = orc.scramble(14, True)
	o = ShardedAStarOracle(StubNet(), 0.2, 100, world)
	assert o.search(start, 10_000_000, capacity=9_000) == STOP_CAPACITY
	assert max(len(rk) for rk in o.ranks) + 1200 > 9_000 and all(len(rk) <= 9_000 for rk in o.ranks)
