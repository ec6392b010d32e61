"""
Randomised differential soak: the HIP path against the CPU oracle on cases drawn at random for a fixed time -- sizes around the
paced forms' thresholds and ragged tiles, pool offsets, both representations; A* and MCTS searches with random parameters, nets
(exact stub, misleading stub, non-uniform policy stub), pools that grow on the way, eager and hipGraph-replayed steps, single and
batched engines.  Everything is compared bit for bit.  A failing case is reported with the numbers that reproduce it and ends the
run with exit code 1.  The oracle is the checker here, as in tests/ (this script is a long-running test, not a benchmark of it).

    python tests/differential/soak.py --minutes 10 --seed 1 > profiles/r04_soak.json
"""
import argparse
import json
import os
import sys
import time
import traceback

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librubiks_amd import cube  # noqa: E402
from librubiks_amd.solving.agents import AStar, AStarBatch, MCTS, MCTSBatch  # noqa: E402
from oracle import c_oracle, cube_oracle as orc  # noqa: E402
from oracle.search_oracle import AStarOracle, MCTSOracle, NoisyStubNet, PolicyStubNet, StubNet  # noqa: E402

dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()


def walk(rng, n: int, depth: int) -> np.ndarray:
	s = orc.repeat_state(orc.SOLVED, n)
	for _ in range(depth):
		s = c_oracle.multi_rotate(s, rng.randint(0, 12, n).astype(np.uint8), threads=4)
	return s


def size(rng, hi: int) -> int:
	special = [1, 63, 64, 65, 255, 256, 257, 1023, 4097, 65_535, 65_536, 65_537, 131_071, 131_072, 196_607, 196_608, 196_609, 262_144, 262_145]
	if rng.rand() < 0.35:
		return int(min(hi, rng.choice(special) + rng.randint(0, 3)))
	return int(min(hi, max(1, np.exp(rng.uniform(0, np.log(hi))))))


def case_cube20(rng):
	n = size(rng, 600_000)
	pad = int(rng.randint(0, 5))
	depth = int(rng.randint(0, 4)) if rng.rand() < 0.3 else int(rng.randint(4, 25))
	base = walk(rng, n + pad, depth)
	k = min(n, 50)
	base[rng.randint(0, n + pad, k)] = orc.SOLVED
	near = orc.multi_rotate(orc.repeat_state(orc.SOLVED, 12), *orc.iter_actions())
	base[rng.randint(0, n + pad, k)] = near[rng.randint(0, 12, k)]
	host, view = base[pad:pad + n], dev(base)[pad:pad + n]
	what = dict(n=n, pad=pad, depth=depth)
	ch, fl = cube.device.expand12(view)
	ref_ch, ref_fl = c_oracle.expand12(host, threads=8)
	assert (ch.cpu().numpy() == ref_ch).all() and (fl.cpu().numpy() == ref_fl).all(), ("expand12", what)
	acts = rng.randint(0, 12, n).astype(np.uint8)
	moved = c_oracle.multi_rotate(host, acts, threads=8)
	assert (cube.device.multi_rotate(view, dev(acts)).cpu().numpy() == moved).all(), ("multi_rotate", what)
	stats = torch.tensor([0, np.iinfo(np.int64).max], dtype=torch.int64, device="cuda")
	out, flags = cube.device.multi_rotate_solved(view, dev(acts), stats=stats)
	ref_flags = orc.multi_is_solved(moved)
	assert (out.cpu().numpy() == moved).all() and (flags.cpu().numpy().astype(bool) == ref_flags).all(), ("multi_rotate_solved", what)
	st = stats.cpu().numpy()
	assert int(st[0]) == int(ref_flags.sum()) and (not ref_flags.any() or int(st[1]) == int(np.argmax(ref_flags))), ("stats", what, st.tolist())
	assert (cube.device.multi_is_solved(view).cpu().numpy().astype(bool) == orc.multi_is_solved(host)).all(), ("multi_is_solved", what)
	if n <= 160_000:
		for dt in (torch.float32, torch.bfloat16):
			got = cube.device.as_oh(view, dtype=dt)
			assert (got.float().cpu().numpy() == c_oracle.as_oh(host)).all(), ("as_oh", str(dt), what)
	# round 5: the walks (lane per game / wave per game with a scan over the moves) and the rollout's walks + goal tests + fan-out in one launch
	games, rows = int(rng.choice([1, 2, 5, 63, 700, 1024, 1025, 3000])), int(rng.choice([1, 2, 7, 8, 30, 31, 64, 65, 130]))
	with_solved = bool(rng.rand() < 0.5)
	if games * rows <= 200_000:
		acts = rng.randint(0, 12, (rows, games)).astype(np.uint8)
		cur = orc.repeat_state(orc.SOLVED, games)
		seq = [cur] if with_solved else []
		for d in range(rows - int(with_solved)):
			cur = c_oracle.multi_rotate(cur, acts[d], threads=4)
			seq.append(cur)
		want = np.stack(seq, axis=1).reshape(games * rows, 20)
		w2 = dict(what, games=games, rows=rows, with_solved=with_solved)
		assert (cube.device.apply_sequences(dev(acts), with_solved, False).cpu().numpy() == want).all(), ("apply_sequences", w2)
		assert (cube.device.apply_sequences(dev(acts), with_solved, True).cpu().numpy() == seq[-1]).all(), ("apply_sequences last", w2)
		st2, sfl, ch2, cfl = cube.device.rollout_fanout(dev(acts), with_solved)
		rch, rfl = c_oracle.expand12(want, threads=8)
		assert (st2.cpu().numpy() == want).all() and (sfl.cpu().numpy().astype(bool) == orc.multi_is_solved(want)).all(), ("rollout states", w2)
		assert (ch2.cpu().numpy() == rch).all() and (cfl.cpu().numpy() == rfl).all(), ("rollout children", w2)
	return what


def case_cube686(rng):
	n = size(rng, 70_000)
	cube.store_repr()
	cube.set_is2024(False)
	try:
		s = orc.solved_686()[None].repeat(n, axis=0)
		for _ in range(int(rng.randint(0, 8))):
			s = c_oracle.multi_rotate686(s, rng.randint(0, 12, n).astype(np.uint8))
		view = dev(s)
		what = dict(n=n)
		acts = rng.randint(0, 12, n).astype(np.uint8)
		moved = c_oracle.multi_rotate686(s, acts)
		assert (cube.device.multi_rotate(view, dev(acts)).cpu().numpy() == moved).all(), ("rotate686", what)
		assert (cube.device.multi_is_solved(view).cpu().numpy().astype(bool) == orc.multi_is_solved686(s)).all(), ("is_solved686", what)
		ch, fl = cube.device.expand12(view)
		ref = c_oracle.multi_rotate686(np.repeat(s, 12, axis=0), np.tile(np.arange(12, dtype=np.uint8), n))
		assert (ch.cpu().numpy().reshape(ref.shape) == ref).all() and (fl.cpu().numpy().astype(bool).reshape(-1) == orc.multi_is_solved686(ref)).all(), ("fanout686", what)
		if n <= 40_000:
			assert (cube.device.as_oh(view).cpu().numpy() == orc.as_oh686(s)).all(), ("as_oh686", what)
	finally:
		cube.restore_repr()
	return what


def _start(rng, depth):
	np.random.seed(int(rng.randint(0, 2 ** 31 - 1)))
	return orc.scramble(depth, True)[0]


def _net(rng, kinds):
	k = kinds[int(rng.randint(0, len(kinds)))]
	noise_seed = int(rng.randint(0, 5))                             # drawn ONCE: the oracle's net and the engine's must be the same net
	return k, {"stub": StubNet, "noisy": lambda: NoisyStubNet(noise_seed), "policy": PolicyStubNet}[k]


def case_astar(rng):
	kind, make = _net(rng, ["stub", "noisy"])
	what = dict(net=kind, depth=int(rng.randint(2, 22)), lam=float(rng.choice([0.0, 0.02, 0.1, 0.3, 0.7, 1.0])),
	            N=int(rng.choice([1, 3, 10, 50, 170, 600])), mode=str(rng.choice(["run-ahead", "hipgraph", "exact"])), grow=bool(rng.rand() < 0.4))
	what["budget"] = budget = int(rng.randint(12 * what["N"] + 50, 12 * what["N"] + 9_000))
	start = _start(rng, what["depth"])
	what["start"] = start.tolist()
	ref = AStarOracle(make(), what["lam"], what["N"])
	solved = ref.search(start, budget)
	cap = max(budget // 5, 12 * what["N"] + 2) if what["grow"] else budget
	agent = AStar(make(), what["lam"], what["N"], capacity=cap, use_hipgraph=what["mode"] == "hipgraph", exact_batch=True if what["mode"] == "exact" else None)
	assert agent.search(start, None, budget) == solved, ("solved", what)
	n = len(ref)
	rs, rG, rp, ra = ref.arrays()
	assert len(agent) == n and (agent.states[1:n + 1] == rs).all() and (agent.G[1:n + 1] == rG).all(), ("states/G", what)
	assert (agent.parents[2:n + 1] == rp).all() and (agent.parent_actions[2:n + 1] == ra).all(), ("parents", what)
	assert list(agent.action_queue) == list(ref.action_queue), ("action_queue", what)
	del what["start"]
	return what


def case_astar_batch(rng):
	kind, make = _net(rng, ["stub", "noisy"])
	S = int(rng.randint(2, 12))
	what = dict(net=kind, S=S, lam=float(rng.choice([0.05, 0.3, 1.0])), N=int(rng.choice([3, 20, 100])), graph=bool(rng.rand() < 0.5))
	budgets = rng.randint(12 * what["N"] + 20, 12 * what["N"] + 4_000, S)
	starts = np.array([_start(rng, int(rng.randint(2, 16))) for _ in range(S)])
	b = AStarBatch(make(), what["lam"], what["N"], S, capacity=int(budgets.max()))
	got = b.search(starts, max_states=budgets, use_graph=what["graph"])
	for i in range(S):
		ref = AStarOracle(make(), what["lam"], what["N"])
		assert ref.search(starts[i], int(budgets[i])) == bool(got[i]), ("solved", i, what)
		st, G, par, act = b.arrays_of(i)
		n = len(ref)
		rs, rG, rp, ra = ref.arrays()
		assert int(b.status[i, 2]) == n and (st[1:n + 1] == rs).all() and (G[1:n + 1] == rG).all() and (par[2:n + 1] == rp).all() and (act[2:n + 1] == ra).all(), ("arrays", i, what)
		assert list(b.action_queue_of(i)) == list(ref.action_queue), ("queue", i, what)
	return what


def _tree_equal(arrs, ref, what):
	n = len(ref)
	assert arrs["n"] == n, ("n", what)
	for k in ("states", "neighbors", "leaves", "N", "W", "L", "V", "P"):
		assert (arrs[k][1:n + 1] == getattr(ref, k)[1:n + 1]).all(), (k, what)


def case_mcts(rng):
	kind, make = _net(rng, ["stub", "policy"])
	what = dict(net=kind, depth=int(rng.randint(1, 18)), c=float(rng.choice([0.1, 0.6, 1.5, 5.0, 50.0])), graph=bool(rng.rand() < 0.5),
	            search_graph=bool(rng.rand() < 0.6), grow=bool(rng.rand() < 0.4), priors=str(rng.choice(["reference", "kernel", "torch"])))
	what["budget"] = budget = int(rng.randint(30, 6_000))
	start = _start(rng, what["depth"])
	ref = MCTSOracle(make(), what["c"], what["search_graph"])
	solved = ref.search(start, budget)
	tree = MCTS(make(), what["c"], what["search_graph"], capacity=max(budget // 5, 13) if what["grow"] else budget, use_hipgraph=what["graph"], priors=what["priors"])
	assert tree.search(start, None, budget) == solved, ("solved", what)
	_tree_equal(tree._export(), ref, what)
	assert list(tree.action_queue) == list(ref.action_queue), ("queue", what)
	return what


def case_mcts_batch(rng):
	kind, make = _net(rng, ["stub", "policy"])
	T = int(rng.randint(2, 10))
	what = dict(net=kind, T=T, c=float(rng.choice([0.6, 2.0, 50.0])), graph=bool(rng.rand() < 0.5), search_graph=bool(rng.rand() < 0.6), grow=bool(rng.rand() < 0.4))
	budgets = rng.randint(30, 4_000, T)
	starts = np.array([_start(rng, int(rng.randint(1, 14))) for _ in range(T)])
	cap = int(budgets.max())
	b = MCTSBatch(make(), what["c"], T, capacity=max(cap // 4, 13) if what["grow"] else cap, max_capacity=cap, search_graph=what["search_graph"])
	got = b.search(starts, max_states=budgets, use_graph=what["graph"], poll=int(rng.choice([1, 7, 16])))
	for i in range(T):
		ref = MCTSOracle(make(), what["c"], what["search_graph"])
		assert ref.search(starts[i], int(budgets[i])) == bool(got[i]), ("solved", i, what)
		_tree_equal(b.tree_arrays(i), ref, (i, what))
		assert list(b.action_queue_of(i)) == list(ref.action_queue) and int(b.status[i, 3]) == ref.sims, ("queue/sims", i, what)
	return what


def case_sharded(rng):
	"""The hash-sharded A* at a random world size, all ranks simulated in one process, against the protocol oracle: every
	iteration's pops and new counts of every rank, every shard, every open queue (tests/test_sharded_gpu.py::_simulate_ranks)."""
	from oracle.sharded_oracle import ShardedAStarOracle
	from tests.test_sharded_gpu import _simulate_ranks
	kind, make = _net(rng, ["stub", "noisy"])
	what = dict(net=kind, world=int(rng.randint(2, 9)), depth=int(rng.randint(3, 18)), lam=float(rng.choice([0.02, 0.1, 0.5, 1.0])), N=int(rng.choice([4, 30, 150, 500])))
	what["budget"] = budget = int(rng.randint(12 * what["N"] + 100, 12 * what["N"] + 12_000))
	start = _start(rng, what["depth"])
	o = ShardedAStarOracle(make(), what["lam"], what["N"], what["world"])
	o.search(start, budget)
	stop, queue, shards, total, iters = _simulate_ranks(what["world"], start, what["lam"], what["N"], budget, capacity=budget, oracle=o, net=make())
	assert total == o.total_states, ("total", what)
	return what


def case_host_surface(rng):
	"""The drop-in NumPy surface (host pointers in, fresh arrays out) and the scramblers' seed parity."""
	n = size(rng, 40_000)
	s = walk(rng, n, int(rng.randint(0, 20)))
	what = dict(n=n)
	faces, dirs = rng.randint(0, 6, n), rng.randint(0, 2, n)
	keep = s.copy()
	assert (cube.multi_rotate(s, faces, dirs) == orc.multi_rotate(s, faces, dirs)).all() and (s == keep).all(), ("multi_rotate", what)
	assert (cube.multi_rotate(s, faces.astype(np.uint8), dirs.astype(np.uint8)) == orc.multi_rotate(s, faces, dirs)).all(), ("multi_rotate u8", what)
	assert (cube.multi_is_solved(s) == orc.multi_is_solved(s)).all(), ("multi_is_solved", what)
	m = min(n, 3000)
	ch = cube.multi_rotate(np.repeat(s[:m], 12, axis=0), *cube.iter_actions(m))
	assert (ch == orc.expand12(s[:m])).all(), ("fan-out idiom", what)
	assert (cube.as_oh(s[:m]).cpu().numpy() == orc.as_oh(s[:m])).all(), ("as_oh", what)
	i = int(rng.randint(0, n))
	f, d = int(rng.randint(0, 6)), int(rng.randint(0, 2))
	assert (cube.rotate(s[i], f, d) == orc.rotate(s[i], f, d)).all() and cube.is_solved(s[i]) == orc.is_solved(s[i]), ("rotate", what)
	seed, depth, games, ws = int(rng.randint(0, 2 ** 31 - 1)), int(rng.randint(1, 40)), int(rng.randint(1, 60)), bool(rng.rand() < 0.5)
	what.update(seed=seed, depth=depth, games=games, with_solved=ws)
	np.random.seed(seed)
	a = cube.scramble(depth, True)
	b_states, b_oh = cube.sequence_scrambler(games, depth, ws)
	np.random.seed(seed)
	ra = orc.scramble(depth, True)
	rb_states, rb_oh = orc.sequence_scrambler(games, depth, ws)
	assert all((x == y).all() for x, y in zip(a, ra)), ("scramble", what)
	assert (b_states == rb_states).all() and (b_oh.cpu().numpy() == rb_oh).all(), ("sequence_scrambler", what)
	return what


_LIVE = {}


def case_reuse(rng):
	"""Long-lived agents in hipGraph mode, searched again and again: the captured step is kept from search to search and must be
	captured anew exactly when something it holds changes -- another net, another lambda, a pool that grew -- never go stale."""
	what = {}
	a = _LIVE.get("astar")
	if a is None or rng.rand() < 0.05:
		a = _LIVE["astar"] = dict(agent=AStar(StubNet(), 0.3, int(rng.choice([5, 40, 150])), capacity=6_000, use_hipgraph=True), net="stub", seed=0)
	if rng.rand() < 0.3:                                             # swap the net (or only its noise)
		a["net"], a["seed"] = str(rng.choice(["stub", "noisy"])), int(rng.randint(0, 3))
		a["agent"].net = StubNet() if a["net"] == "stub" else NoisyStubNet(a["seed"])
	if rng.rand() < 0.3:
		a["agent"].lambda_ = float(rng.choice([0.02, 0.3, 1.0]))
	agent = a["agent"]
	budget = int(rng.randint(12 * agent.expansions + 50, 30_000))  # beyond 6 000: the pool grows in place (and stays grown)
	start = _start(rng, int(rng.randint(2, 20)))
	what["astar"] = dict(net=a["net"], noise=a["seed"], lam=agent.lambda_, N=agent.expansions, budget=budget, captures_before=agent.captures)
	ref = AStarOracle(StubNet() if a["net"] == "stub" else NoisyStubNet(a["seed"]), agent.lambda_, agent.expansions)
	assert agent.search(start, None, budget) == ref.search(start, budget), ("astar solved", what)
	n = len(ref)
	rs, rG, rp, ra = ref.arrays()
	assert len(agent) == n and (agent.states[1:n + 1] == rs).all() and (agent.G[1:n + 1] == rG).all() and (agent.parents[2:n + 1] == rp).all() \
	       and (agent.parent_actions[2:n + 1] == ra).all() and list(agent.action_queue) == list(ref.action_queue), ("astar arrays", what)
	m = _LIVE.get("mcts")
	if m is None or rng.rand() < 0.05:
		tree = MCTS(StubNet(), float(rng.choice([0.6, 2.0, 50.0])), bool(rng.rand() < 0.5), capacity=1_500, use_hipgraph=True)
		tree.max_capacity = 12_000
		m = _LIVE["mcts"] = dict(agent=tree, net="stub")
	if rng.rand() < 0.3:
		m["net"] = str(rng.choice(["stub", "policy"]))
		m["agent"].net = StubNet() if m["net"] == "stub" else PolicyStubNet()
	tree = m["agent"]
	budget = int(rng.randint(30, 5_000))
	start = _start(rng, int(rng.randint(1, 16)))
	what["mcts"] = dict(net=m["net"], c=tree.c, search_graph=tree.search_graph, budget=budget)
	ref = MCTSOracle(StubNet() if m["net"] == "stub" else PolicyStubNet(), tree.c, tree.search_graph)
	assert tree.search(start, None, budget) == ref.search(start, budget), ("mcts solved", what)
	_tree_equal(tree._export(), ref, what)
	assert list(tree.action_queue) == list(ref.action_queue), ("mcts queue", what)
	return what


CASES = dict(cube20=case_cube20, sharded=case_sharded, host_surface=case_host_surface, reuse=case_reuse, cube686=case_cube686, astar=case_astar, astar_batch=case_astar_batch, mcts=case_mcts, mcts_batch=case_mcts_batch)

if __name__ == "__main__":
	ap = argparse.ArgumentParser()
	ap.add_argument("--minutes", type=float, default=10.0)
	ap.add_argument("--seed", type=int, default=1)
	ap.add_argument("--only", default="")
	args = ap.parse_args()
	rng = np.random.RandomState(args.seed)
	kinds = [k for k in CASES if not args.only or k in args.only.split(",")]
	counts, seconds = {k: 0 for k in kinds}, {k: 0.0 for k in kinds}
	t_end = time.time() + 60 * args.minutes
	last_print, failure, i = time.time(), None, 0
	while time.time() < t_end:
		kind = kinds[i % len(kinds)]
		i += 1
		state = rng.get_state()[1][:4].tolist()
		t0 = time.time()
		try:
			CASES[kind](rng)
		except Exception as e:                                       # an assertion (a difference), or an engine error: both end the run
			failure = dict(kind=kind, case_index=i - 1, rng_head=state, error=f"{type(e).__name__}: {e}"[:2000], trace=traceback.format_exc()[-1500:])
			break
		counts[kind] += 1
		seconds[kind] += time.time() - t0
		if time.time() - last_print > 45:
			print(f"[soak] {sum(counts.values())} cases so far: {counts}", file=sys.stderr, flush=True)
			last_print = time.time()
	print(json.dumps({"bench": "soak", "seed": args.seed, "minutes": args.minutes, "cases": counts, "seconds_by_kind": {k: round(v, 1) for k, v in seconds.items()},
	                  "total_cases": sum(counts.values()), "failure": failure,
	                  "what": "engine vs CPU oracle, bit for bit, on randomly drawn cases (sizes around the paced thresholds and ragged tiles, both representations; A*/MCTS single and batched, "
	                          "stub / misleading / policy nets, pools growing on the way, eager and hipGraph)"}))
	sys.exit(1 if failure else 0)
