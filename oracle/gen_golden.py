"""
Golden-vector capture -- TEST INFRASTRUCTURE.  Runs ONLY in the build container.

Imports the real reference from /root/reference (never copied, never shipped), drives it on
seeded inputs and writes small input/output fixtures into tests/golden/.  The fixtures are data:
arrays, hashes and strings.  The GPU box has no /root/reference; tests there read the fixtures.

    PYTHONDONTWRITEBYTECODE=1 python3 oracle/gen_golden.py [--skip-1m]

Fixtures
  cube_tables.npz   delta table of the reference, the frontend's maps.json arrays, solved states
  cube_kat.npz      known answers: scrambles K1/K2, random-walk fan-outs K3 (arrays at n=256, SHA-256
                    at n=10 000 and n=1 000 000), sequence_scrambler K5, one-hot, 6x8x6 moves
  cube_text.json    stringify() texts, iter_actions literals, hashes
  astar_trace.npz   unmodified reference AStar driven by an exact-integer stub net
  mcts_trace.npz    unmodified reference MCTS driven by the same stub (one case with a non-uniform exact policy)
  adi_trace.npz     unmodified reference Train.ADI_traindata (train.py:256-339), stub net, all four reward methods
  evaluator_trace.npz  unmodified reference Evaluator.eval (solving/evaluation.py:56-96) over the unmodified agents and the stubs:
                    the `res` and `states` matrices (times are wall clock: not data)
"""
import argparse
import hashlib
import json
import os
import sys
import warnings

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

sys.dont_write_bytecode = True
sys.path.insert(0, REF)
warnings.filterwarnings("ignore", category=DeprecationWarning)

from librubiks import cube  # noqa: E402
from librubiks.cube.cube import _Cube2024  # noqa: E402
from librubiks.solving import agents  # noqa: E402


def sha(a: np.ndarray) -> str:
	return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def random_walk(n: int, depth: int):
	"""K3 recipe: n solved states, `depth` rounds of per-row random moves."""
	s = cube.repeat_state(cube.get_solved(), n)
	for _ in range(depth):
		f = np.random.randint(0, 6, n)
		d = np.random.randint(0, 2, n)
		s = cube.multi_rotate(s, f, d)
	return s


def fan_out(states: np.ndarray) -> np.ndarray:
	"""The reference's own 12-child idiom (agents.py:277-281)."""
	return cube.multi_rotate(np.repeat(states, 12, axis=0), *cube.iter_actions(len(states)))


class StubNet:
	"""
	Exact small-integer heuristic: value = -(number of cubies whose code differs from solved),
	policy logits all zero.  Integer-valued float32 everywhere, so search control flow is identical
	on any hardware.
	"""
	def __init__(self):
		self.solved_oh = cube.as_oh(cube.get_solved())

	def eval(self):
		return self

	def __call__(self, x, policy=True, value=True):
		out = []
		if policy:
			out.append(torch.zeros(len(x), 12))
		if value:
			out.append(-(20 - (x * self.solved_oh).sum(dim=1, keepdim=True)))
		return out if len(out) > 1 else out[0]


class NoisyStubNet(StubNet):
	"""The twin of oracle.search_oracle.NoisyStubNet for the reference's torch tensors: the stub's value plus exact integer noise
	((one-hot . w) mod 7) - 3 with w = RandomState(seed).randint(0, 50, 480).  It misleads the search, so that states are rediscovered
	over shorter ways: BOTH relaxation cases of agents.py:333-367 really happen (with the plain stub they almost never do)."""
	def __init__(self, seed: int = 0):
		super().__init__()
		self.w = torch.from_numpy(np.random.RandomState(seed).randint(0, 50, 480).astype(np.float32))

	def __call__(self, x, policy=True, value=True):
		out = []
		if policy:
			out.append(torch.zeros(len(x), 12))
		if value:
			base = -(20 - (x * self.solved_oh).sum(dim=1, keepdim=True))
			out.append(base + torch.remainder((x * self.w).sum(dim=1, keepdim=True), 7.0) - 3.0)
		return out if len(out) > 1 else out[0]


def tables():
	with open(os.path.join(REF, "frontend", "src", "assets", "maps.json")) as f:
		front = json.load(f)
	cube.set_is2024(False)
	solved686 = cube.get_solved()
	cube.set_is2024(True)
	np.savez_compressed(
		os.path.join(OUT, "cube_tables.npz"),
		delta_maps=_Cube2024.maps,
		frontend_map_neg=np.array(front["map_neg"], dtype=np.int8),
		frontend_map_pos=np.array(front["map_pos"], dtype=np.int8),
		solved2024=cube.get_solved(),
		solved686=solved686,
		action_space=np.array(cube.action_space),
	)


def kats(skip_1m: bool):
	d, text = {}, {}
	np.random.seed(0)
	d["k1_state"], d["k1_faces"], d["k1_dirs"] = cube.scramble(5)
	np.random.seed(42)
	d["k2a_state"], d["k2a_faces"], d["k2a_dirs"] = cube.scramble(1)
	d["k2b_state"], d["k2b_faces"], d["k2b_dirs"] = cube.scramble(20)
	np.random.seed(7)
	d["k7_state"], d["k7_faces"], d["k7_dirs"] = cube.scramble(6, True)

	# K3 small: full arrays
	np.random.seed(1)
	p = random_walk(256, 20)
	d["k3_256_parents"], d["k3_256_children"] = p, fan_out(p)
	# per-row random actions, both directions (the reference's own test only draws dir 0)
	np.random.seed(2)
	f, dr = np.random.randint(0, 6, 256), np.random.randint(0, 2, 256)
	d["mr_faces"], d["mr_dirs"], d["mr_out"] = f, dr, cube.multi_rotate(p, f, dr)
	# goal test truth table: a batch with solved rows sprinkled in
	mix = p.copy()
	mix[[3, 77, 200]] = cube.get_solved()
	d["solved_mix"], d["solved_mix_flags"] = mix, cube.multi_is_solved(mix)
	# parents one move from solved: the fan-out must flag exactly the inverse move
	near = cube.multi_rotate(cube.repeat_state(cube.get_solved(), 12), *cube.iter_actions())
	d["near_parents"] = near
	d["near_children_solved"] = cube.multi_is_solved(fan_out(near))

	np.random.seed(1)
	p = random_walk(10_000, 20)
	c = fan_out(p)
	text["k3_10k_parents_sha256"], text["k3_10k_children_sha256"] = sha(p), sha(c)
	text["k3_10k_solved_children"] = int(cube.multi_is_solved(c).sum())
	d["k3_10k_parent0"], d["k3_10k_child0"] = p[0], c[0]

	if not skip_1m:
		np.random.seed(1)
		p = random_walk(1_000_000, 20)
		h, nsolved = hashlib.sha256(), 0
		for lo in range(0, len(p), 100_000):
			c = fan_out(p[lo:lo + 100_000])
			h.update(np.ascontiguousarray(c).tobytes())
			nsolved += int(cube.multi_is_solved(c).sum())
		text["k3_1m_parents_sha256"], text["k3_1m_children_sha256"] = sha(p), h.hexdigest()
		text["k3_1m_solved_children"] = nsolved
		d["k3_1m_parent0"] = p[0]
	else:
		old = json.load(open(os.path.join(OUT, "cube_text.json")))
		for k in ("k3_1m_parents_sha256", "k3_1m_children_sha256", "k3_1m_solved_children"):
			text[k] = old[k]
		d["k3_1m_parent0"] = np.load(os.path.join(OUT, "cube_kat.npz"))["k3_1m_parent0"]

	np.random.seed(0)
	s, oh = cube.sequence_scrambler(4, 5, True)
	d["k5_states"], d["k5_oh"] = s, oh.numpy()
	np.random.seed(0)
	s, oh = cube.sequence_scrambler(3, 4, False)
	d["k5b_states"] = s
	d["oh_single"] = cube.as_oh(d["k1_state"]).numpy()

	# rendering known answers (the three literals of tests/test_cube.py plus random states)
	st = cube.get_solved()
	text["str_solved"] = cube.stringify(st)
	text["str_F"] = cube.stringify(cube.rotate(st, 0, 1))
	for m in ((0, 0), (1, 0), (2, 0), (3, 0), (4, 0), (5, 0), (0, 1), (1, 1), (2, 1), (3, 1), (4, 1), (5, 1)):
		st = cube.rotate(st, *m)
	text["str_all12"] = cube.stringify(st)
	d["as633_states"] = d["k3_256_parents"][:16]
	d["as633_out"] = np.array([cube.as633(s) for s in d["as633_states"]])
	text["iter_actions_2"] = cube.iter_actions(2).tolist()
	text["rev_actions"] = cube.rev_actions(np.arange(12)).tolist()
	# the public names of the module a drop-in has to carry (cube/cube.py:22 re-exports the maps helpers; modules are not names of the API)
	import types
	text["public_names_cube"] = sorted(n for n, v in vars(cube).items() if not n.startswith("_") and not isinstance(v, types.ModuleType))

	# 6x8x6 representation
	cube.set_is2024(False)
	np.random.seed(5)
	s6 = random_walk(64, 12)
	f, dr = np.random.randint(0, 6, 64), np.random.randint(0, 2, 64)
	d["r686_states"], d["r686_faces"], d["r686_dirs"] = s6, f, dr
	d["r686_out"] = cube.multi_rotate(s6, f, dr)
	d["r686_all12"] = np.array([[cube.rotate(s, *cube.action_space[a]) for a in range(12)] for s in s6[:8]])
	d["r686_correct"] = cube.as_correct(torch.from_numpy(s6)).numpy()
	d["r686_as633"] = np.array([cube.as633(s) for s in s6[:8]])
	st = cube.rotate(cube.rotate(cube.get_solved(), 0, True), 5, False)
	d["r686_correct_FRp"] = cube.as_correct(torch.from_numpy(st).unsqueeze(0)).numpy()
	text["str686_F"] = cube.stringify(cube.rotate(cube.get_solved(), 0, 1))
	cube.set_is2024(True)

	np.savez_compressed(os.path.join(OUT, "cube_kat.npz"), **d)
	with open(os.path.join(OUT, "cube_text.json"), "w") as f:
		json.dump(text, f, indent=1, sort_keys=True)


def astar_traces():
	"""Unmodified reference AStar + exact stub; also records the pop order of every iteration."""
	out = {}
	cases = {
		"a": dict(seed=7, depth=6, lambda_=0.5, expansions=10, max_states=200_000),
		"b": dict(seed=11, depth=8, lambda_=0.2, expansions=64, max_states=3_000),   # runs out of budget
		"c": dict(seed=3, depth=5, lambda_=1.0, expansions=1, max_states=50_000),
		"d": dict(seed=19, depth=7, lambda_=0.1, expansions=300, max_states=60_000),
		# a misleading heuristic (NoisyStubNet): states are rediscovered over shorter ways, both relaxation cases happen, several
		# shortcuts hit one parent in one batch (the last assignment must stay).  The counts are recorded with the trace.
		"e": dict(seed=11, depth=14, lambda_=0.05, expansions=50, max_states=20_000, net="noisy"),
		"f": dict(seed=12, depth=16, lambda_=0.02, expansions=200, max_states=20_000, net="noisy"),
	}
	for tag, c in cases.items():
		np.random.seed(c["seed"])
		state, faces, dirs = cube.scramble(c["depth"], True)
		agent = agents.AStar(NoisyStubNet() if c.get("net") == "noisy" else StubNet(), lambda_=c["lambda_"], expansions=c["expansions"])
		relax = [0, 0]
		inner_relax = agent.relax_seen_states
		def counted_relax(*a, inner_relax=inner_relax, relax=relax, agent=agent, **k):
			before = agent.G.copy()
			r = inner_relax(*a, **k)
			relax[0] += int((agent.G[:len(before)] != before).sum())
			relax[1] += 1
			return r
		agent.relax_seen_states = counted_relax
		pops = []
		inner = agent.expand_batch
		agent.expand_batch = lambda idcs, inner=inner, pops=pops: (pops.append(np.array(idcs)), inner(idcs))[1]
		solved = agent.search(state, time_limit=None, max_states=c["max_states"])
		n = len(agent)
		out[f"{tag}_params"] = np.array([c["seed"], c["depth"], c["expansions"], c["max_states"]])
		out[f"{tag}_lambda"] = np.array(c["lambda_"])
		out[f"{tag}_start"] = state
		out[f"{tag}_solved"] = np.array(solved)
		out[f"{tag}_n"] = np.array(n)
		out[f"{tag}_states"] = agent.states[1:n + 1].copy()
		out[f"{tag}_G"] = agent.G[1:n + 1].copy()
		out[f"{tag}_parents"] = agent.parents[2:n + 1].astype(np.int64)
		out[f"{tag}_parent_actions"] = agent.parent_actions[2:n + 1].astype(np.int64)
		out[f"{tag}_action_queue"] = np.array(list(agent.action_queue), dtype=np.int64)
		out[f"{tag}_pop_lens"] = np.array([len(p) for p in pops])
		out[f"{tag}_pops"] = np.concatenate(pops) if pops else np.zeros(0, dtype=np.int64)
		out[f"{tag}_relaxed"] = np.array(relax[0])                           # G entries lowered by relax_seen_states over the whole search
		print(f"astar {tag}: solved={solved} n={n} iters={len(pops)} queue_len={len(agent.action_queue)} G entries relaxed={relax[0]}")
	np.savez_compressed(os.path.join(OUT, "astar_trace.npz"), **out)


def mcts_traces():
	out = {}
	cases = {
		"a": dict(seed=7, depth=6, c=0.6, search_graph=False, max_states=4_000),
		"b": dict(seed=2, depth=5, c=5.0, search_graph=True, max_states=10_000),    # solves at 2 474 states
		"c": dict(seed=2, depth=4, c=5.0, search_graph=False, max_states=10_000),   # solves at 3 307 states
		"d": dict(seed=23, depth=12, c=2.5, search_graph=False, max_states=2_500),
		"e": dict(seed=2, depth=2, c=1.0, search_graph=True, max_states=10_000),    # solves at 1 568 states
		"f": dict(seed=31, depth=9, c=3.0, search_graph=False, max_states=3_000, net="policy"),   # non-uniform priors
		# search_graph cases in which the breadth-first search really SHORTENS the queue (the descents run in circles under a
		# strongly non-uniform policy and a large c): 9 moves found, 3 after agents.py:613-633; 9 -> 5
		"g": dict(seed=1032, depth=3, c=50.0, search_graph=True, max_states=4_000, net="policy"),
		"h": dict(seed=1010, depth=5, c=50.0, search_graph=True, max_states=4_000, net="policy"),
	}
	for tag, c in cases.items():
		np.random.seed(c["seed"])
		state, faces, dirs = cube.scramble(c["depth"], True)
		agent = agents.MCTS(PolicyStubNet() if c.get("net") == "policy" else StubNet(), c=c["c"], search_graph=c["search_graph"])
		sims = [0]
		inner = agent.expand_leaf
		def counted(v, a, inner=inner, sims=sims):
			sims[0] += 1
			return inner(v, a)
		agent.expand_leaf = counted
		solved = agent.search(state, time_limit=None, max_states=c["max_states"])
		n = len(agent)
		out[f"{tag}_params"] = np.array([c["seed"], c["depth"], int(c["search_graph"]), c["max_states"]])
		out[f"{tag}_c"] = np.array(c["c"])
		out[f"{tag}_start"] = state
		out[f"{tag}_solved"] = np.array(solved)
		out[f"{tag}_n"] = np.array(n)
		out[f"{tag}_sims"] = np.array(sims[0])
		out[f"{tag}_states"] = agent.states[1:n + 1].copy()
		out[f"{tag}_neighbors"] = agent.neighbors[1:n + 1].astype(np.int32)
		out[f"{tag}_leaves"] = agent.leaves[1:n + 1].copy()
		out[f"{tag}_N"] = agent.N[1:n + 1].astype(np.int32)
		out[f"{tag}_W"] = agent.W[1:n + 1].astype(np.float32)   # exact: stub values are small integers
		out[f"{tag}_L"] = agent.L[1:n + 1].astype(np.float32)
		out[f"{tag}_V"] = agent.V[1:n + 1].astype(np.float32)
		if c.get("net") == "policy":
			out[f"{tag}_P"] = agent.P[1:n + 1].copy()                       # float64 of float32 softmax outputs
		out[f"{tag}_action_queue"] = np.array(list(agent.action_queue), dtype=np.int64)
		assert (agent.W[1:n + 1] == out[f"{tag}_W"]).all() and (agent.L[1:n + 1] == out[f"{tag}_L"]).all()
		print(f"mcts {tag}: solved={solved} n={n} sims={sims[0]} queue_len={len(agent.action_queue)}")
	np.savez_compressed(os.path.join(OUT, "mcts_trace.npz"), **out)


class PolicyStubNet(StubNet):
	"""
	The stub with a NON-uniform policy whose softmax is exact on any hardware: every logit is 0 or -inf, and the number
	of finite logits is 8 (corner 0 has an even code) or 4 (odd code), so the probabilities are exactly 0, 1/8 or 1/4
	whether an implementation divides by the sum or multiplies by its reciprocal.  Used for one MCTS trace so that
	selection with unequal priors is pinned to the reference and not only to the oracle port.
	"""
	@staticmethod
	def logits_table() -> np.ndarray:
		t = np.zeros((24, 12), np.float32)
		for c in range(24):
			banned = [(c + 3 * j) % 12 for j in range(4)] if c % 2 == 0 else [(c + j) % 12 for j in range(8)]
			t[c, banned] = -np.inf
		return t

	def __call__(self, x, policy=True, value=True):
		out = []
		if policy:
			code = x[:, :24].argmax(dim=1)                                   # code of corner 0 (0..23)
			out.append(torch.from_numpy(self.logits_table())[code])
		if value:
			out.append(-(20 - (x * self.solved_oh).sum(dim=1, keepdim=True)))
		return out if len(out) > 1 else out[0]


def adi_traces():
	"""
	The unmodified reference `Train.ADI_traindata` (train.py:256-339) with the exact-integer stub net, fixed seeds, all
	four reward methods.  `self` is a bare namespace carrying exactly the attributes the method reads.
	"""
	import types
	from librubiks import train as ref_train
	from librubiks.utils.ticktock import TickTock
	out = {}
	cases = {
		"lapanfix": dict(seed=12, games=37, depth=9, alpha=0.3, ff=3),
		"paper": dict(seed=12, games=37, depth=9, alpha=0.3, ff=3),
		"schultzfix": dict(seed=13, games=20, depth=11, alpha=0.0, ff=1),
		"reward0": dict(seed=14, games=50, depth=6, alpha=1.0, ff=4),
	}
	for method, c in cases.items():
		me = types.SimpleNamespace(rollout_games=c["games"], rollout_depth=c["depth"], reward_method=method,
		                           adi_ff_batches=c["ff"], tt=TickTock(), with_analysis=False)
		me._get_adi_ff_slices = types.MethodType(ref_train.Train._get_adi_ff_slices, me)
		np.random.seed(c["seed"])
		oh, policy, value, lw = ref_train.Train.ADI_traindata(me, StubNet(), c["alpha"])
		out[f"{method}_params"] = np.array([c["seed"], c["games"], c["depth"], c["ff"]])
		out[f"{method}_alpha"] = np.array(c["alpha"])
		out[f"{method}_oh_sha256"] = np.array(sha(oh.cpu().numpy()))
		out[f"{method}_oh_idx"] = oh.cpu().numpy().reshape(len(oh), 20, 24).argmax(axis=2).astype(np.int8)   # = the states
		out[f"{method}_policy"] = policy.numpy().astype(np.int64)
		out[f"{method}_value"] = value.numpy().astype(np.float32)
		out[f"{method}_loss_weights"] = lw.numpy().astype(np.float32)
		print(f"adi {method}: n={len(oh)} value range {float(value.min())}..{float(value.max())} zeros={(value == 0).sum().item()}")
	np.savez_compressed(os.path.join(OUT, "adi_trace.npz"), **out)


def evaluator_traces():
	"""
	The unmodified reference `Evaluator.eval` (solving/evaluation.py:56-96): scrambles drawn from the global generator between the
	searches, one `agent.search` per game, bounded by max_states.  Recorded: res (moves of each solution, -1 = none) and states
	(len(agent) after each game).  The start states are recorded too, so that a CPU test can replay the games with the oracle.
	"""
	from librubiks.solving.evaluation import Evaluator
	out = {}
	cases = {
		"astar": dict(seed=31, games=4, depths=[3, 6, 9, 12], max_states=6_000, agent=lambda: agents.AStar(StubNet(), lambda_=0.2, expansions=30)),
		"astar_noisy": dict(seed=32, games=3, depths=[8, 14], max_states=5_000, agent=lambda: agents.AStar(NoisyStubNet(), lambda_=0.05, expansions=50)),
		"astar_deep": dict(seed=35, games=3, depths=range(0), max_states=2_000, agent=lambda: agents.AStar(StubNet(), lambda_=0.5, expansions=10)),
		"mcts_graph": dict(seed=33, games=4, depths=[2, 4, 5], max_states=4_000, agent=lambda: agents.MCTS(PolicyStubNet(), c=5.0, search_graph=True)),
		"mcts": dict(seed=36, games=3, depths=[2, 4], max_states=3_000, agent=lambda: agents.MCTS(StubNet(), c=5.0, search_graph=False)),
		"bfs": dict(seed=34, games=3, depths=[1, 2, 3], max_states=3_000, agent=lambda: agents.BFS()),
		"bfs_budget": dict(seed=37, games=2, depths=[4, 7], max_states=5_000, agent=lambda: agents.BFS()),      # the budget ends the deeper games mid-layer
	}
	for tag, c in cases.items():
		agent = c["agent"]()
		starts = []
		inner = agent.search
		def recorded(state, *a, inner=inner, starts=starts, **k):
			starts.append(np.array(state))
			return inner(state, *a, **k)
		agent.search = recorded
		np.random.seed(c["seed"])
		ev = Evaluator(n_games=c["games"], scrambling_depths=c["depths"], max_time=None, max_states=c["max_states"])
		res, states, times = ev.eval(agent)
		deep = c["depths"] == range(0)
		out[f"{tag}_params"] = np.array([c["seed"], c["games"], c["max_states"], int(deep)])
		out[f"{tag}_depths"] = np.array([0] if deep else list(c["depths"]))
		out[f"{tag}_starts"] = np.array(starts, dtype=np.int8)
		out[f"{tag}_res"] = res.astype(np.int64)
		out[f"{tag}_states"] = states.astype(np.int64)
		print(f"evaluator {tag}: res={res.tolist()} states={states.tolist()}")
	np.savez_compressed(os.path.join(OUT, "evaluator_trace.npz"), **out)


if __name__ == "__main__":
	ap = argparse.ArgumentParser()
	ap.add_argument("--skip-1m", action="store_true", help="reuse the committed 1 M-state hashes (saves ~1 min)")
	ap.add_argument("--only", default="")
	args = ap.parse_args()
	os.makedirs(OUT, exist_ok=True)
	todo = args.only.split(",") if args.only else ["tables", "kats", "astar", "mcts", "adi", "evaluator"]
	if "tables" in todo: tables()
	if "kats" in todo: kats(args.skip_1m)
	if "astar" in todo: astar_traces()
	if "mcts" in todo: mcts_traces()
	if "adi" in todo: adi_traces()
	if "evaluator" in todo: evaluator_traces()
	print("golden vectors written to", OUT)
