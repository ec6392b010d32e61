"""
Pins the search oracle (oracle/search_oracle.py) to the traces captured from the UNMODIFIED reference A* and MCTS
agents (oracle/gen_golden.py, exact-integer stub net).  CPU only.
"""
import numpy as np
import pytest

from oracle import cube_oracle as orc
from oracle.search_oracle import AStarOracle, MCTSOracle, NoisyStubNet, PolicyStubNet, StubNet, adi_traindata_oracle


def _apply(state, queue):
	for a in queue:
		state = orc.rotate(state, a // 2, 1 - a % 2)
	return state


@pytest.mark.parametrize("tag", ["a", "b", "d", "c", "e", "f"])
def test_astar_oracle_reproduces_reference_trace(golden, tag):
	"""Traces e and f were driven by the misleading NoisyStubNet: the unmodified reference lowered 26 / 2 G entries in
	relax_seen_states (both cases, several shortcuts on one parent in one batch) -- with the plain stub it almost never does."""
	t = golden["astar_trace"]
	seed, depth, expansions, max_states = (int(x) for x in t[f"{tag}_params"])
	np.random.seed(seed)
	start, _, _ = orc.scramble(depth, True)
	assert (start == t[f"{tag}_start"]).all()
	agent = AStarOracle(NoisyStubNet() if tag in ("e", "f") else StubNet(), float(t[f"{tag}_lambda"]), expansions)
	solved = agent.search(start, max_states)
	assert solved == bool(t[f"{tag}_solved"]) and len(agent) == int(t[f"{tag}_n"])
	assert int(t["e_relaxed"]) >= 20                                       # the relaxation code really ran in the reference
	states, G, parents, pact = agent.arrays()
	assert (states == t[f"{tag}_states"]).all()
	assert (G == t[f"{tag}_G"]).all()
	assert (parents == t[f"{tag}_parents"]).all()
	assert (pact == t[f"{tag}_parent_actions"]).all()
	assert list(agent.action_queue) == t[f"{tag}_action_queue"].tolist()
	assert [len(p) for p in agent.pops] == t[f"{tag}_pop_lens"].tolist()
	assert (np.concatenate(agent.pops) == t[f"{tag}_pops"]).all()
	if solved:
		assert orc.is_solved(_apply(start, agent.action_queue))


def test_astar_trace_a_matches_survey_hashes(golden):
	"""seed 7 / depth 6 / lambda 0.5 / N 10 -> 1 882 states, 19 iterations, queue [2, 8, 3, 8]."""
	import hashlib
	t = golden["astar_trace"]
	assert int(t["a_n"]) == 1882 and len(t["a_pop_lens"]) == 19 and t["a_action_queue"].tolist() == [2, 8, 3, 8]
	assert hashlib.sha256(t["a_states"].tobytes()).hexdigest() == "cda6dfc635e725b5ce2c3f2cdbe9b315c9acad744f8abd44eceda424764dd585"
	assert hashlib.sha256(t["a_G"].tobytes()).hexdigest() == "8d178f1cf5da40b2160bcf08e7427c2f389eb04147605430fe1044655b2ba1f0"
	assert hashlib.sha256(t["a_parents"].tobytes()).hexdigest() == "a6b721d892115c0198cc62e4e1d1ed58ede395ee6c7806a0eb20eed338c989a3"


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e", "f", "g", "h"])
def test_mcts_oracle_reproduces_reference_trace(golden, tag):
	t = golden["mcts_trace"]
	seed, depth, search_graph, max_states = (int(x) for x in t[f"{tag}_params"])
	np.random.seed(seed)
	start, _, _ = orc.scramble(depth, True)
	assert (start == t[f"{tag}_start"]).all()
	agent = MCTSOracle(PolicyStubNet() if f"{tag}_P" in t else StubNet(), float(t[f"{tag}_c"]), bool(search_graph))
	solved = agent.search(start, max_states)
	n = len(agent)
	if f"{tag}_P" in t:                      # traces f, g, h: non-uniform priors, exactly 0, 1/8 or 1/4
		assert (agent.P[1:n + 1] == t[f"{tag}_P"]).all() and len(np.unique(t[f"{tag}_P"])) == 3
	assert solved == bool(t[f"{tag}_solved"]) and n == int(t[f"{tag}_n"]) and agent.sims == int(t[f"{tag}_sims"])
	assert (agent.states[1:n + 1] == t[f"{tag}_states"]).all()
	assert (agent.neighbors[1:n + 1] == t[f"{tag}_neighbors"]).all()
	assert (agent.leaves[1:n + 1] == t[f"{tag}_leaves"]).all()
	assert (agent.N[1:n + 1] == t[f"{tag}_N"]).all()
	assert (agent.W[1:n + 1] == t[f"{tag}_W"]).all()
	assert (agent.L[1:n + 1] == t[f"{tag}_L"]).all()
	assert (agent.V[1:n + 1] == t[f"{tag}_V"]).all()
	assert list(agent.action_queue) == t[f"{tag}_action_queue"].tolist()
	if solved:
		assert orc.is_solved(_apply(start, agent.action_queue))


@pytest.mark.parametrize("method", ["lapanfix", "paper", "schultzfix", "reward0"])
def test_adi_oracle_reproduces_reference(golden, method):
	"""The ADI restatement against the output of the unmodified `Train.ADI_traindata` (train.py:256-339)."""
	import hashlib
	t = golden["adi_trace"]
	seed, games, depth, _ = (int(x) for x in t[f"{method}_params"])
	np.random.seed(seed)
	oh, policy, value, lw = adi_traindata_oracle(StubNet(), games, depth, float(t[f"{method}_alpha"]), method)
	assert hashlib.sha256(np.ascontiguousarray(oh).tobytes()).hexdigest() == str(t[f"{method}_oh_sha256"])
	assert (oh.reshape(len(oh), 20, 24).argmax(axis=2) == t[f"{method}_oh_idx"]).all()
	assert (policy == t[f"{method}_policy"]).all()
	assert (value == t[f"{method}_value"]).all()
	assert np.allclose(lw, t[f"{method}_loss_weights"], rtol=1e-6, atol=0)
