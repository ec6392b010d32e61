"""
The reference's Evaluator (ref:librubiks/solving/evaluation.py:56-96) as a composition of things the oracle already restates:
scrambles drawn from the global generator (game by game, the deep depth first) and one search per game.  The fixture
tests/golden/evaluator_trace.npz holds what the UNMODIFIED Evaluator returned over the unmodified agents (oracle/gen_golden.py);
here the CPU oracle replays those games -- which pins the order of the draws and the meaning of `res` and `states` that
librubiks_amd.solving.evaluation is tested against on the GPU (tests/test_evaluation_gpu.py).
"""
import os

import numpy as np
import pytest

from oracle import cube_oracle as orc
from oracle.search_oracle import AStarOracle, BFSOracle, MCTSOracle, NoisyStubNet, PolicyStubNet, StubNet

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "evaluator_trace.npz"))

#: the agents of oracle/gen_golden.py::evaluator_traces, as oracle searches
CASES = {
	"astar": lambda: AStarOracle(StubNet(), 0.2, 30),
	"astar_noisy": lambda: AStarOracle(NoisyStubNet(), 0.05, 50),
	"astar_deep": lambda: AStarOracle(StubNet(), 0.5, 10),
	"mcts_graph": lambda: MCTSOracle(PolicyStubNet(), 5.0, True),
	"mcts": lambda: MCTSOracle(StubNet(), 5.0, False),
	"bfs": BFSOracle,
	"bfs_budget": BFSOracle,
}


def drawn_starts(tag: str) -> np.ndarray:
	"""The scrambles in the order evaluation.py:68-74 draws them."""
	seed, games, _, deep = (int(x) for x in GOLD[f"{tag}_params"])
	np.random.seed(seed)
	starts = []
	for d in GOLD[f"{tag}_depths"]:
		for _ in range(games):
			if deep:
				d = np.random.randint(100, 1000)
			starts.append(orc.scramble(int(d), True)[0])
	return np.array(starts, dtype=np.int8)


@pytest.mark.parametrize("tag", list(CASES))
def test_scrambles_are_drawn_game_by_game(tag):
	assert (drawn_starts(tag) == GOLD[f"{tag}_starts"]).all()


@pytest.mark.parametrize("tag", list(CASES))
def test_oracle_replays_the_reference_evaluator(tag):
	_, games, max_states, _ = (int(x) for x in GOLD[f"{tag}_params"])
	res, states = [], []
	for start in GOLD[f"{tag}_starts"]:
		agent = CASES[tag]()
		solved = agent.search(start, max_states)
		res.append(len(agent.action_queue) if solved else -1)
		states.append(len(agent))
	shape = GOLD[f"{tag}_res"].shape
	assert (np.reshape(res, shape) == GOLD[f"{tag}_res"]).all()
	assert (np.reshape(states, shape) == GOLD[f"{tag}_states"]).all()
