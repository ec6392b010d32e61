"""The one-step agents on the drop-in surface: the body of the reference's tests/test_agents.py:18-36."""
import numpy as np
import pytest
import torch

from librubiks_amd import cube
from librubiks_amd.solving.agents import BFS, EGVM, PolicySearch, RandomSearch, ValueSearch
from oracle.search_oracle import StubNet
from tests.test_astar_gpu import TinyNet

pytestmark = pytest.mark.gpu


def test_agents_action_queue_is_consistent():
	net = TinyNet().cuda().eval()
	agents = [RandomSearch(), BFS(), PolicySearch(net, sample_policy=False), PolicySearch(net, sample_policy=True),
	          ValueSearch(net), EGVM(net, 0.1, 4, 12), ValueSearch(StubNet()), EGVM(StubNet(), 0.2, 8, 6)]
	np.random.seed(0)
	for agent in agents:
		state, _, _ = cube.scramble(4)
		found = agent.search(state, .5)
		assert all(0 <= a < cube.action_dim for a in agent.action_queue)
		s = state
		for a in agent.action_queue:
			s = cube.rotate(s, *cube.action_space[a])
		assert found == cube.is_solved(s), str(agent)


def test_bfs_finds_shortest_paths():
	np.random.seed(3)
	for depth in (1, 2, 3):
		state, faces, dirs = cube.scramble(depth, True)
		agent = BFS()
		assert agent.search(state, time_limit=20, max_states=200_000)
		assert 1 <= len(agent.action_queue) <= depth
		s = state
		for a in agent.action_queue:
			s = cube.rotate(s, *cube.action_space[a])
		assert cube.is_solved(s)
