"""The /info and /solve contract of the reference's demo server (api.py:43-62, rubiks.ts:12-28) on the device engines."""
import json

import numpy as np
import pytest

from librubiks_amd import cube
from librubiks_amd.wire import SolveService
from oracle import cube_oracle as orc
from oracle.search_oracle import StubNet

pytestmark = pytest.mark.gpu


def test_info_and_solve_contract():
	service = SolveService(StubNet(), max_states=100_000)
	info = service.info()
	assert info["cuda"] is True and info["agents"][:2] == ["A*", "MCTS"] and len(info["agents"]) == 7
	assert info["parameters"]["A*"] == {"lambda_": 0.07, "expansions": 27}
	np.random.seed(4)
	state, _, _ = orc.scramble(4, True)
	for idx in (0, 1, 3, 5):
		body = json.dumps({"agentIdx": idx, "timeLimit": 2, "state": [int(x) for x in state]})
		resp = json.loads(service.solve_json(body.encode()))
		assert set(resp) == {"solution", "actions", "exploredStates"}
		assert isinstance(resp["solution"], bool) and isinstance(resp["exploredStates"], int)
		s = state
		for a in resp["actions"]:
			s = cube.rotate(s, *cube.action_space[a])
		assert cube.is_solved(s) == resp["solution"], idx
	assert json.loads(service.solve_json(json.dumps({"agentIdx": 5, "timeLimit": 1, "state": cube.get_solved().tolist()}))) == \
		{"solution": True, "actions": [], "exploredStates": 0}
	with pytest.raises(ValueError):
		service.solve({"agentIdx": 0, "timeLimit": 1, "state": [0] * 19})
	with pytest.raises(IndexError):
		service.solve({"agentIdx": 9, "timeLimit": 1, "state": cube.get_solved().tolist()})
