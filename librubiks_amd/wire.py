"""
The demo server's wire format on the device engines (reference: librubiks/api.py:43-62 and the TypeScript interfaces
IInfoResponse / ISolveRequest / ISolveResponse in frontend/src/app/common/rubiks.ts:12-28).

Transport is not this project's business (the reference uses Flask, which also downloads its weights at import);
`SolveService` maps request dictionaries to response dictionaries, so any HTTP layer can wrap it:

    service = SolveService(net)                                  # a value/policy net on the GPU
    service.info()                                               # -> IInfoResponse
    service.solve({"agentIdx": 0, "timeLimit": 5, "state": [...20 ints...]})   # -> ISolveResponse
    service.solve_json(body_bytes)                               # the same from / to JSON text
"""
from __future__ import annotations

import json

import numpy as np
import torch

from librubiks_amd import cube
from librubiks_amd.solving.agents import AStar, BFS, EGVM, MCTS, PolicySearch, RandomSearch, ValueSearch

#: the reference's serving parameters (api.py:25-27)
ASTAR_PARAMS = {"lambda_": 0.07, "expansions": 27}
MCTS_PARAMS = {"c": 4.13}
EGVM_PARAMS = {"epsilon": 0.375, "workers": 10, "depth": 50}


class SolveService:
	def __init__(self, net, max_states: int = 2_000_000):
		self.max_states = max_states
		self.agents = [                                          # order = agentIdx of the frontend (api.py:29-37)
			# time-limited searches start with pools of `max_states` / 200 000 nodes and grow them (doubling) while
			# time is left; a search that hits the agents' max_capacity reports it (agent.capacity_exhausted)
			("A*", AStar(net, **ASTAR_PARAMS, capacity=max_states)),
			("MCTS", MCTS(net, **MCTS_PARAMS, search_graph=True, capacity=min(max_states, 200_000))),
			("Greedy policy", PolicySearch(net)),
			("Greedy value", ValueSearch(net)),
			("EGVM", EGVM(net, **EGVM_PARAMS)),
			("BFS", BFS()),
			("Random actions", RandomSearch()),
		]

	def info(self) -> dict:
		return {
			"cuda": torch.cuda.is_available(),
			"agents": [name for name, _ in self.agents],
			"parameters": {"A*": ASTAR_PARAMS, "MCTS": MCTS_PARAMS, "EGVM": EGVM_PARAMS},
		}

	def solve(self, request: dict) -> dict:
		idx = int(request["agentIdx"])
		if not 0 <= idx < len(self.agents):
			raise IndexError(f"agentIdx {idx} outside 0..{len(self.agents) - 1}")
		state = np.asarray(request["state"], dtype=cube.dtype)
		if state.shape != (20,) or state.min() < 0 or state.max() > 23:
			raise ValueError("state must be 20 cubie codes in 0..23")
		agent = self.agents[idx][1]
		found = agent.search(state, float(request["timeLimit"]))
		out = {"solution": bool(found), "actions": [int(a) for a in agent.action_queue], "exploredStates": int(len(agent))}
		if getattr(agent, "capacity_exhausted", False):           # not part of the reference's contract: extra key, only when it happened
			out["capacityExhausted"] = True
		return out

	def solve_json(self, body) -> str:
		if isinstance(body, (bytes, bytearray)):
			body = body.decode("utf-8")
		return json.dumps(self.solve(json.loads(body)))
