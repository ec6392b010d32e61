"""
The reference's evaluation loop (ref:librubiks/solving/evaluation.py:56-96) on one MI355X: the same games played one after the
other through the drop-in agents, and in lock-step on the device (librubiks_amd.solving.evaluation.Evaluator, `batched`).
Games are bounded by max_states, as the batched form requires.  With the exact stub net `res`/`states` must be equal in both
modes (asserted); with the random-init bf16 fc_small the batch shapes of the forwards differ between the modes, so values may
differ in their last bf16 bit and single games may take another course: the share of equal entries is reported, not asserted.

    python benchmarks/evaluator.py > profiles/r04_evaluator.json
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks.nets import FastStub, FcSmall  # noqa: E402
from librubiks_amd.solving import agents  # noqa: E402
from librubiks_amd.solving.evaluation import Evaluator  # noqa: E402

bf16 = FcSmall().cuda().eval().to(torch.bfloat16)
CASES = [
	# the reference's evaluation settings are 100 games per depth; bounded here so that the sequential leg stays within a minute
	("AStar, stub net, N = 100, lambda = 0.2", lambda: agents.AStar(FastStub(), 0.2, 100, capacity=60_000), dict(n_games=32, depths=[8, 12, 16], max_states=50_000)),
	("AStar, fc_small bf16 fused + folded, N = 100, lambda = 0.2", lambda: agents.AStar(bf16, 0.2, 100, capacity=60_000, fused_first_layer="folded"),
	 dict(n_games=32, depths=[8, 12, 16], max_states=50_000)),
	("MCTS, stub net, c = 0.6, search_graph, priors in the kernel", lambda: agents.MCTS(FastStub(), 0.6, True, capacity=10_000, priors="kernel"),
	 dict(n_games=32, depths=[4, 8, 12], max_states=8_000)),
	("MCTS, fc_small bf16, c = 0.6, search_graph, priors in the kernel", lambda: agents.MCTS(bf16, 0.6, True, capacity=10_000, priors="kernel"),
	 dict(n_games=32, depths=[4, 8, 12], max_states=8_000)),
]
for name, make, c in CASES:
	out = {"agent": name, **c, "games": c["n_games"] * len(c["depths"])}
	results = {}
	for mode in ("sequential", "batched"):
		agent = make()
		np.random.seed(5)
		Evaluator(2, [3], max_states=2_000).eval(agent, batched=mode == "batched")        # warm: engines, allocator, library handles
		np.random.seed(11)
		ev = Evaluator(c["n_games"], c["depths"], None, c["max_states"], batch_games=96)
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		res, states, times = ev.eval(agent, batched=mode == "batched")
		torch.cuda.synchronize()
		dt = time.perf_counter() - t0
		results[mode] = (res, states)
		out[mode] = {"seconds": dt, "games_per_s": res.size / dt, "states_per_s": float(states.sum()) / dt,
		             "solved_share_by_depth": [(r != -1).mean() for r in res], "states_total": int(states.sum())}
	same_res = float((results["sequential"][0] == results["batched"][0]).mean())
	same_states = float((results["sequential"][1] == results["batched"][1]).mean())
	if "stub" in name:
		assert same_res == 1.0 and same_states == 1.0, name
	out["equal_entries"] = {"res": same_res, "states": same_states}
	out["batched_over_sequential"] = out["sequential"]["seconds"] / out["batched"]["seconds"]
	print(json.dumps(out), flush=True)
