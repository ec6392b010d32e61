"""
The heads' last layer as one launch (`rk_tail_linear`, oh_linear.TailLinear) against the two torch launches it replaces, on the
searches themselves and on one box: configs[3] (256 trees, hipGraph-replayed step) and configs[2] (A*, N = 1000), fc_small bf16,
first layer fused + folded, `oh_linear.FUSE_TAIL` off / on alternating.

    python benchmarks/tail_ab.py [--sims 1024] > profiles/r05_tail_ab.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--sims", type=int, default=1024)
	ap.add_argument("--rounds", type=int, default=3)
	args = ap.parse_args()
	from benchmarks.nets import FcSmall
	from librubiks_amd import cube, oh_linear
	from librubiks_amd.solving.agents import AStar, MCTSBatch
	net = FcSmall().cuda().eval().to(torch.bfloat16)
	T, depth = 256, 14
	starts = []
	for g in range(T):
		np.random.seed(g)
		starts.append(cube.scramble(depth, True)[0])
	starts = np.array(starts)
	cap = 12 * args.sims + 64
	games = []
	for g in range(3):
		np.random.seed(g)
		games.append(cube.scramble(depth, True)[0])

	def mcts():
		trees = MCTSBatch(net, 0.6, T, capacity=cap, max_path=16384, fused_first_layer="folded")
		trees.search(starts, max_states=cap, max_sims=16, use_graph=True, poll=8)
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		trees.search(starts, max_states=cap, max_sims=args.sims, use_graph=True, poll=64)
		torch.cuda.synchronize()
		dt = time.perf_counter() - t0
		steps = trees.simulations
		del trees
		torch.cuda.empty_cache()
		return dt / max(steps, 1) * 1e3

	def astar():
		agent = AStar(net, 0.16, 1000, fused_first_layer="folded")
		agent.search(games[0], time_limit=None, max_states=40_000)
		tot = it = 0
		for st in games:
			torch.cuda.synchronize()
			t0 = time.perf_counter()
			agent.search(st, time_limit=None, max_states=150_000)
			torch.cuda.synchronize()
			tot += time.perf_counter() - t0
			it += agent.iterations
		del agent
		torch.cuda.empty_cache()
		return tot / max(it, 1) * 1e3

	rec = {"unit": "ms per MCTS step (256 trees) / per A* iteration (N = 1000), wall clock", "sims": args.sims, "mcts": {"torch tail": [], "rk_tail_linear": []},
	       "astar": {"torch tail": [], "rk_tail_linear": []}}
	for _ in range(args.rounds):
		for on in (False, True):
			oh_linear.FUSE_TAIL = on
			key = "rk_tail_linear" if on else "torch tail"
			rec["mcts"][key].append(mcts())
			rec["astar"][key].append(astar())
			print(key, rec["mcts"][key][-1], rec["astar"][key][-1], file=sys.stderr, flush=True)
	oh_linear.FUSE_TAIL = True
	for leg in ("mcts", "astar"):
		for key in ("torch tail", "rk_tail_linear"):
			rec[leg][key + ", median"] = sorted(rec[leg][key])[len(rec[leg][key]) // 2]
	print(json.dumps(rec))


if __name__ == "__main__":
	main()
