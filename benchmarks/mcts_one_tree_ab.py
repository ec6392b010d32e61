"""
The reference's own MCTS regime -- ONE tree, twelve rows per net forward (ref:librubiks/solving/agents.py:476-595) -- with the net's two
ends as they were (first layer in its LDS-tiled form, the heads' last layer as torch's activation kernel + narrow GEMM) against what
ships (direct form, rk_tail_linear): microseconds per simulation, fc_small bf16 fused + folded, eager and replayed as a hipGraph, on one
box.  `--trees T` runs T trees in lock-step instead (12 T rows).

    python benchmarks/mcts_one_tree_ab.py > profiles/r05_mcts_one_tree_net_ends.json
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
	ap.add_argument("--trees", type=int, default=1)
	ap.add_argument("--sims", type=int, default=4096)
	ap.add_argument("--rounds", type=int, default=2)
	args = ap.parse_args()
	from benchmarks.nets import FcSmall
	from librubiks_amd import cube, oh_linear
	from librubiks_amd.solving.agents import MCTSBatch
	net = FcSmall().cuda().eval().to(torch.bfloat16)
	starts = []
	for g in range(args.trees):
		np.random.seed(g)
		starts.append(cube.scramble(14, True)[0])
	starts = np.array(starts)
	cap = 12 * args.sims + 64

	def run(graph):
		trees = MCTSBatch(net, 0.6, args.trees, capacity=cap, max_path=16384, fused_first_layer="folded")
		trees.search(starts, max_states=cap, max_sims=32, use_graph=graph, poll=8)
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		trees.search(starts, max_states=cap, max_sims=args.sims, use_graph=graph, poll=64)
		torch.cuda.synchronize()
		dt = time.perf_counter() - t0
		steps = trees.simulations
		del trees
		torch.cuda.empty_cache()
		return dt / max(steps, 1) * 1e6

	rec = {"unit": "us per simulation step, wall clock", "trees": args.trees, "sims": args.sims, "net": "fc_small bf16 random init, first layer fused + folded", "runs": []}
	for _ in range(args.rounds):
		for label, ends in (("first layer tiled, torch tail", ("mfma_tiled", False)), ("direct form, rk_tail_linear", (None, True))):
			oh_linear.MFMA_FORM, oh_linear.FUSE_TAIL = ends
			for graph in (False, True):
				rec["runs"].append({"net ends": label, "hipgraph": graph, "us_per_step": run(graph)})
				print(rec["runs"][-1], file=sys.stderr, flush=True)
	oh_linear.MFMA_FORM, oh_linear.FUSE_TAIL = None, True
	print(json.dumps(rec))


if __name__ == "__main__":
	main()
