"""
Which torch operators a search step issues around the engine's kernels (torch.profiler, operator table by input shape): the net's
tail behind the fused first layer -- GEMMs, activations and any copy torch adds on its own.  A diagnostic, not a timing.

    python benchmarks/step_ops.py [mcts|astar]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
	what = sys.argv[1] if len(sys.argv) > 1 else "mcts"
	from benchmarks.nets import FcSmall
	from librubiks_amd import cube
	from librubiks_amd.solving.agents import AStar, MCTSBatch
	from torch.profiler import ProfilerActivity, profile
	net = FcSmall().cuda().eval().to(torch.bfloat16)
	if what == "mcts":
		T = 256
		starts = []
		for g in range(T):
			np.random.seed(g)
			starts.append(cube.scramble(14, True)[0])
		starts = np.array(starts)
		agent = MCTSBatch(net, 0.6, T, capacity=12 * 64 + 64, max_path=4096, fused_first_layer="folded")
		run = lambda: agent.search(starts, max_states=12 * 64 + 64, max_sims=24, use_graph=False, poll=8)
	else:
		agent = AStar(net, 0.16, 1000, fused_first_layer="folded")
		np.random.seed(0)
		st = cube.scramble(14, True)[0]
		run = lambda: agent.search(st, time_limit=None, max_states=40_000)
	run()
	torch.cuda.synchronize()
	with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
		run()
		torch.cuda.synchronize()
	print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60, max_shapes_column_width=90))


if __name__ == "__main__":
	main()
