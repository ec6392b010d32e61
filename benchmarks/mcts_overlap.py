"""
VERDICT r4 #3: does one half's backup + descent run under the other half's net forward?  configs[3] (256 trees, fc_small bf16, first
layer fused + folded) advanced three ways, each for `--sims` simulations:

    one_stream     the captured step of round 4 (net on all 3 072 children, then backup + select of all 256 trees)
    two_halves     the captured step that advances trees [0,128) and [128,256) on two streams, skewed by half a step
                   (MCTSBatch overlap_halves: backup+select A || net B, then net A || backup+select B)
    two_halves_eager  the same schedule issued launch by launch on two torch streams with events (no hipGraph)

    python benchmarks/mcts_overlap.py [--sims 1024] > profiles/r05_mcts_overlap.json
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 benchmarks/mcts_overlap.py --sims 256 --trace-marks
    python benchmarks/mcts_overlap_summary.py DIR > profiles/r05_mcts_overlap_trace.json

Under rocprofv3 the summary script measures, per form, how much of the backup + select kernels' time lies under another kernel.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks.nets import FcSmall  # noqa: E402
from librubiks_amd import _ffi, cube  # noqa: E402
from librubiks_amd.solving.agents import MCTSBatch  # noqa: E402


def eager_two_halves(agent: MCTSBatch, starts, cap, sims):
	"""The skewed schedule without a graph: two torch streams, events between the phases."""
	agent._begin(starts, cap, sims, use_graph=False)
	T, na = agent.n_trees, agent.n_trees // 2
	for _ in range(2):
		agent._step(agent._oh, agent._h)
		agent.simulations += 1
	s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
	s0.wait_stream(torch.cuda.current_stream())
	s1.wait_stream(torch.cuda.current_stream())
	with torch.cuda.stream(s0):
		pa, va = agent._net_half(0, na)
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	keep = []
	for _ in range(sims - 2):
		with torch.cuda.stream(s0):
			agent._backup_half(0, na, pa, va)
		with torch.cuda.stream(s1):
			pb, vb = agent._net_half(na, T - na)
		s0.wait_stream(s1)
		s1.wait_stream(s0)
		with torch.cuda.stream(s0):
			pa, va = agent._net_half(0, na)
		with torch.cuda.stream(s1):
			agent._backup_half(na, T - na, pb, vb)
		s0.wait_stream(s1)
		s1.wait_stream(s0)
		keep = [pa, va, pb, vb]
	torch.cuda.synchronize()
	dt = time.perf_counter() - t0
	torch.cuda.current_stream().wait_stream(s0)
	agent.simulations = sims
	del keep
	return dt / (sims - 2) * 1e3


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--sims", type=int, default=1024)
	ap.add_argument("--trees", type=int, default=256)
	ap.add_argument("--trace-marks", action="store_true", help="a recognisable marker kernel (a goal test of three states) in front of every form's timed part, for the trace summary")
	a = ap.parse_args()
	_ffi.check(_ffi.lib().rk_init(0))
	net = FcSmall().cuda().eval().to(torch.bfloat16)
	T, sims = a.trees, a.sims
	starts = []
	for g in range(T):
		np.random.seed(g)
		starts.append(cube.scramble(14, True)[0])
	starts = np.array(starts)
	cap = 12 * sims + 64
	marker = torch.zeros((3, 20), dtype=torch.int8, device="cuda")        # the marker: a goal test of three states (no other launch of that kernel here)
	out = {"bench": "mcts_overlap", "trees": T, "sims": sims}
	for form in ("one_stream", "two_halves", "two_halves_eager"):
		agent = MCTSBatch(net, 0.6, T, capacity=cap, max_path=16384, fused_first_layer="folded", overlap_halves=form == "two_halves")
		if form == "two_halves_eager":
			agent.search(starts, max_states=cap, max_sims=8, use_graph=False)
			if a.trace_marks:
				cube.device.multi_is_solved(marker)
			out[form + "_ms_per_step"] = eager_two_halves(agent, starts, cap, sims)
			agent._finish()
		else:
			agent.search(starts, max_states=cap, max_sims=16, use_graph=True, poll=8)
			torch.cuda.synchronize()
			if a.trace_marks:
				cube.device.multi_is_solved(marker)
			t0 = time.perf_counter()
			agent.search(starts, max_states=cap, max_sims=sims, use_graph=True, poll=64)
			torch.cuda.synchronize()
			out[form + "_ms_per_step"] = (time.perf_counter() - t0) / agent.simulations * 1e3
		del agent
		torch.cuda.empty_cache()
	print(json.dumps(out), flush=True)


if __name__ == "__main__":
	main()
