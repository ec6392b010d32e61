"""
Search-loop benchmarks on one MI355X (BASELINE.json configs[2] and configs[3]); random-init fc_small net.

    python benchmarks/search.py astar [--games 5] [--depth 14] [--expansions 1000] [--max-states 150000]
    python benchmarks/search.py mcts  [--trees 256] [--sims 4096] [--graph 1]

Prints one JSON object per benchmark.
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
from librubiks_amd.solving.agents import AStar, MCTSBatch  # noqa: E402


class TimedNet(torch.nn.Module):
	"""Wraps the net to separate its share of the wall time (adds a sync per call: use only for the breakdown)."""
	def __init__(self, net):
		super().__init__()
		self.net, self.seconds, self.rows = net, 0.0, 0

	def forward(self, x, policy=True, value=True):
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		out = self.net(x, policy=policy, value=value)
		torch.cuda.synchronize()
		self.seconds += time.perf_counter() - t0
		self.rows += len(x)
		return out


FUSED = {0: False, 1: True, 2: "epilogue", 3: "folded"}
FUSED_NOTE = {0: "", 1: ", first layer fused (rk_ohl)", 2: ", first layer fused with ELU + BatchNorm epilogue (rk_ohl)",
              3: ", first layer fused with ELU + BatchNorm epilogue (rk_ohl), other BatchNorm layers folded"}

def bench_astar(args):
	net = FcSmall().cuda().eval()
	if args.bf16:
		net = net.to(torch.bfloat16)          # the agent then asks the engine for a bf16 one-hot (exact 0/1) directly
	rows = []
	for timed in ((False,) if args.fused else (False, True)):
		use = TimedNet(net) if timed else net
		agent = AStar(use, args.lam, args.expansions, fused_first_layer=FUSED[args.fused])
		np.random.seed(12345)
		agent.search(cube.scramble(args.depth, True)[0], time_limit=None, max_states=min(args.max_states, 40 * args.expansions))   # warm-up: engine, allocator, GEMM selection
		if timed:
			use.seconds, use.rows = 0.0, 0
		tot_t = tot_states = tot_iter = solved = 0
		for g in range(args.games):
			np.random.seed(g)
			state, _, _ = cube.scramble(args.depth, True)
			torch.cuda.synchronize()
			t0 = time.perf_counter()
			ok = agent.search(state, time_limit=None, max_states=args.max_states)
			torch.cuda.synchronize()
			tot_t += time.perf_counter() - t0
			tot_states += len(agent)
			tot_iter += agent.iterations
			solved += ok
		row = {"bench": "astar", "config": f"depth-{args.depth} scrambles, lambda={args.lam}, N={args.expansions}, max_states={args.max_states}, fc_small random init"
		       + (" bf16" if args.bf16 else " fp32") + FUSED_NOTE[args.fused], "games": args.games, "solved": solved, "states": tot_states, "iterations": tot_iter,
		       "seconds": tot_t, "states_per_s": tot_states / tot_t, "expansions_per_s": tot_iter * args.expansions / tot_t,
		       "ms_per_iteration": tot_t / max(tot_iter, 1) * 1e3}
		if timed:
			row["net_seconds"] = use.seconds
			row["net_share"] = use.seconds / tot_t
			row["note"] = "breakdown run (sync around every net call)"
		rows.append(row)
		print(json.dumps(row), flush=True)
	return rows


def bench_astar_batch(args):
	"""Many scrambles at once (the evaluation workload): AStarBatch against the same searches run one after the other."""
	from librubiks_amd.solving.agents import AStarBatch
	net = FcSmall().cuda().eval()
	if args.bf16:
		net = net.to(torch.bfloat16)
	S = args.searches
	starts = []
	for g in range(S):
		np.random.seed(g)
		starts.append(cube.scramble(args.depth, True)[0])
	starts = np.array(starts)
	out = {}
	for mode in ("sequential", "batch", "batch+graph"):
		# engines are created and run once before the clock starts (pool allocation, GEMM selection, first capture)
		if mode == "sequential":
			agent = AStar(net, args.lam, args.expansions, fused_first_layer=FUSED[args.fused])
			agent.search(starts[0], None, 5 * 12 * args.expansions)
		else:
			agent = AStarBatch(net, args.lam, args.expansions, S, capacity=args.max_states, fused_first_layer=FUSED[args.fused])
			if args.slice >= 0:
				agent.net_slice_rows = args.slice or 10 ** 9                 # 0: the whole padded batch in one forward (round 2's behaviour)
			agent.search(starts, max_states=5 * 12 * args.expansions, use_graph=mode.endswith("graph"), poll=2)
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		if mode == "sequential":
			states = iters = solved = 0
			for g in range(min(S, args.sequential_games)):
				solved += agent.search(starts[g], None, args.max_states)
				states += len(agent)
				iters += agent.iterations
			scale = S / min(S, args.sequential_games)
			states, iters, solved = states * scale, iters * scale, solved * scale
			torch.cuda.synchronize()
			dt = (time.perf_counter() - t0) * scale
		else:
			res = agent.search(starts, max_states=args.max_states, use_graph=mode.endswith("graph"), poll=args.poll)
			torch.cuda.synchronize()
			dt = time.perf_counter() - t0
			states, iters, solved = int(agent.status[:, 2].sum()), int(agent.status[:, 3].sum()), int(res.sum())
		out[mode] = {"seconds": dt, "states": int(states), "states_per_s": states / dt, "search_iterations": int(iters), "solved": int(solved)}
	row = {"bench": "astar_batch", "config": f"{S} depth-{args.depth} scrambles, lambda={args.lam}, N={args.expansions}, max_states={args.max_states} each, "
	       f"fc_small random init {'bf16' if args.bf16 else 'fp32'}" + FUSED_NOTE[args.fused]
	       + (f", net forwards of at most {args.slice} rows" if args.slice > 0 else ", ONE net forward on the whole padded batch" if args.slice == 0 else ", net forwards of whole searches (the agent's default: one search per forward from K = 4096 rows on, else up to 16384 rows)"), **out,
	       "modes": "sequential = AStar, one search after the other; batch = AStarBatch, eager, net on the searches' NEW rows compacted on the device "
	                "(exact_batch, the default for a real net); batch+graph = AStarBatch on the padded batch, iteration replayed as a hipGraph",
	       "speedup_batch_vs_sequential": out["sequential"]["seconds"] / out["batch"]["seconds"],
	       "speedup_batch_graph_vs_sequential": out["sequential"]["seconds"] / out["batch+graph"]["seconds"]}
	print(json.dumps(row), flush=True)
	return row


def bench_mcts(args):
	net = FcSmall().cuda().eval()
	if args.bf16:
		net = net.to(torch.bfloat16)          # bf16 one-hot straight from the engine; P, V come back as float32 -> float64 statistics
	T = args.trees
	starts = []
	for g in range(T):
		np.random.seed(g)
		s, _, _ = cube.scramble(args.depth, True)
		starts.append(s)
	starts = np.array(starts)
	cap = args.sims * 12 + 64
	agent = MCTSBatch(net, args.c, T, capacity=cap, max_path=args.max_path, fused_first_layer=FUSED[args.fused])
	# one-time costs stay out of the timed search, as in bench_astar: pool allocation (6 GB), GEMM kernel selection, first capture
	agent.search(starts, max_states=cap, max_sims=16, use_graph=bool(args.graph), poll=8)
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	solved = agent.search(starts, max_states=cap, max_sims=args.sims, use_graph=bool(args.graph), poll=args.poll)
	torch.cuda.synchronize()
	dt = time.perf_counter() - t0
	st = agent.status
	row = {"bench": "mcts", "config": f"{T} trees x {args.sims} sims, depth-{args.depth} scrambles, c={args.c}, fc_small random init {'bf16' if args.bf16 else 'fp32'}, "
	       f"hipGraph={'on' if args.graph else 'off'}" + FUSED_NOTE[args.fused], "seconds": dt, "tree_sims": int(st[:, 3].sum()), "tree_sims_per_s": float(st[:, 3].sum()) / dt,
	       "steps": agent.simulations, "ms_per_step": dt / agent.simulations * 1e3, "solved": int(solved.sum()), "states": int(st[:, 2].sum()),
	       "states_per_s": float(st[:, 2].sum()) / dt, "max_path_len": int(st[:, 4].max())}
	print(json.dumps(row), flush=True)
	return row


if __name__ == "__main__":
	ap = argparse.ArgumentParser()
	ap.add_argument("what", choices=["astar", "mcts", "astar_batch"])
	ap.add_argument("--searches", type=int, default=64, help="upper bound of iterations between two polls of the engines")
	ap.add_argument("--sequential-games", type=int, default=16)
	ap.add_argument("--games", type=int, default=5)
	ap.add_argument("--depth", type=int, default=14)
	ap.add_argument("--expansions", type=int, default=1000)
	ap.add_argument("--max-states", type=int, default=150_000)
	ap.add_argument("--lam", type=float, default=0.16)
	ap.add_argument("--bf16", type=int, default=0)
	ap.add_argument("--slice", type=int, default=-1, help="astar_batch: rows per net forward (0 = whole padded batch, -1 = the agent's default)")
	ap.add_argument("--fused", type=int, default=0, help="1: first Linear reads the 20-byte states (rk_ohl_*), no one-hot batch; "
	                "2: and its ELU + BatchNorm run in the kernel's epilogue; 3: and the other BatchNorm layers are folded into the next Linear")
	ap.add_argument("--trees", type=int, default=256)
	ap.add_argument("--sims", type=int, default=4096)
	ap.add_argument("--c", type=float, default=0.6)
	ap.add_argument("--graph", type=int, default=1)
	ap.add_argument("--poll", type=int, default=64, help="upper bound of iterations between two polls of the engines")
	ap.add_argument("--max-path", type=int, default=16384)
	a = ap.parse_args()
	_ffi.check(_ffi.lib().rk_init(0))
	{"astar": bench_astar, "mcts": bench_mcts, "astar_batch": bench_astar_batch}[a.what](a)
