"""
Why is AStar(use_hipgraph=True) slower than eager launches at N = 700 (round-2 verdict, item 5)?  Runs a fixed number of
A* iterations with the exact one-kernel stub net in ONE mode, for rocprofv3 --kernel-trace; the summary mode then reads
both traces and splits the time per iteration into kernel-busy time and gaps between consecutive kernels.

    rocprofv3 --kernel-trace --output-format csv -d DIR_E -- python3 benchmarks/astar_graph_gaps.py run --mode eager
    rocprofv3 --kernel-trace --output-format csv -d DIR_G -- python3 benchmarks/astar_graph_gaps.py run --mode graph
    python benchmarks/astar_graph_gaps.py summary --eager DIR_E --graph DIR_G
"""
import argparse
import csv
import glob
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(args):
	import numpy as np
	import torch
	from benchmarks.nets import FastStub
	from librubiks_amd import cube
	from librubiks_amd.solving.agents import AStar
	np.random.seed(5)
	start, _, _ = cube.scramble(16, True)
	agent = AStar(FastStub(), 0.2, args.expansions, use_hipgraph=args.mode == "graph", poll=8)
	agent.search(start, None, 20 * 12 * args.expansions)                   # warm-up (allocator, first capture)
	torch.cuda.synchronize()
	import time
	t0 = time.perf_counter()
	agent.search(start, None, args.max_states)
	torch.cuda.synchronize()
	dt = time.perf_counter() - t0
	print(json.dumps({"mode": args.mode, "expansions": args.expansions, "iterations": agent.iterations, "states": len(agent),
	                  "us_per_iteration_wall": dt / agent.iterations * 1e6}), flush=True)


def load(directory):
	hits = sorted(glob.glob(os.path.join(directory, "**", "*kernel_trace.csv"), recursive=True))
	rows = []
	with open(hits[-1], newline="") as f:
		for r in csv.DictReader(f):
			rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
	rows.sort()
	return rows


def summary(args):
	out = {}
	for mode, d in (("eager", args.eager), ("graph", args.graph)):
		rows = load(d)
		# the timed search = the second half of the trace: take the last 60 % of the dispatches, whole iterations of k_end to k_end
		ends = [i for i, r in enumerate(rows) if "k_end" in r[2]]
		ends = ends[len(ends) * 2 // 5:]
		busy, gaps, per_kernel, spans = [], [], {}, []
		for a, b in zip(ends[:-1], ends[1:]):
			seg = rows[a + 1:b + 1]
			spans.append(seg[-1][1] - rows[a][1])
			busy.append(sum(e - s for s, e, _ in seg))
			gaps.append(sum(max(0, seg[i][0] - (rows[a][1] if i == 0 else seg[i - 1][1])) for i in range(len(seg))))
			for s, e, n in seg:
				key = n.split("(")[0][-60:]
				per_kernel.setdefault(key, []).append(e - s)
		out[mode] = {"iterations": len(spans), "kernels_per_iteration": statistics.fmean(len(rows[a + 1:b + 1]) for a, b in zip(ends[:-1], ends[1:])),
		             "us_per_iteration": statistics.fmean(spans) / 1e3, "us_kernels_busy": statistics.fmean(busy) / 1e3,
		             "us_gaps_between_kernels": statistics.fmean(gaps) / 1e3,
		             "us_per_kernel": {k: round(statistics.fmean(v) / 1e3 * len(v) / len(spans), 2) for k, v in sorted(per_kernel.items())}}
	print(json.dumps(out, indent=1))


if __name__ == "__main__":
	ap = argparse.ArgumentParser()
	sub = ap.add_subparsers(dest="cmd", required=True)
	r = sub.add_parser("run")
	r.add_argument("--mode", choices=["eager", "graph"], required=True)
	r.add_argument("--expansions", type=int, default=700)
	r.add_argument("--max-states", type=int, default=400_000)
	s = sub.add_parser("summary")
	s.add_argument("--eager", required=True)
	s.add_argument("--graph", required=True)
	a = ap.parse_args()
	(run if a.cmd == "run" else summary)(a)
