"""
The latency floor of one A* iteration, kernel by kernel ("a per-kernel floor table ... that closes the topic").

    rocprofv3 --kernel-trace --stats --output-format csv -d DIR/n<N> -- python3 benchmarks/astar_floor.py --n <N>     (N = 1, 10, 100, 700)
    python benchmarks/astar_floor.py --table DIR --out profiles/r04_astar_floor.json

One search per process with the exact stub net as ONE kernel (benchmarks/nets.py FastStub), iteration replayed as a hipGraph
(so that the host is out of the picture), depth-16 scramble, 200 000 states.  With N = 1 an iteration moves 12 children: every
kernel then costs its launch plus its chain of DEPENDENT memory round trips and barriers and nothing else -- that is the floor of
the six-launch structure; what N = 100 and N = 700 add on top is the work that scales.  The table holds rocprofv3's average
duration of every engine kernel at each N, their sum, the stub's kernels, and the event-timed microseconds per iteration.
"""
import argparse
import csv
import glob
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
csv.field_size_limit(1 << 30)
ENGINE = ("k_expand_lookup", "k_append<", "k_new_rows<", "k_records_sort<", "k_merge_pass", "k_queue_insert<", "k_end<", "k_pop_wide")


def run(n):
	import numpy as np
	import torch
	from benchmarks.nets import FastStub
	from librubiks_amd import cube
	from librubiks_amd.solving.agents import AStar
	agent = AStar(FastStub(), 0.2, n, poll=64, use_hipgraph=True)
	np.random.seed(3)
	state, _, _ = cube.scramble(16, True)
	budget = 200_000 if n >= 10 else 30_000
	agent.search(state, None, 3000 + 12 * n)
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	agent.search(state, None, budget)
	torch.cuda.synchronize()
	dt = time.perf_counter() - t0
	print(json.dumps({"N": n, "iterations": agent.iterations, "states": len(agent), "us_per_iteration": dt / max(agent.iterations, 1) * 1e6}), flush=True)


def table(root, out):
	rec = {"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 benchmarks/astar_floor.py --n N  (one process per N)", "N": {}}
	for d in sorted((p for p in glob.glob(os.path.join(root, "n*")) if os.path.isdir(p)), key=lambda p: int(os.path.basename(p)[1:])):
		n = int(os.path.basename(d)[1:])
		hits = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))
		if not hits:
			continue
		rows = list(csv.DictReader(open(hits[-1], newline="")))
		eng, other = {}, {}
		iters = max((int(r["Calls"]) for r in rows if "k_end<" in r["Name"]), default=1)
		for r in rows:
			name = r["Name"]
			per_iter = float(r["TotalDurationNs"]) / iters / 1e3
			if "rk::" in name and any(k in name for k in ENGINE):
				eng[name.split("rk::", 1)[1].split("(", 1)[0]] = round(per_iter, 2)
			elif per_iter > 0.3:
				other[(name[:60])] = round(per_iter, 2)
		line = None
		log = os.path.join(root, f"n{n}.log")
		if os.path.exists(log):
			for l in open(log):
				if l.startswith("{"):
					line = json.loads(l)
		rec["N"][str(n)] = {"iterations_traced": iters, "engine_kernels_us_per_iteration": eng, "engine_sum_us": round(sum(eng.values()), 2),
		                    "other_kernels_us_per_iteration": other, "us_per_iteration_wall_under_rocprof": line and line["us_per_iteration"]}
	with open(out, "w") as f:
		json.dump(rec, f, indent=1)
	print(json.dumps(rec))


if __name__ == "__main__":
	ap = argparse.ArgumentParser()
	ap.add_argument("--n", type=int, default=100)
	ap.add_argument("--table")
	ap.add_argument("--out", default="astar_floor.json")
	a = ap.parse_args()
	if a.table:
		table(a.table, a.out)
	else:
		run(a.n)
