"""
The rocprofv3 record of bench.py's search legs ("profiles/r04_search* rocprof stats of the same legs agree").

    python benchmarks/search_legs_summary.py --stats DIR --bench-log FILE --out profiles/r04_search_legs.json [--out-stats CSV]

  --stats      directory of `rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 bench.py ...`
  --bench-log  stdout of that very run (its JSON line carries the event-based figures of the same process)
From the kernel statistics of the whole process:
  A*    engine us per iteration = total time of the engine's kernels (expand + lookup, append, rows, sort, merge, insert, end,
        wide pop) / launches of k_end<false> (one per iteration, warm-up and both passes included);
  MCTS  average duration of the backup + select (+ expand ahead) kernel over all its launches (warm-up, replayed and eager run).
and beside them the bench line's own `astar_engine_us_per_iteration` (HIP events around the two engine parts of an iteration:
the kernels plus the gaps between them) and `mcts_select_us` (HIP event pair around the launch).
"""
import argparse
import csv
import glob
import json
import os
import sys

csv.field_size_limit(1 << 30)
ASTAR = ("k_expand_lookup", "k_append<", "k_new_rows<", "k_records_sort<", "k_merge_pass", "k_queue_insert<", "k_end<", "k_pop_wide")


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--stats", required=True)
	ap.add_argument("--bench-log", required=True)
	ap.add_argument("--out", required=True)
	ap.add_argument("--out-stats")
	a = ap.parse_args()
	hits = sorted(glob.glob(os.path.join(a.stats, "**", "*kernel_stats.csv"), recursive=True))
	if not hits:
		sys.exit(f"no kernel_stats.csv under {a.stats}")
	with open(hits[-1], newline="") as f:
		rows = list(csv.DictReader(f))
	line = None
	for l in open(a.bench_log):
		if l.startswith("{"):
			line = json.loads(l)
	astar = {}
	for r in rows:
		name = r["Name"]
		if "rk::" in name and any(k in name for k in ASTAR) and "kb_" not in name:
			short = name.split("rk::", 1)[1].split("(", 1)[0]
			astar[short] = {"calls": int(r["Calls"]), "total_ns": int(float(r["TotalDurationNs"])), "avg_ns": float(r["AverageNs"])}
	iters = max((v["calls"] for k, v in astar.items() if k.startswith("k_end<")), default=0)
	engine_us = sum(v["total_ns"] for v in astar.values()) / max(iters, 1) / 1e3
	sel = [r for r in rows if "k_mcts_backup_select" in r["Name"]]
	sel_calls = sum(int(r["Calls"]) for r in sel)
	sel_us = sum(float(r["TotalDurationNs"]) for r in sel) / max(sel_calls, 1) / 1e3
	# round 5: the process also runs the step that advances the batch as two halves (launches over half the trees); the figure that
	# belongs beside the bench line's `mcts_select_us` (an event pair around the launch over ALL trees) is the average of the
	# full-batch launches only -- taken from the kernel trace by grid size (the largest grid the kernel was launched with)
	half_calls = half_us = None
	traces = sorted(glob.glob(os.path.join(a.stats, "**", "*kernel_trace.csv"), recursive=True))
	if traces:
		by_grid = {}
		with open(traces[-1], newline="") as f:
			for r in csv.DictReader(f):
				if "k_mcts_backup_select" in r["Kernel_Name"]:
					g = int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)
					by_grid.setdefault(g, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
		if by_grid:
			full = max(by_grid)
			sel_calls, sel_us = len(by_grid[full]), sum(by_grid[full]) / len(by_grid[full]) / 1e3
			rest = [d for g, v in by_grid.items() if g != full for d in v]
			if rest:
				half_calls, half_us = len(rest), sum(rest) / len(rest) / 1e3
	rec = {
		"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline",
		"astar_iterations_in_process": iters, "astar_engine_us_per_iteration_rocprof": engine_us, "astar_kernels": astar,
		"mcts_backup_select_launches": sel_calls, "mcts_select_us_rocprof": sel_us,
		"mcts_backup_select_half_batch_launches": half_calls, "mcts_select_us_rocprof_half_batch": half_us,
		"bench_line": {k: line.get(k) for k in ("event_pair_overhead_us", "astar_engine_us_per_iteration", "astar_engine_us_per_iteration_less_event_overhead",
		                                         "astar_net_share", "astar_ms_per_iteration", "astar_states_per_s", "mcts_select_us",
		                                         "mcts_select_us_less_event_overhead", "mcts_ms_per_step", "mcts_tree_sims_per_s")} if line else None,
	}
	if line and line.get("astar_engine_us_per_iteration"):
		rec["astar_engine_events_over_rocprof"] = line["astar_engine_us_per_iteration"] / max(engine_us, 1e-9)
		rec["mcts_select_events_over_rocprof"] = line["mcts_select_us"] / max(sel_us, 1e-9)
		if line.get("event_pair_overhead_us") is not None:
			rec["astar_engine_events_less_overhead_over_rocprof"] = line["astar_engine_us_per_iteration_less_event_overhead"] / max(engine_us, 1e-9)
			rec["mcts_select_events_less_overhead_over_rocprof"] = line["mcts_select_us_less_event_overhead"] / max(sel_us, 1e-9)
		rec["reading"] = ("the event figures bracket whole calls (several kernels and the gaps between them, under rocprofv3's own per-launch overhead in this run); "
		                  "the rocprof figures are kernel time only")
	with open(a.out, "w") as f:
		json.dump(rec, f, indent=1)
	if a.out_stats:
		with open(a.out_stats, "w", newline="") as f:
			w = csv.writer(f)
			w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
			for r in rows[:40]:
				n = r["Name"]
				w.writerow([n if len(n) < 160 else n[:157] + "..."] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")])
	print(json.dumps({k: v for k, v in rec.items() if k != "astar_kernels"}))


if __name__ == "__main__":
	main()
