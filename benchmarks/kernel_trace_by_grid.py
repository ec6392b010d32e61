"""
Per-launch-shape kernel durations from a rocprofv3 kernel trace: rocprofv3's own kernel_stats.csv averages a kernel NAME over every
launch of the process, whatever its size; a benchmark that runs one kernel at 256 states, 1 M and 12 M rows needs them apart.

    python benchmarks/kernel_trace_by_grid.py DIR [--min-calls 3] > profiles/r04_kernels_trace_by_grid.csv

DIR = output directory of `rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 benchmarks/kernels.py`.
One row per (kernel, grid size in work-items, workgroup size): launches, average / median / minimum duration in ns.
With --segments: one row per RUN of consecutive launches of the same kernel and shape (a benchmark's timing loop), in the order
of the trace -- the same kernel and size timed in several regimes (same buffer, rotating buffers, replayed from a graph) then
stay apart; the first `--skip` launches of a run (its warm-up) are left out of the statistics.
"""
import argparse
import csv
import glob
import os
import statistics
import sys

csv.field_size_limit(1 << 30)


def short(name: str) -> str:
	s = name[5:] if name.startswith("void ") else name
	depth = 0
	for i, ch in enumerate(s):
		depth += ch == "<"
		depth -= ch == ">"
		if ch == "(" and depth == 0:
			return s[:i]
	return s


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("dir")
	ap.add_argument("--min-calls", type=int, default=3)
	ap.add_argument("--only", default="rk::")
	ap.add_argument("--segments", action="store_true")
	ap.add_argument("--skip", type=int, default=0)
	a = ap.parse_args()
	hits = sorted(glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True))
	if not hits:
		sys.exit(f"no kernel_trace.csv under {a.dir}")
	groups = {}
	with open(hits[-1], newline="") as f:
		rows = sorted(csv.DictReader(f), key=lambda r: int(r["Start_Timestamp"]))
	w = csv.writer(sys.stdout)
	if a.segments:
		w.writerow(["Run", "Kernel", "Grid_Size", "Workgroup_Size", "Launches", "AverageNs", "MedianNs", "MinNs"])
		runs, cur = [], None
		for row in rows:
			grid = int(row.get("Grid_Size") or row.get("Grid_Size_X") or 0)
			wg = int(row.get("Workgroup_Size") or row.get("Workgroup_Size_X") or 0)
			key = (short(row["Kernel_Name"]), grid, wg)
			if cur is None or cur[0] != key:
				cur = (key, [])
				runs.append(cur)
			cur[1].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
		k = 0
		for (name, grid, wg), d in runs:
			d = d[a.skip:]
			if a.only in name and len(d) >= a.min_calls:
				k += 1
				w.writerow([k, name if len(name) < 120 else name[:117] + "...", grid, wg, len(d), round(statistics.fmean(d), 1), statistics.median(d), min(d)])
		return
	for row in rows:
		if a.only not in row["Kernel_Name"]:
			continue
		grid = int(row.get("Grid_Size") or row.get("Grid_Size_X") or 0)
		wg = int(row.get("Workgroup_Size") or row.get("Workgroup_Size_X") or 0)
		groups.setdefault((short(row["Kernel_Name"]), grid, wg), []).append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
	w.writerow(["Kernel", "Grid_Size", "Workgroup_Size", "Launches", "AverageNs", "MedianNs", "MinNs"])
	for (name, grid, wg), d in sorted(groups.items(), key=lambda kv: (kv[0][0], kv[0][1])):
		if len(d) >= a.min_calls:
			w.writerow([name if len(name) < 120 else name[:117] + "...", grid, wg, len(d), round(statistics.fmean(d), 1), statistics.median(d), min(d)])


if __name__ == "__main__":
	main()
