"""
Reads the rocprofv3 kernel trace of `benchmarks/mcts_overlap.py --trace-marks` and answers VERDICT r4 #3's question with the
timestamps: per form (a marker launch of rk::k_multi_is_solved precedes every form's timed part), how long the backup + select kernels ran,
how much of that time lay UNDER another kernel (any kernel of another stream running at the same time), the busy time of the
device per simulation step, and the average duration of the net's kernels at full and at half batch size.

    python benchmarks/mcts_overlap_summary.py DIR > profiles/r05_mcts_overlap_trace.json
"""
import csv
import glob
import json
import os
import sys

csv.field_size_limit(1 << 30)


def main():
	d = sys.argv[1]
	hits = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
	if not hits:
		sys.exit(f"no kernel_trace.csv under {d}")
	with open(hits[-1], newline="") as f:
		rows = sorted(csv.DictReader(f), key=lambda r: int(r["Start_Timestamp"]))
	# split at the marker: rk::k_multi_is_solved is launched only by mcts_overlap.py, once in front of every form's timed part
	forms, cur = [], []
	names = ["(before the first marker)", "one_stream", "two_halves", "two_halves_eager"]
	seen_mcts = False                                                    # (the scrambles' own goal tests come before any MCTS kernel)
	for r in rows:
		seen_mcts = seen_mcts or "k_mcts_" in r["Kernel_Name"]
		if seen_mcts and "k_multi_is_solved" in r["Kernel_Name"]:
			forms.append(cur)
			cur = []
			continue
		cur.append(r)
	forms.append(cur)
	out = {"trace": os.path.basename(hits[-1]), "forms": {}}
	for name, ks in zip(names, forms):
		if name.startswith("(") or not ks:
			continue
		iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in ks]
		sel = [(s, e) for s, e, n in iv if "k_mcts_backup_select" in n]
		other = sorted((s, e) for s, e, n in iv if "k_mcts_backup_select" not in n)
		# up to the next form's set-up: stop at the first gap of more than 20 ms between select launches
		for i in range(1, len(sel)):
			if sel[i][0] - sel[i - 1][1] > 20_000_000:
				sel = sel[:i]
				break
		if not sel:
			continue
		lo, hi = sel[0][0], sel[-1][1]
		under = 0
		j = 0
		for s, e in sel:
			while j < len(other) and other[j][1] <= s:
				j += 1
			k = j
			covered, at = 0, s
			while k < len(other) and other[k][0] < e:
				a, b = max(other[k][0], at), min(other[k][1], e)
				if b > a:
					covered += b - a
					at = b
				k += 1
			under += covered
		total_sel = sum(e - s for s, e in sel)
		# busy time of the device: union of all kernel intervals inside [lo, hi]
		allv = sorted((s, e) for s, e, n in iv if e > lo and s < hi)
		busy, end = 0, lo
		for s, e in allv:
			s = max(s, end)
			if e > s:
				busy += e - s
				end = e
		per = {}
		for s, e, n in iv:
			if s < lo or e > hi:
				continue
			key = n.split("(")[0][:90]
			per.setdefault(key, []).append(e - s)
		top = sorted(((sum(v), k, len(v)) for k, v in per.items()), reverse=True)[:8]
		launches_per_step = 2 if name != "one_stream" else 1
		steps = len(sel) / launches_per_step
		out["forms"][name] = {
			"select_launches": len(sel), "steps": steps, "select_us_avg": total_sel / len(sel) / 1e3,
			"select_time_under_another_kernel_frac": under / total_sel,
			"wall_us_per_step": (hi - lo) / steps / 1e3, "device_busy_us_per_step": busy / steps / 1e3,
			"kernel_time_sum_us_per_step": sum(e - s for s, e, n in iv if s >= lo and e <= hi) / steps / 1e3,
			"top_kernels_us_avg_and_calls_per_step": [{"kernel": k, "avg_us": t / c / 1e3, "calls_per_step": c / steps} for t, k, c in top],
		}
	print(json.dumps(out, indent=1))


if __name__ == "__main__":
	main()
