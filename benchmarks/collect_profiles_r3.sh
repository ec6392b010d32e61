#!/bin/bash
# gpurun_out/r3 and gpurun_out/r3diag (raw output of benchmarks/run_gpu_round3*.sh) -> the tracked records profiles/r03_* (see profiles/README.md)
cd "$(dirname "$0")/.."
O=gpurun_out/r3
D=gpurun_out/r3diag
P=profiles
if [ -f $O/bench.log ]; then
	cp $O/pytest_gpu.log $P/r03_pytest_gpu.log
	tail -1 $O/bench.log > $P/r03_bench.json
	grep '^{' $O/bench_prof.log | tail -1 > $P/r03_bench_under_rocprof.json
	grep '^{' $O/bench_2ranks_gloo.log | tail -1 > $P/r03_bench_2ranks_gloo_rehearsal.json
	cp $O/expand12_pmc.json $P/r03_expand12_pmc.json; cp $O/expand12_kernel_stats.csv $P/r03_expand12_kernel_stats.csv
	grep '^{"kernel": "k_expand12p"' $O/pmc_summary.log > $P/r03_expand12_trace_summary.json
	for f in sizes sizes_unpaced kernels kernels686; do [ -s $O/$f.json ] && cp $O/$f.json $P/r03_$f.json; done
fi
if [ -f $O/search.json ]; then
	for f in astar_small search astar_batch; do [ -s $O/$f.json ] && cp $O/$f.json $P/r03_$f.json; done
	[ -s $O/sharded.json ] && cp $O/sharded.json $P/r03_sharded_rehearsal.json
	python benchmarks/pmc_summary.py --stats $O/prof_mcts --kernel k_mcts_backup_select --out-stats $P/r03_mcts4096_bf16_folded_kernel_stats.csv > /dev/null
	python benchmarks/pmc_summary.py --stats $O/prof_astar100 --kernel k_queue_insert --out-stats $P/r03_astar100_kernel_stats.csv > /dev/null
fi
if [ -f $D/regimes_pmc.json ]; then
	cp $D/regimes_plain.json $P/r03_regimes.json; cp $D/regimes_pmc.json $P/r03_regimes_pmc.json; cp $D/layout_ab.json $P/r03_layout_ab.json
	cp $D/gaps_summary.json $P/r03_astar_graph_gaps.json
	python - $D $P <<'PY'
import csv, glob, json, sys
D, P = sys.argv[1:3]
def strip(paths, out):
	rows = []
	for f in paths:
		for l in open(f):
			if l.startswith("{"):
				d = json.loads(l); d.pop("ms", None); rows.append(d)
	rows.sort(key=lambda d: (d["parents"], d["id"], str(d["grid_blocks"])))
	with open(out, "w") as o:
		for d in rows: o.write(json.dumps(d) + "\n")
strip([f"{D}/tune_1m.json", f"{D}/tune_16m.json"], f"{P}/r03_tune_expand.json")
strip(sorted(glob.glob(f"{D}/tune_[0-9]*.json")), f"{P}/r03_tune_sizes.json")
with open(f"{P}/r03_queue_insert_grid.json", "w") as o:
	for g in (8, 128, 512):
		f = sorted(glob.glob(f"{D}/prof_ins_{g}/**/*kernel_stats.csv", recursive=True))[-1]
		for r in csv.reader(open(f)):
			if "k_queue_insert" in r[0]:
				o.write(json.dumps({"kernel": "rk::k_queue_insert<false>", "workload": "A* N = 100, stub net, depth-16 scramble, 200 k states (benchmarks/astar_profile.py under rocprofv3 --kernel-trace --stats)",
				                    "min_workgroups": g, "calls": int(r[1]), "avg_us": float(r[3]) / 1e3, "min_us": float(r[5]) / 1e3, "max_us": float(r[6]) / 1e3}) + "\n")
PY
fi
git status --short $P | head -40
