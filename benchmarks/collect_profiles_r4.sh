#!/bin/bash
# gpurun_out/r4/<stage> (raw output of benchmarks/run_gpu_round4.sh <stage>) -> the tracked records profiles/r04_* (see profiles/README.md)
cd "$(dirname "$0")/.."
O=gpurun_out/r4
P=profiles
last() { grep '^{' "$1" | tail -1; }
[ -f $O/tests/pytest_gpu.log ] && cp $O/tests/pytest_gpu.log $P/r04_pytest_gpu.log
if [ -f $O/bench/bench.log ]; then
	last $O/bench/bench.log > $P/r04_bench.json
	last $O/bench/bench_prof.log > $P/r04_bench_under_rocprof.json
	last $O/bench/bench_2ranks_gloo.log > $P/r04_bench_2ranks_self_spawned_gloo.json
	cp $O/bench/expand12_pmc.json $P/r04_expand12_pmc.json; cp $O/bench/expand12_kernel_stats.csv $P/r04_expand12_kernel_stats.csv
	grep '^{"kernel": "k_expand12p"' $O/bench/pmc_summary.log > $P/r04_expand12_trace_summary.json
	cp $O/bench/search_legs.json $P/r04_search_legs.json; cp $O/bench/search_legs_kernel_stats.csv $P/r04_search_legs_kernel_stats.csv
fi
if [ -f $O/kernels/kernels.json ]; then
	cp $O/kernels/kernels.json $P/r04_kernels.json; cp $O/kernels/kernels686.json $P/r04_kernels686.json
	python benchmarks/kernel_trace_by_grid.py $O/kernels/prof_kernels --segments --min-calls 8 --skip 2 > $P/r04_kernels_trace_runs.csv
	python benchmarks/kernel_trace_by_grid.py $O/kernels/prof_kernels686 --segments --min-calls 8 --skip 2 > $P/r04_kernels686_trace_runs.csv
	cp $O/kernels/paced_pmc.json $P/r04_paced_pmc.json
fi
if [ -f $O/rows/rows_fit_rocprof.json ]; then
	cp $O/rows/rows_trace_runs.csv $P/r04_rows_trace_runs.csv
	cp $O/rows/rows_fit_rocprof.json $P/r04_rows_fit_rocprof.json; grep '^{' $O/rows/rows_fit_events.json > $P/r04_rows_fit_events.json
fi
if [ -f $O/streams/pace_streams.json ]; then
	cp $O/streams/pace_streams.json $P/r04_pace_streams.json; cp $O/streams/pace_streams_overlapping.json $P/r04_pace_streams_overlapping.json
fi
if [ -d $O/astar/n100 ]; then
	python benchmarks/astar_floor.py --table $O/astar --out $P/r04_astar_floor.json > /dev/null
fi
if [ -f $O/search/astar_small.json ]; then
	cp $O/search/astar_small.json $P/r04_astar_small.json
	[ -s $O/search/grow_cost.json ] && cp $O/search/grow_cost.json $P/r04_grow_cost.json
fi
[ -s $O/sharded/rehearsal.json ] && cp $O/sharded/rehearsal.json $P/r04_sharded_rehearsal.json
[ -s $O/mcts_priors_cost.json ] && cp $O/mcts_priors_cost.json $P/r04_mcts_priors_cost.json
[ -s $O/phase/sweep.json ] && cp $O/phase/sweep.json $P/r04_phase_sweep.json
git status --short $P | head -40
