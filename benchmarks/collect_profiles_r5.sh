#!/bin/bash
# gpurun_out/r5/<stage> (raw output of benchmarks/run_gpu_round5.sh <stage>) -> the tracked records profiles/r05_* (see profiles/README.md)
cd "$(dirname "$0")/.."
O=gpurun_out/r5
P=profiles
last() { grep '^{' "$1" | tail -1; }
[ -f $O/tests/pytest_gpu.log ] && cp $O/tests/pytest_gpu.log $P/r05_pytest_gpu.log
if [ -f $O/bench/bench.log ]; then
	last $O/bench/bench.log > $P/r05_bench.json
	last $O/bench/bench_prof.log > $P/r05_bench_under_rocprof.json
	cp $O/bench/expand12_pmc.json $P/r05_expand12_pmc.json; cp $O/bench/expand12_kernel_stats.csv $P/r05_expand12_kernel_stats.csv
	grep '^{"kernel": "k_expand12p"' $O/bench/pmc_summary.log > $P/r05_expand12_trace_summary.json
	cp $O/bench/search_legs.json $P/r05_search_legs.json; cp $O/bench/search_legs_kernel_stats.csv $P/r05_search_legs_kernel_stats.csv
	[ -s $O/bench/adi_cube.json ] && cp $O/bench/adi_cube.json $P/r05_adi_cube.json     # (r05_adi_cube_kernels.csv holds BOTH versions of the walk: assembled by hand from two runs, not overwritten here)
fi
if [ -f $O/sharded/w1_graph.json ]; then
	{ for f in w1_eager w1_graph w1_eager_stub w1_graph_stub; do echo "{\"run\": \"$f\", \"record\": $(last $O/sharded/$f.json)}"; done; } > $P/r05_sharded_rehearsal.json
	last $O/sharded/bench_2ranks_gloo.log > $P/r05_bench_2ranks_gloo.json
	grep -a "^nccl\|^rk_comm\|passed\|failed" $O/sharded/captured_collectives.log > $P/r05_captured_collectives.txt
fi
if [ -f $O/mcts/mcts_overlap.json ]; then
	cp $O/mcts/mcts_overlap.json $P/r05_mcts_overlap.json
fi
python benchmarks/profiles_index.py > $P/INDEX.md
git status --short $P | head -40
