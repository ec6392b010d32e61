#!/bin/bash
# Round-5 GPU work on one MI355X box, in stages (one gpurun call each):
#   gpurun --timeout 1100 -- 'RK_COMMIT=<sha> bash benchmarks/run_gpu_round5.sh <stage>'
# stages: tests | bench | sharded | mcts   (profiles/README.md maps the records to these commands)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
STAGE=${1:-tests}
O=gpurun_out/r5/$STAGE
mkdir -p $O
# a step that was killed (timeout, fault) ends the script: no further GPU work behind it
step() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -ge 124 ]; then echo "STOP: '$*' ended with $rc" | tee -a $O/stop.log; exit $rc; fi; return $rc; }
case $STAGE in
tests)
	step 1050 python -m pytest tests -m gpu -x -q --durations=12 ${RK_PYTEST_ARGS:-} > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log; tail -25 $O/pytest_gpu.log
	grep -q "exit 0" $O/pytest_gpu.log || exit 1
	step 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
	;;
bench)
	step 400 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-600
	step 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/bench_prof.log 2>&1
	step 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-search-legs > $O/pmc_f.log 2>&1
	step 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-search-legs > $O/pmc_w.log 2>&1
	python benchmarks/pmc_summary.py --stats $O/prof_stats --fetch $O/pmc_f --write $O/pmc_w --kernel k_expand12p --commit "${RK_COMMIT:-unknown}" \
		--command "python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-search-legs" --out-pmc $O/expand12_pmc.json --out-stats $O/expand12_kernel_stats.csv > $O/pmc_summary.log 2>&1; cut -c1-400 $O/pmc_summary.log
	python benchmarks/search_legs_summary.py --stats $O/prof_stats --bench-log $O/bench_prof.log --out $O/search_legs.json --out-stats $O/search_legs_kernel_stats.csv > $O/search_legs.log 2>&1; cut -c1-600 $O/search_legs.log
	find $O -name "*kernel_trace.csv" -size +40M -delete
	step 200 python benchmarks/adi_cube.py > $O/adi_cube.json 2> $O/adi_cube.err; cat $O/adi_cube.json
	step 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_adi -- python3 benchmarks/adi_cube.py > $O/adi_cube_prof.log 2>&1
	python benchmarks/kernel_trace_by_grid.py $O/prof_adi --min-calls 50 > $O/adi_cube_kernels.csv; cat $O/adi_cube_kernels.csv
	;;
sharded)
	# configs[4] on the one-GPU box: world 1 eager against the captured iteration (the phase split rides in the same object), then the
	# whole multi-GPU bench line with two ranks sharing the GPU over gloo (host-staged collectives: a rehearsal of the code path, not a number)
	RK_SHARD_GRAPH=0 step 300 python benchmarks/sharded.py --net fc_small_bf16 --games 3 --max-states 2000000 2> $O/w1_eager.err | grep '^{' > $O/w1_eager.json; cut -c1-900 $O/w1_eager.json
	RK_SHARD_GRAPH=1 step 300 python benchmarks/sharded.py --net fc_small_bf16 --games 3 --max-states 2000000 2> $O/w1_graph.err | grep '^{' > $O/w1_graph.json; cut -c1-900 $O/w1_graph.json
	RK_SHARD_GRAPH=1 step 300 python benchmarks/sharded.py --net stub --games 3 --max-states 2000000 2> $O/w1_graph_stub.err | grep '^{' > $O/w1_graph_stub.json; cut -c1-600 $O/w1_graph_stub.json
	RK_SHARD_GRAPH=0 step 300 python benchmarks/sharded.py --net stub --games 3 --max-states 2000000 2> $O/w1_eager_stub.err | grep '^{' > $O/w1_eager_stub.json; cut -c1-600 $O/w1_eager_stub.json
	RK_BENCH_BACKEND=gloo step 500 python bench.py --gpus 2 --steps 50 --warmup 5 > $O/bench_2ranks_gloo.log 2>&1; tail -1 $O/bench_2ranks_gloo.log | cut -c1-3000
	step 400 python -m pytest tests/test_sharded_gpu.py -q -m gpu -s -k "captured_iteration_with_device_collectives" > $O/captured_collectives.log 2>&1; grep -a "nccl\|rk_comm\|passed\|failed" $O/captured_collectives.log | cut -c1-1200
	;;
mcts)
	step 300 python benchmarks/mcts_overlap.py --sims 1024 > $O/mcts_overlap.json 2> $O/mcts_overlap.err; cat $O/mcts_overlap.json
	step 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 benchmarks/mcts_overlap.py --sims 200 --trace-marks > $O/traced.json 2> $O/traced.err
	python benchmarks/mcts_overlap_summary.py $O/trace > $O/mcts_overlap_trace.json; cut -c1-400 $O/mcts_overlap_trace.json
	;;
*) echo "unknown stage $STAGE"; exit 2;;
esac
