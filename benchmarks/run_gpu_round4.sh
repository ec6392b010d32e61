#!/bin/bash
# Round-4 GPU work on one MI355X box, in stages (one gpurun call each, a call is limited to 20 minutes):
#   gpurun --timeout 1200 -- 'RK_COMMIT=<sha> bash benchmarks/run_gpu_round4.sh <stage>'
# stages: tests | bench | search | kernels | rows | streams | astar | evaluator   (profiles/README.md maps the records to these commands)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
STAGE=${1:-tests}
O=gpurun_out/r4/$STAGE
mkdir -p $O
# a step that was killed (timeout, fault) ends the script: no further GPU work behind it
step() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -ge 124 ]; then echo "STOP: '$*' ended with $rc" | tee -a $O/stop.log; exit $rc; fi; return $rc; }
case $STAGE in
tests)
	step 1150 python -m pytest tests -m gpu -x -q --durations=12 ${RK_PYTEST_ARGS:-} > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log; tail -25 $O/pytest_gpu.log
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
	RK_BENCH_BACKEND=gloo step 300 python bench.py --gpus 2 --steps 50 --warmup 5 > $O/bench_2ranks_gloo.log 2>&1; tail -1 $O/bench_2ranks_gloo.log | cut -c1-200
	;;
kernels)
	step 300 python benchmarks/kernels.py 2>/dev/null | grep '^{' > $O/kernels.json
	step 200 python benchmarks/kernels686.py 2>/dev/null | grep '^{' > $O/kernels686.json
	step 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kernels -- python3 benchmarks/kernels.py > $O/kernels_prof.log 2>&1
	step 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kernels686 -- python3 benchmarks/kernels686.py > $O/kernels686_prof.log 2>&1
	for c in FETCH_SIZE WRITE_SIZE; do
		step 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_paced_$c -- python3 benchmarks/paced_pmc.py > $O/pmc_paced_$c.log 2>&1
	done
	python benchmarks/paced_pmc.py --summarise $O --out $O/paced_pmc.json > $O/paced_pmc_summary.log 2>&1; cut -c1-600 $O/paced_pmc_summary.log
	;;
rows)
	step 300 python benchmarks/rows_fit.py > $O/rows_fit_events.json 2> $O/rows_fit.err; tail -3 $O/rows_fit_events.json | cut -c1-300
	step 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rows -- python3 benchmarks/rows_fit.py > $O/rows_fit_prof.log 2>&1
	python benchmarks/kernel_trace_by_grid.py $O/prof_rows --segments --min-calls 100 --skip 10 > $O/rows_trace_runs.csv
	python benchmarks/rows_fit.py --fit $O/rows_trace_runs.csv > $O/rows_fit_rocprof.json; cut -c1-400 $O/rows_fit_rocprof.json
	;;
streams)
	step 400 python benchmarks/pace_streams.py 2> $O/pace_streams.err | grep '^{' > $O/pace_streams.json; cut -c1-600 $O/pace_streams.json
	RK_PACE_SERIAL=0 step 400 python benchmarks/pace_streams.py 2> $O/pace_streams_overlapping.err | grep '^{' > $O/pace_streams_overlapping.json; cut -c1-600 $O/pace_streams_overlapping.json
	;;
search)
	step 500 python benchmarks/astar_small.py > $O/astar_small.json 2> $O/astar_small.err; cut -c1-300 $O/astar_small.json
	step 300 python benchmarks/grow_cost.py 2> $O/grow_cost.err | grep '^{' > $O/grow_cost.json; cut -c1-300 $O/grow_cost.json
	;;
protocol)
	step 900 python benchmarks/reference_protocol.py 2> $O/reference_protocol.err > $O/reference_protocol.json; cut -c1-600 $O/reference_protocol.json; tail -3 $O/reference_protocol.err
	;;
evaluator)
	step 300 python benchmarks/graph_kept.py 2> $O/graph_kept.err | grep '^{' > $O/graph_kept.json; cut -c1-400 $O/graph_kept.json
	step 900 python benchmarks/evaluator.py 2> $O/evaluator.err | grep '^{' > $O/evaluator.json; cut -c1-700 $O/evaluator.json
	;;
astar)
	for n in 1 10 100 700; do
		step 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/n$n -- python3 benchmarks/astar_floor.py --n $n > $O/n$n.log 2>&1
	done
	python benchmarks/astar_floor.py --table $O --out $O/astar_floor.json | cut -c1-1500
	for n in 10 100 700; do python benchmarks/astar_floor.py --n $n; done > $O/astar_floor_unprofiled.json 2>/dev/null; cat $O/astar_floor_unprofiled.json
	;;
esac
find $O -name "*kernel_trace.csv" -size +3M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
du -sh $O
