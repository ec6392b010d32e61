#!/bin/bash
# Round 3, second validation: tests of the engines that changed (MCTS expand-ahead, sliced forwards, sharded net batch), then the
# search benchmarks behind them.  Raw output: gpurun_out/r3b/.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r3b
mkdir -p $O
step() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -ge 124 ]; then echo "STOP: '$*' ended with $rc" | tee -a $O/stop.log; exit $rc; fi; return $rc; }
step 1100 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log; tail -15 $O/pytest_gpu.log
grep -q "exit 0" $O/pytest_gpu.log || exit 1
rm -f $O/astar_batch.json $O/search.json
for a in "--bf16 1 --slice 0" "--bf16 1" "--bf16 1 --slice 8192" "--bf16 1 --slice 32768" "--bf16 1 --fused 3 --slice 0" "--bf16 1 --fused 3"; do step 300 python benchmarks/search.py astar_batch $a 2>/dev/null | tail -1 >> $O/astar_batch.json; done
for a in "--bf16 1 --slice 0" "--bf16 1" "--bf16 1 --fused 3"; do step 300 python benchmarks/search.py astar_batch --expansions 100 --max-states 50000 $a 2>/dev/null | tail -1 >> $O/astar_batch.json; done
cut -c1-900 $O/astar_batch.json
for a in "--bf16 1" "--bf16 1 --fused 3" "--fused 3"; do step 300 python benchmarks/search.py mcts $a 2>/dev/null | tail -1 >> $O/search.json; done
cut -c1-600 $O/search.json
step 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mcts -- python3 benchmarks/search.py mcts --bf16 1 --fused 3 > $O/prof_mcts.log 2>&1
python - $O <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/prof_mcts/**/*kernel_stats.csv", recursive=True))[-1]
for i, r in enumerate(csv.reader(open(f))):
	if i < 9: print([c[:70] for c in r[:6]])
PY
step 300 rocprofv3 --kernel-trace --output-format csv -d $O/gaps_eager -- python3 benchmarks/astar_graph_gaps.py run --mode eager > $O/gaps_eager.log 2>&1; tail -1 $O/gaps_eager.log
step 300 rocprofv3 --kernel-trace --output-format csv -d $O/gaps_graph -- python3 benchmarks/astar_graph_gaps.py run --mode graph > $O/gaps_graph.log 2>&1; tail -1 $O/gaps_graph.log
python benchmarks/astar_graph_gaps.py summary --eager $O/gaps_eager --graph $O/gaps_graph > $O/gaps_summary.json 2>&1; cat $O/gaps_summary.json
step 400 python benchmarks/sizes.py 2>/dev/null | grep '^{' > $O/sizes.json; cat $O/sizes.json
find $O -name "*kernel_trace.csv" -size +3M -delete
du -sh $O
