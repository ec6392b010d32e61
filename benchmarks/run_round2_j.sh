set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2j
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log; tail -12 $O/pytest_gpu.log
grep -q "exit 0" $O/pytest_gpu.log || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for a in "" "--bf16 1" "--fused 1" "--bf16 1 --fused 1"; do python benchmarks/search.py astar $a 2>/dev/null | grep '^{' >> $O/search.json; done
for a in "" "--bf16 1" "--fused 1" "--bf16 1 --fused 1"; do python benchmarks/search.py mcts $a 2>/dev/null | tail -1 >> $O/search.json; done
cat $O/search.json | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mcts256 -- python3 benchmarks/search.py mcts --sims 256 > $O/prof_mcts256.log 2>&1
f=$(find $O/prof_mcts256 -name "*kernel_stats.csv"); grep "k_mcts" $f | cut -c1-200
