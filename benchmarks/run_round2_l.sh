set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2l
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_mcts_gpu.py tests/test_configs_full_gpu.py tests/test_oh_linear_gpu.py tests/test_engine_errors_gpu.py -k "mcts" -m gpu -x -q > $O/pytest.log 2>&1; echo "exit $?" >> $O/pytest.log; tail -4 $O/pytest.log
grep -q "exit 0" $O/pytest.log || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mcts256 -- python3 benchmarks/search.py mcts --sims 256 > $O/prof_mcts256.log 2>&1
f=$(find $O/prof_mcts256 -name "*kernel_stats.csv"); grep "k_mcts" $f | cut -c1-200
for a in "" "--bf16 1" "--bf16 1 --fused 1"; do python benchmarks/search.py mcts $a 2>/dev/null | tail -1 >> $O/search.json; done
cat $O/search.json | cut -c1-330
