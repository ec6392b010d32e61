set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2m
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_oh_linear_gpu.py tests/test_mcts_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "exit $?" >> $O/pytest.log; tail -4 $O/pytest.log
grep -q "exit 0" $O/pytest.log || exit 1
timeout -k 10 400 python benchmarks/oh_linear.py 2>/dev/null | grep '^{' > $O/oh_linear.json; cat $O/oh_linear.json | cut -c1-700
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ohl -- python3 benchmarks/oh_linear.py > $O/prof_ohl.log 2>&1
f=$(find $O/prof_ohl -name "*kernel_stats.csv"); grep "k_ohl\|k_as_oh\|Cijk" $f | cut -c1-60,200-330
