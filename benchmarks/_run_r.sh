set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2x
mkdir -p $O
timeout -k 10 800 python -m pytest tests/test_astar_gpu.py tests/test_configs_full_gpu.py tests/test_sharded_gpu.py tests/test_engine_errors_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "exit $?" >> $O/pytest.log; tail -15 $O/pytest.log
grep -q "exit 0" $O/pytest.log || exit 1
cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_astar100 -o a100 -- python3 $GRAFT_REPO_ROOT/benchmarks/astar_profile.py --expansions 100 --net stub > /dev/null 2>&1
cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_astar1000 -o a1000 -- python3 $GRAFT_REPO_ROOT/benchmarks/astar_profile.py --expansions 1000 --net bf16 --max-states 400000 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && grep -h "rk::" $O/prof_astar100/*kernel_stats.csv | cut -c1-120 && grep -h "rk::" $O/prof_astar1000/*kernel_stats.csv | cut -c1-120
for a in "--bf16 1 --fused 3" "--fused 0"; do timeout -k 10 200 python benchmarks/search.py astar $a 2>/dev/null | tail -1 >> $O/search.json; done
cut -c1-400 $O/search.json
