set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 400 python -m pytest tests/test_astar_gpu.py tests/test_engine_errors_gpu.py tests/test_oh_linear_gpu.py -m gpu -x -q > gpurun_out/r2/pytest_astar2.log 2>&1; echo "exit $?" >> gpurun_out/r2/pytest_astar2.log; tail -25 gpurun_out/r2/pytest_astar2.log
grep -q "exit 0" gpurun_out/r2/pytest_astar2.log || exit 1
timeout -k 10 500 python -m pytest tests/test_sharded_gpu.py tests/test_configs_full_gpu.py tests/test_adi_gpu.py tests/test_simple_agents_gpu.py tests/test_wire_gpu.py -k "not mcts" -m gpu -x -q > gpurun_out/r2/pytest_sharded2.log 2>&1; echo "exit $?" >> gpurun_out/r2/pytest_sharded2.log; tail -15 gpurun_out/r2/pytest_sharded2.log
timeout -k 10 300 python benchmarks/astar_small.py > gpurun_out/r2/astar_small2.json 2>&1; cat gpurun_out/r2/astar_small2.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/prof_astar100b -- python3 benchmarks/astar_profile.py --expansions 100 --net stub --graph 1 > gpurun_out/r2/prof_astar100b.log 2>&1
timeout -k 10 400 python benchmarks/oh_linear.py > gpurun_out/r2/oh_linear.json 2>&1; cat gpurun_out/r2/oh_linear.json
for d in prof_astar100b; do f=$(find gpurun_out/r2/$d -name "*kernel_stats.csv"); python3 - "$f" <<'PY'
import csv,sys
csv.field_size_limit(1<<30)
rows=list(csv.reader(open(sys.argv[1])))
print(sys.argv[1])
for r in rows[1:14]:
    print(r[0][:70].ljust(70), r[1].rjust(7), r[3][:9].rjust(10), r[4][:6].rjust(7))
PY
done
