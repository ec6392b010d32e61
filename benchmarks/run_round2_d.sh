set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 600 python -m pytest tests/test_mcts_gpu.py tests/test_cube_gpu.py tests/test_configs_full_gpu.py -k "mcts or 686" -m gpu -x -q > gpurun_out/r2/pytest_mcts.log 2>&1; echo "exit $?" >> gpurun_out/r2/pytest_mcts.log; tail -8 gpurun_out/r2/pytest_mcts.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/prof_astar100 -- python3 benchmarks/astar_profile.py --expansions 100 --net stub > gpurun_out/r2/prof_astar100.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/prof_astar100g -- python3 benchmarks/astar_profile.py --expansions 100 --net stub --graph 1 > gpurun_out/r2/prof_astar100g.log 2>&1
python benchmarks/search.py mcts > gpurun_out/r2/mcts_fp32.json 2>&1; tail -1 gpurun_out/r2/mcts_fp32.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/prof_mcts -- python3 benchmarks/search.py mcts --sims 1024 > gpurun_out/r2/prof_mcts.log 2>&1
for d in prof_astar100 prof_astar100g prof_mcts; do f=$(find gpurun_out/r2/$d -name "*kernel_stats.csv"); python3 - "$f" <<'PY'
import csv,sys
csv.field_size_limit(1<<30)
rows=list(csv.reader(open(sys.argv[1])))
print(sys.argv[1])
for r in rows[1:16]:
    print(r[0][:70].ljust(70), r[1].rjust(7), r[3][:9].rjust(10), r[4][:6].rjust(7))
PY
done
