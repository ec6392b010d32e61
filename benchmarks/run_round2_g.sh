set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r2/pytest_g.log 2>&1; echo "exit $?" >> gpurun_out/r2/pytest_g.log; tail -14 gpurun_out/r2/pytest_g.log
grep -q "exit 0" gpurun_out/r2/pytest_g.log || exit 1
timeout -k 10 300 python benchmarks/astar_small.py > gpurun_out/r2/astar_small4.json 2>&1; grep stub gpurun_out/r2/astar_small4.json | cut -c1-120
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/prof_astar100d -- python3 benchmarks/astar_profile.py --expansions 100 --net stub > gpurun_out/r2/prof_astar100d.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/prof_mcts256b -- python3 benchmarks/search.py mcts --sims 256 > gpurun_out/r2/prof_mcts256b.log 2>&1
python benchmarks/search.py mcts > gpurun_out/r2/mcts_fp32_c.json 2>&1; tail -1 gpurun_out/r2/mcts_fp32_c.json | cut -c1-400
python benchmarks/search.py mcts --bf16 1 > gpurun_out/r2/mcts_bf16_c.json 2>&1; tail -1 gpurun_out/r2/mcts_bf16_c.json | cut -c1-400
python bench.py > gpurun_out/r2/bench_g.log 2>&1; tail -1 gpurun_out/r2/bench_g.log | cut -c1-1500
for d in prof_astar100d prof_mcts256b; do f=$(find gpurun_out/r2/$d -name "*kernel_stats.csv"); python3 - "$f" <<'PY'
import csv,sys
csv.field_size_limit(1<<30)
rows=list(csv.reader(open(sys.argv[1])))
print(sys.argv[1])
for r in rows[1:11]:
    print(r[0][:70].ljust(70), r[1].rjust(7), r[3][:9].rjust(10), r[4][:6].rjust(7))
PY
done
