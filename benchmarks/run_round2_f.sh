set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 600 python -m pytest tests/test_astar_gpu.py tests/test_sharded_gpu.py tests/test_oh_linear_gpu.py tests/test_mcts_gpu.py tests/test_astar_batch_gpu.py -m gpu -x -q > gpurun_out/r2/pytest_f.log 2>&1; echo "exit $?" >> gpurun_out/r2/pytest_f.log; tail -8 gpurun_out/r2/pytest_f.log
grep -q "exit 0" gpurun_out/r2/pytest_f.log || exit 1
timeout -k 10 300 python benchmarks/astar_small.py > gpurun_out/r2/astar_small3.json 2>&1; grep stub gpurun_out/r2/astar_small3.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/prof_astar100c -- python3 benchmarks/astar_profile.py --expansions 100 --net stub > gpurun_out/r2/prof_astar100c.log 2>&1
timeout -k 10 400 python benchmarks/oh_linear.py > gpurun_out/r2/oh_linear2.json 2>&1; cat gpurun_out/r2/oh_linear2.json | cut -c1-400
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/prof_mcts256 -- python3 benchmarks/search.py mcts --sims 256 > gpurun_out/r2/prof_mcts256.log 2>&1
python benchmarks/search.py mcts > gpurun_out/r2/mcts_fp32_b.json 2>&1; tail -1 gpurun_out/r2/mcts_fp32_b.json | cut -c1-400
python benchmarks/search.py mcts --bf16 1 > gpurun_out/r2/mcts_bf16.json 2>&1; tail -1 gpurun_out/r2/mcts_bf16.json | cut -c1-400
timeout -k 10 200 python benchmarks/sharded.py --depth 14 --expansions 100 --max-states 300000 --games 2 --net stub > gpurun_out/r2/sharded_w1.json 2>&1; tail -3 gpurun_out/r2/sharded_w1.json | cut -c1-600
RK_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 benchmarks/sharded.py --depth 14 --expansions 100 --max-states 300000 --games 2 --net stub > gpurun_out/r2/sharded_w2_gloo.json 2>&1; tail -3 gpurun_out/r2/sharded_w2_gloo.json | cut -c1-600
for d in prof_astar100c prof_mcts256; do f=$(find gpurun_out/r2/$d -name "*kernel_stats.csv"); python3 - "$f" <<'PY'
import csv,sys
csv.field_size_limit(1<<30)
rows=list(csv.reader(open(sys.argv[1])))
print(sys.argv[1])
for r in rows[1:14]:
    print(r[0][:70].ljust(70), r[1].rjust(7), r[3][:9].rjust(10), r[4][:6].rjust(7))
PY
done
