set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 300 python -m pytest tests/test_astar_gpu.py tests/test_engine_errors_gpu.py -m gpu -x -q > gpurun_out/r2/pytest_astar.log 2>&1; echo "exit $?" >> gpurun_out/r2/pytest_astar.log; tail -15 gpurun_out/r2/pytest_astar.log
grep -q "exit 0" gpurun_out/r2/pytest_astar.log || exit 1
timeout -k 10 400 python -m pytest tests/test_sharded_gpu.py tests/test_cube_gpu.py tests/test_configs_full_gpu.py -k "not mcts" -m gpu -x -q > gpurun_out/r2/pytest_sharded.log 2>&1; echo "exit $?" >> gpurun_out/r2/pytest_sharded.log; tail -15 gpurun_out/r2/pytest_sharded.log
timeout -k 10 300 python benchmarks/astar_small.py > gpurun_out/r2/astar_small.json 2>&1; cat gpurun_out/r2/astar_small.json
python benchmarks/kernels686.py > gpurun_out/r2/k686.json 2>&1; cat gpurun_out/r2/k686.json
python benchmarks/tune_expand.py 16 24:2048 24:1024 24:1536 24:2560 24:3072 24:4096 29:2048 29:4096 > gpurun_out/r2/tune2.log 2>&1; tail -9 gpurun_out/r2/tune2.log | cut -c1-100,330-420
