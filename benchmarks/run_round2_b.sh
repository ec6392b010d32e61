set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
python -m pytest tests/test_cube_gpu.py -m gpu -x -q > gpurun_out/r2/pytest_cube.log 2>&1; tail -3 gpurun_out/r2/pytest_cube.log
python benchmarks/kernels686.py > gpurun_out/r2/k686.json 2>&1; cat gpurun_out/r2/k686.json
python benchmarks/tune_expand.py 16 24:3907 24:1954 24:977 24:2048 28 29:1954 0 17 18 > gpurun_out/r2/tune.log 2>&1; tail -12 gpurun_out/r2/tune.log | cut -c1-250
