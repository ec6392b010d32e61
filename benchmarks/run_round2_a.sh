set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r2/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r2/pytest_gpu.log
python bench.py > gpurun_out/r2/bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/prof_stats -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/r2/bench_prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r2/pmc_f -- python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline > gpurun_out/r2/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r2/pmc_w -- python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline > gpurun_out/r2/pmc_w.log 2>&1
python benchmarks/tune_expand.py 16 24 24:2048 24:4096 28 0 17 18 > gpurun_out/r2/tune.log 2>&1
tail -5 gpurun_out/r2/pytest_gpu.log; tail -2 gpurun_out/r2/bench.log
