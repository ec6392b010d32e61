set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 300 python -m pytest tests/test_astar_gpu.py tests/test_sharded_gpu.py tests/test_cube_gpu.py tests/test_astar_batch_gpu.py -m gpu -x -q > gpurun_out/r2/pytest_h.log 2>&1; echo "exit $?" >> gpurun_out/r2/pytest_h.log; tail -5 gpurun_out/r2/pytest_h.log
grep -q "exit 0" gpurun_out/r2/pytest_h.log || exit 1
python bench.py > gpurun_out/r2/bench_h.log 2>&1; tail -1 gpurun_out/r2/bench_h.log | cut -c1-400
timeout -k 10 300 python benchmarks/astar_small.py > gpurun_out/r2/astar_small5.json 2>&1; grep stub gpurun_out/r2/astar_small5.json | cut -c1-120
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/prof_astar100e -- python3 benchmarks/astar_profile.py --expansions 100 --net stub > gpurun_out/r2/prof_astar100e.log 2>&1
python benchmarks/sizes.py > gpurun_out/r2/sizes.json 2>&1; cat gpurun_out/r2/sizes.json | cut -c1-300
for d in prof_astar100e; do f=$(find gpurun_out/r2/$d -name "*kernel_stats.csv"); python3 - "$f" <<'PY'
import csv,sys
csv.field_size_limit(1<<30)
rows=list(csv.reader(open(sys.argv[1])))
print(sys.argv[1])
for r in rows[1:11]:
    print(r[0][:70].ljust(70), r[1].rjust(7), r[3][:9].rjust(10), r[4][:6].rjust(7))
PY
done
