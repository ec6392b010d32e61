#!/bin/bash
# Round 3, third pass: new tests (rk_comm, net batch bound), AStarBatch kernel profile at N = 1000, eager vs hipGraph A*, sharded rehearsals, kernel floors.  gpurun_out/r3c/
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r3c
mkdir -p $O
step() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -ge 124 ]; then echo "STOP: '$*' ended with $rc" | tee -a $O/stop.log; exit $rc; fi; return $rc; }
step 600 python -m pytest tests/test_sharded_gpu.py tests/test_abi.py tests/test_oh_linear_gpu.py -m gpu -x -q --durations=5 > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log; tail -12 $O/pytest_gpu.log
step 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_batch -- python3 benchmarks/search.py astar_batch --bf16 1 --sequential-games 2 > $O/prof_batch.log 2>&1; tail -1 $O/prof_batch.log | cut -c1-400
python - $O <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/prof_batch/**/*kernel_stats.csv", recursive=True))[-1]
for i, r in enumerate(csv.reader(open(f))):
	if i < 16: print([c[:64] for c in r[:7]])
PY
step 400 python benchmarks/astar_small.py 2>/dev/null | grep '^{' > $O/astar_small.json; cut -c1-220 $O/astar_small.json
rm -f $O/sharded.json
step 200 python benchmarks/sharded.py --depth 20 --expansions 700 --max-states 2000000 --games 2 --net fc_small_bf16 --fused folded --time-limit 30 2>/dev/null | grep '^{' >> $O/sharded.json
RK_BENCH_BACKEND=gloo step 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 benchmarks/sharded.py --depth 20 --expansions 700 --max-states 2000000 --games 2 --net fc_small_bf16 --time-limit 30 2>/dev/null | grep '^{' >> $O/sharded.json
RK_BENCH_BACKEND=gloo step 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29535 benchmarks/sharded.py --depth 20 --expansions 700 --max-states 2000000 --games 2 --net fc_small_bf16 --fused folded --time-limit 30 2>/dev/null | grep '^{' >> $O/sharded.json
cut -c1-900 $O/sharded.json
find $O -name "*kernel_trace.csv" -size +3M -delete
du -sh $O
