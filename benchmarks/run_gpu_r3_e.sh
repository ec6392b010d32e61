#!/bin/bash
# Round 3, fifth pass: AStarBatch with budget-aware polling; queue-insert grid A/B.  gpurun_out/r3f/
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r3f
mkdir -p $O
step() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -ge 124 ]; then echo "STOP: '$*' ended with $rc" | tee -a $O/stop.log; exit $rc; fi; return $rc; }
step 300 python -m pytest tests/test_astar_batch_gpu.py tests/test_astar_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
rm -f $O/astar_batch.json
for a in "--bf16 1" "--bf16 1 --fused 3" "--bf16 1 --slice 0" ""; do step 300 python benchmarks/search.py astar_batch $a 2>/dev/null | tail -1 >> $O/astar_batch.json; done
for a in "--bf16 1" "--bf16 1 --fused 3"; do step 300 python benchmarks/search.py astar_batch --expansions 100 --max-states 50000 $a 2>/dev/null | tail -1 >> $O/astar_batch.json; done
python - $O/astar_batch.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
	d = json.loads(l); print(d["config"][34:150], "| seq", round(d["sequential"]["seconds"], 3), "batch", round(d["batch"]["seconds"], 3), "graph", round(d["batch+graph"]["seconds"], 3), "iters", d["batch"]["search_iterations"])
PY
for g in 8 128 512; do
	RK_INSERT_MIN_GRID=$g step 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ins_$g -- python3 benchmarks/astar_profile.py --expansions 100 --net stub > $O/prof_ins_$g.log 2>&1
	python - $O/prof_ins_$g $g <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
for r in csv.reader(open(f)):
	if "k_queue_insert" in r[0] or "k_end" in r[0] or "k_records_sort" in r[0]: print("min grid", sys.argv[2], r[0][:40], "calls", r[1], "avg", r[3], "min", r[5], "max", r[6])
PY
	tail -1 $O/prof_ins_$g.log | cut -c1-200
done
find $O -name "*kernel_trace.csv" -size +2M -delete
du -sh $O
