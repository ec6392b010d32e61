#!/bin/bash
# Round 3, fourth pass: AStarBatch with one search per forward at N = 1000, kernel floors.  gpurun_out/r3e/
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r3e
mkdir -p $O
step() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -ge 124 ]; then echo "STOP: '$*' ended with $rc" | tee -a $O/stop.log; exit $rc; fi; return $rc; }
rm -f $O/astar_batch.json
for a in "--bf16 1" "--bf16 1 --fused 3" "--bf16 1 --slice 24000" "--bf16 1 --fused 3 --slice 24000" ""; do step 300 python benchmarks/search.py astar_batch $a 2>/dev/null | tail -1 >> $O/astar_batch.json; done
python - $O/astar_batch.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
	d = json.loads(l); print(d["config"][60:200], "| seq", round(d["sequential"]["seconds"], 3), "batch", round(d["batch"]["seconds"], 3), "graph", round(d["batch+graph"]["seconds"], 3))
PY
step 300 python benchmarks/kernels.py 2>/dev/null | grep '^{' > $O/kernels.json; cut -c1-230 $O/kernels.json
du -sh $O
