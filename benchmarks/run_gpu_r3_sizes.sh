#!/bin/bash
# Round 3: ring-form fan-out vs batch size and grid shape, all with parents from HBM (inputs rotate over >= 640 MB).  Raw output: gpurun_out/r3s/.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r3s
mkdir -p $O
step() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -ge 124 ]; then echo "STOP: '$*' ended with $rc" | tee -a $O/stop.log; exit $rc; fi; return 0; }
for n in 250000 500000 1000000 2000000 4000000 8000000 16000000 32000000; do
	t=$((n / 64))
	g2=$((t / 8)); g3=$((t / 12)); g4=$((t / 16)); g15=$((t / 6))
	extra=""
	if [ $n -ge 8000000 ]; then extra="200 202:3072"; fi
	RK_TUNE_N=$n step 300 python benchmarks/tune_expand.py 24:3072 100 101:$g15 101:$g2 102:$g2 102:$g3 102:$g4 104:$g4 101:3072 102:3072 102:4096 102:6144 $extra > $O/tune_$n.json 2>$O/tune_$n.err
	python - $O/tune_$n.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
	if l.startswith("{"):
		d = json.loads(l); print(d["parents"], d["id"], d["grid_blocks"], d["correct"], d["ms_median"], d["frac_of_8TBs"], d["variant"][:60])
PY
done
