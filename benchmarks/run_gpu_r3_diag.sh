#!/bin/bash
# Round-3 diagnosis of the fan-out kernel's cache regimes (one MI355X box).  Raw output: gpurun_out/r3d/.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r3d
mkdir -p $O
step() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -ge 124 ]; then echo "STOP: '$*' ended with $rc" | tee -a $O/stop.log; exit $rc; fi; return 0; }
step 300 python bench.py --steps 300 --warmup 30 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-1500
step 400 python benchmarks/tune_expand.py 24:3072 100 101:3072 102:3072 104:3072 108:3072 101:2048 102:2048 104:2048 108:2048 102:1024 104:1024 108:1024 108:512 104:512 121:3072 124:2048 128:1024 142:2048 44 45 40 > $O/tune_1m.json 2>$O/tune_1m.err; cat $O/tune_1m.json | cut -c1-260
RK_TUNE_N=16000000 step 400 python benchmarks/tune_expand.py 24:3072 100 101:3072 102:3072 104:3072 108:3072 104:2048 108:1024 108:512 124:2048 142:2048 44 45 40 > $O/tune_16m.json 2>$O/tune_16m.err; cat $O/tune_16m.json | cut -c1-260
step 300 python benchmarks/layout_ab.py > $O/layout_ab.json 2>$O/layout_ab.err; cat $O/layout_ab.json
step 200 python benchmarks/pmc_regimes.py --manifest $O/manifest.json > $O/regimes_plain.json 2>$O/regimes_plain.err; cat $O/regimes_plain.json
rocprofv3 -L > $O/counters_list.txt 2>&1
p=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum" "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum" "FETCH_SIZE" "WRITE_SIZE"; do
	p=$((p+1))
	step 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_$p -- python3 benchmarks/pmc_regimes.py --manifest $O/manifest_pmc_$p.json > $O/pmc_$p.log 2>&1
	echo "pass $p ($grp): $(tail -1 $O/pmc_$p.log | cut -c1-200)"
done
python benchmarks/pmc_regimes_summary.py --manifest $O/manifest.json --passes $O/pmc_* --out $O/regimes_pmc.json > $O/regimes_summary.log 2>&1; cat $O/regimes_summary.log | cut -c1-1800
# keep the merge-back small: the raw traces are not needed
find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
du -sh $O
