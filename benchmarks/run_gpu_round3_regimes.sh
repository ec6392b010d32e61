#!/bin/bash
# The cache regimes of the fan-out with PMC counters, shipping (paced) kernel: gpurun_out/r3reg/ -> profiles/r03_regimes_paced.json, r03_regimes_paced_pmc.json
# (the same passes as run_gpu_round3_diag.sh took for the ring form; one counter group per rocprofv3 pass, never combined with other traces).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r3reg
mkdir -p $O
step() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -ge 124 ]; then echo "STOP: '$*' ended with $rc" | tee -a $O/stop.log; exit $rc; fi; return $rc; }
step 200 python benchmarks/pmc_regimes.py --manifest $O/manifest.json > $O/regimes_plain.json 2>$O/regimes_plain.err; cat $O/regimes_plain.json
p=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum" "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
	p=$((p+1))
	step 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_$p -- python3 benchmarks/pmc_regimes.py --manifest $O/manifest_pmc_$p.json > $O/pmc_$p.log 2>&1
done
python benchmarks/pmc_regimes_summary.py --manifest $O/manifest.json --passes $O/pmc_* --kernel k_expand12p --out $O/regimes_pmc.json > $O/regimes_summary.log 2>&1
find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
tail -3 $O/regimes_summary.log | cut -c1-300
du -sh $O
