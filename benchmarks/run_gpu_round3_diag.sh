#!/bin/bash
# Round-3 diagnostic records on one MI355X box (tuning build needed: python -m librubiks_amd.build --tune).  Raw output: gpurun_out/r3diag/.
#   cache regimes of the fan-out with PMC counters  -> profiles/r03_regimes.json, r03_regimes_pmc.json
#   kernel shapes by batch size, parents from HBM   -> profiles/r03_tune_sizes.json (RK_TUNE_N sweep), r03_tune_expand.json (1 M and 16 M incl. geometry diagnostics)
#   SoA vs AoS interleaved                          -> profiles/r03_layout_ab.json
#   queue-insert grid A/B, eager vs hipGraph gaps   -> profiles/r03_queue_insert_grid.json, r03_astar_graph_gaps.json
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r3diag
mkdir -p $O
step() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -ge 124 ]; then echo "STOP: '$*' ended with $rc" | tee -a $O/stop.log; exit $rc; fi; return $rc; }
step 200 python benchmarks/pmc_regimes.py --manifest $O/manifest.json > $O/regimes_plain.json 2>$O/regimes_plain.err; cat $O/regimes_plain.json
p=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum" "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum" "FETCH_SIZE" "WRITE_SIZE"; do
	p=$((p+1))
	step 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_$p -- python3 benchmarks/pmc_regimes.py --manifest $O/manifest_pmc_$p.json > $O/pmc_$p.log 2>&1
done
python benchmarks/pmc_regimes_summary.py --manifest $O/manifest.json --passes $O/pmc_* --kernel k_expand12p --out $O/regimes_pmc.json > $O/regimes_summary.log 2>&1
step 400 python benchmarks/tune_expand.py 24:3072 100 101:3072 102:3072 104:3072 108:3072 102:2048 104:1024 108:512 121:3072 142:3072 44 45 40 > $O/tune_1m.json 2>/dev/null
RK_TUNE_N=16000000 step 400 python benchmarks/tune_expand.py 24:3072 100 102:3072 108:512 124:2048 142:2048 200 202:3072 44 45 40 > $O/tune_16m.json 2>/dev/null
for n in 250000 500000 1000000 2000000 4000000 8000000 16000000 32000000; do
	t=$((n / 64)); g2=$((t / 8)); g4=$((t / 16)); g15=$((t / 6))
	extra=""; if [ $n -ge 8000000 ]; then extra="200"; fi
	RK_TUNE_N=$n step 300 python benchmarks/tune_expand.py 24:3072 100 101:$g15 102:$g2 104:$g4 102:3072 102:4096 $extra > $O/tune_$n.json 2>/dev/null
done
step 300 python benchmarks/layout_ab.py > $O/layout_ab.json 2>/dev/null; cat $O/layout_ab.json | cut -c1-300
for g in 8 128 512; do
	RK_INSERT_MIN_GRID=$g step 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ins_$g -- python3 benchmarks/astar_profile.py --expansions 100 --net stub > $O/prof_ins_$g.log 2>&1
done
step 300 rocprofv3 --kernel-trace --output-format csv -d $O/gaps_eager -- python3 benchmarks/astar_graph_gaps.py run --mode eager > $O/gaps_eager.log 2>&1
step 300 rocprofv3 --kernel-trace --output-format csv -d $O/gaps_graph -- python3 benchmarks/astar_graph_gaps.py run --mode graph > $O/gaps_graph.log 2>&1
python benchmarks/astar_graph_gaps.py summary --eager $O/gaps_eager --graph $O/gaps_graph > $O/gaps_summary.json 2>&1
find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
du -sh $O
