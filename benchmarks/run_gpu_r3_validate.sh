#!/bin/bash
# Round 3: full GPU suite, smoke, bench, the same bench under rocprofv3 (stats) and the two PMC passes, sizes, kernels.  Raw output: gpurun_out/r3v/.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r3v
mkdir -p $O
step() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -ge 124 ]; then echo "STOP: '$*' ended with $rc" | tee -a $O/stop.log; exit $rc; fi; return $rc; }
step 1100 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log; tail -15 $O/pytest_gpu.log
grep -q "exit 0" $O/pytest_gpu.log || exit 1
step 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
step 300 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-1800
step 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/bench_prof.log 2>&1; tail -1 $O/bench_prof.log | cut -c1-600
step 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline > $O/pmc_f.log 2>&1
step 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline > $O/pmc_w.log 2>&1
python benchmarks/pmc_summary.py --stats $O/prof_stats --fetch $O/pmc_f --write $O/pmc_w --kernel k_expand12r --commit "${RK_COMMIT:-unknown}" --out-pmc $O/expand12_pmc.json --out-stats $O/expand12_kernel_stats.csv > $O/pmc_summary.log 2>&1; cat $O/pmc_summary.log | cut -c1-900
step 400 python benchmarks/sizes.py 2>/dev/null | grep '^{' > $O/sizes.json; cat $O/sizes.json
step 300 python benchmarks/kernels.py 2>/dev/null | grep '^{' > $O/kernels.json; cat $O/kernels.json | cut -c1-250
find $O -name "*kernel_trace.csv" -size +3M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
du -sh $O
