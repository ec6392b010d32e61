#!/bin/bash
# Regenerates the round-3 measurement records on one MI355X box: `gpurun --timeout 1200 -- 'RK_COMMIT=<sha> bash benchmarks/run_gpu_round3.sh'`,
# then `bash benchmarks/collect_profiles_r3.sh` (profiles/README.md maps files to commands).  Raw output: gpurun_out/r3/.
# The search-loop records come from run_gpu_round3_search.sh (a second call: one call is limited to 20 minutes).
# The diagnostic records (cache regimes with PMC counters, kernel shapes by size, layout A/B, ...) come from run_gpu_round3_diag.sh.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
# a step that was killed (timeout, fault) ends the script: no further GPU work behind it
step() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -ge 124 ]; then echo "STOP: '$*' ended with $rc" | tee -a $O/stop.log; exit $rc; fi; return $rc; }
step 1100 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log; tail -14 $O/pytest_gpu.log
grep -q "exit 0" $O/pytest_gpu.log || exit 1
step 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
# headline: bench, the same command under rocprofv3 (stats), and the two PMC passes (never combined)
step 300 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-400
step 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/bench_prof.log 2>&1
step 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline > $O/pmc_f.log 2>&1
step 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline > $O/pmc_w.log 2>&1
python benchmarks/pmc_summary.py --stats $O/prof_stats --fetch $O/pmc_f --write $O/pmc_w --kernel k_expand12p --commit "${RK_COMMIT:-unknown}" \
	--command "python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline" --out-pmc $O/expand12_pmc.json --out-stats $O/expand12_kernel_stats.csv > $O/pmc_summary.log 2>&1; cut -c1-400 $O/pmc_summary.log
# the N > 1 contract of bench.py, rehearsed with two ranks sharing the GPU (host-staged gloo collectives)
RK_BENCH_BACKEND=gloo step 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 50 --warmup 5 > $O/bench_2ranks_gloo.log 2>&1; tail -1 $O/bench_2ranks_gloo.log | cut -c1-200
# kernels and sizes
step 400 python benchmarks/sizes.py 2>/dev/null | grep '^{' > $O/sizes.json; cut -c1-160 $O/sizes.json
RK_PACE=0 step 400 python benchmarks/sizes.py 2>/dev/null | grep '^{' > $O/sizes_unpaced.json       # the ring form alone, same box
step 300 python benchmarks/kernels.py 2>/dev/null | grep '^{' > $O/kernels.json
step 200 python benchmarks/kernels686.py 2>/dev/null | grep '^{' > $O/kernels686.json
find $O -name "*kernel_trace.csv" -size +3M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
du -sh $O
