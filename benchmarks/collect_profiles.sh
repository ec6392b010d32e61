#!/bin/bash
# gpurun_out/r2 (raw output of benchmarks/run_gpu_round2.sh) -> the tracked records profiles/r02_* (see profiles/README.md)
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r2
P=profiles
C=$(git rev-parse --short HEAD)
cp $O/pytest_gpu.log $P/r02_pytest_gpu.log
tail -1 $O/bench.log > $P/r02_bench.json
grep '^{' $O/bench_prof.log | tail -1 > $P/r02_bench_under_rocprof.json
grep '^{' $O/bench_2ranks_gloo.log | tail -1 > $P/r02_bench_2ranks_gloo_rehearsal.json
python benchmarks/pmc_summary.py --stats $O/prof_stats --fetch $O/pmc_f --write $O/pmc_w --kernel k_expand12 --commit $C \
	--out-pmc $P/r02_expand12_pmc.json --out-stats $P/r02_expand12_kernel_stats.csv | head -1 > $P/r02_expand12_trace_summary.json
for f in kernels kernels686 sizes oh_linear astar_small search adi astar_batch; do cp $O/$f.json $P/r02_$f.json; done
cp $O/sharded.json $P/r02_sharded_rehearsal.json
python benchmarks/pmc_summary.py --stats $O/prof_astar100 --kernel k_queue_insert --out-stats $P/r02_astar100_kernel_stats.csv > /dev/null
python benchmarks/pmc_summary.py --stats $O/prof_astar1000 --kernel k_queue_insert --out-stats $P/r02_astar1000_bf16_kernel_stats.csv --top 22 > /dev/null
python benchmarks/pmc_summary.py --stats $O/prof_mcts256 --kernel k_mcts_backup_select --out-stats $P/r02_mcts_kernel_stats.csv > /dev/null
git status --short $P | head -30
