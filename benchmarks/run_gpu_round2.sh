#!/bin/bash
# Reproduces the round-2 measurement records of profiles/ on one MI355X box (run from the repository root, e.g. through
# `gpurun -- 'bash benchmarks/run_gpu_round2.sh'`).  Raw output goes to gpurun_out/r2/; profiles/README.md says which file
# becomes which record (benchmarks/pmc_summary.py condenses the rocprofv3 directories).
set -x
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r2
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log; tail -12 $O/pytest_gpu.log
grep -q "exit 0" $O/pytest_gpu.log || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
# headline: bench, the same command under rocprofv3 (stats), and the two PMC passes (never combined)
python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/bench_prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline > $O/pmc_w.log 2>&1
# the N > 1 contract of bench.py, rehearsed with two ranks sharing the GPU (host-staged gloo collectives)
RK_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 50 --warmup 5 > $O/bench_2ranks_gloo.log 2>&1; tail -1 $O/bench_2ranks_gloo.log | cut -c1-300
# kernels
python benchmarks/kernels.py 2>/dev/null | grep '^{' > $O/kernels.json
python benchmarks/kernels686.py 2>/dev/null | grep '^{' > $O/kernels686.json
python benchmarks/sizes.py 2>/dev/null | grep '^{' > $O/sizes.json
python benchmarks/oh_linear.py 2>/dev/null | grep '^{' > $O/oh_linear.json
# search loops
timeout -k 10 300 python benchmarks/astar_small.py 2>/dev/null | grep '^{' > $O/astar_small.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_astar100 -- python3 benchmarks/astar_profile.py --expansions 100 --net stub > $O/prof_astar100.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_astar1000 -- python3 benchmarks/astar_profile.py --expansions 1000 --net bf16 --max-states 400000 > $O/prof_astar1000.log 2>&1
rm -f $O/search.json
for a in "" "--bf16 1" "--fused 1" "--bf16 1 --fused 1" "--bf16 1 --fused 2" "--bf16 1 --fused 3" "--fused 3"; do python benchmarks/search.py astar $a 2>/dev/null | grep '^{' >> $O/search.json; done
for a in "" "--bf16 1" "--fused 1" "--bf16 1 --fused 1" "--bf16 1 --fused 3" "--fused 3"; do python benchmarks/search.py mcts $a 2>/dev/null | tail -1 >> $O/search.json; done
timeout -k 10 600 python benchmarks/adi.py 2>/dev/null | grep '^{' > $O/adi.json
rm -f $O/astar_batch.json
for a in "" "--bf16 1" "--bf16 1 --fused 3"; do python benchmarks/search.py astar_batch --expansions 100 --max-states 50000 $a 2>/dev/null | tail -1 >> $O/astar_batch.json; done
for a in "--bf16 1" "--bf16 1 --fused 3"; do python benchmarks/search.py astar_batch $a 2>/dev/null | tail -1 >> $O/astar_batch.json; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mcts256 -- python3 benchmarks/search.py mcts --sims 256 > $O/prof_mcts256.log 2>&1
# sharded search rehearsals (no multi-GPU node: world 1, and two ranks over gloo on the one GPU)
rm -f $O/sharded.json
timeout -k 10 200 python benchmarks/sharded.py --depth 14 --expansions 100 --max-states 300000 --games 2 --net stub 2>/dev/null | grep '^{' >> $O/sharded.json
RK_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 benchmarks/sharded.py --depth 14 --expansions 100 --max-states 300000 --games 2 --net stub 2>/dev/null | grep '^{' >> $O/sharded.json
timeout -k 10 200 python benchmarks/sharded.py --depth 20 --expansions 700 --max-states 2000000 --games 2 --net fc_small_bf16 --time-limit 30 2>/dev/null | grep '^{' >> $O/sharded.json
timeout -k 10 200 python benchmarks/sharded.py --depth 20 --expansions 700 --max-states 2000000 --games 2 --net fc_small_bf16 --fused folded --time-limit 30 2>/dev/null | grep '^{' >> $O/sharded.json
# configs[3] across ranks: trees partitioned (world 1, and two ranks over gloo sharing the GPU)
timeout -k 10 300 python benchmarks/sharded.py --mcts 256 --net fc_small_bf16 --fused folded 2>/dev/null | grep '^{' >> $O/sharded.json
RK_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29535 benchmarks/sharded.py --mcts 128 --net fc_small_bf16 --fused folded 2>/dev/null | grep '^{' >> $O/sharded.json
cat $O/search.json | cut -c1-300
