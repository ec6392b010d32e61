set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2f
mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_oh_linear_gpu.py tests/test_astar_gpu.py tests/test_mcts_gpu.py tests/test_engine_errors_gpu.py -m gpu -x -q > $O/pytest_i.log 2>&1; echo "exit $?" >> $O/pytest_i.log; tail -5 $O/pytest_i.log
grep -q "exit 0" $O/pytest_i.log || exit 1
python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/bench_prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline > $O/pmc_w.log 2>&1
timeout -k 10 300 python benchmarks/astar_small.py > $O/astar_small.json 2>&1; grep stub $O/astar_small.json | cut -c1-120
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_astar100 -- python3 benchmarks/astar_profile.py --expansions 100 --net stub > $O/prof_astar100.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_astar1000 -- python3 benchmarks/astar_profile.py --expansions 1000 --net bf16 --max-states 400000 > $O/prof_astar1000.log 2>&1
for a in "" "--bf16 1" "--fused 1" "--bf16 1 --fused 1"; do python benchmarks/search.py mcts $a 2>/dev/null | tail -1 >> $O/search.json; done
for a in "" "--bf16 1" "--fused 1" "--bf16 1 --fused 1"; do python benchmarks/search.py astar $a 2>/dev/null | grep '^{' >> $O/search.json; done
cat $O/search.json | cut -c1-330
timeout -k 10 400 python benchmarks/oh_linear.py 2>/dev/null | grep '^{' > $O/oh_linear.json
timeout -k 10 200 python benchmarks/sharded.py --depth 14 --expansions 100 --max-states 300000 --games 2 --net stub 2>/dev/null | grep '^{' > $O/sharded.json
RK_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 benchmarks/sharded.py --depth 14 --expansions 100 --max-states 300000 --games 2 --net stub 2>/dev/null | grep '^{' >> $O/sharded.json
timeout -k 10 200 python benchmarks/sharded.py --depth 20 --expansions 700 --max-states 2000000 --games 1 --net fc_small_bf16 --time-limit 30 2>/dev/null | grep '^{' >> $O/sharded.json
cat $O/sharded.json | cut -c1-500
python benchmarks/kernels.py 2>/dev/null | grep '^{' > $O/kernels.json; cat $O/kernels.json | cut -c1-200
