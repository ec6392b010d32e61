#!/bin/bash
# Round-3 records of the search loops on one MI355X box (second call after run_gpu_round3.sh).  Raw output: gpurun_out/r3/.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
step() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -ge 124 ]; then echo "STOP: '$*' ended with $rc" | tee -a $O/stop.log; exit $rc; fi; return $rc; }
# search loops
step 400 python benchmarks/astar_small.py 2>/dev/null | grep '^{' > $O/astar_small.json
rm -f $O/search.json $O/astar_batch.json $O/sharded.json
for a in "" "--bf16 1" "--bf16 1 --fused 3" "--fused 3"; do step 300 python benchmarks/search.py astar $a 2>/dev/null | grep '^{' >> $O/search.json; done
for a in "--bf16 1" "--bf16 1 --fused 3" "--fused 3"; do step 300 python benchmarks/search.py mcts $a 2>/dev/null | tail -1 >> $O/search.json; done
for a in "--bf16 1" "--bf16 1 --fused 3" "--bf16 1 --slice 0" ""; do step 300 python benchmarks/search.py astar_batch $a 2>/dev/null | tail -1 >> $O/astar_batch.json; done
for a in "--bf16 1" "--bf16 1 --fused 3"; do step 300 python benchmarks/search.py astar_batch --expansions 100 --max-states 50000 $a 2>/dev/null | tail -1 >> $O/astar_batch.json; done
step 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mcts -- python3 benchmarks/search.py mcts --bf16 1 --fused 3 > $O/prof_mcts.log 2>&1
step 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_astar100 -- python3 benchmarks/astar_profile.py --expansions 100 --net stub > $O/prof_astar100.log 2>&1
# sharded search rehearsals (no multi-GPU node: world 1, and two ranks over gloo on the one GPU, with the real net)
step 200 python benchmarks/sharded.py --depth 20 --expansions 700 --max-states 2000000 --games 2 --net fc_small_bf16 --fused folded --time-limit 30 2>/dev/null | grep '^{' >> $O/sharded.json
RK_BENCH_BACKEND=gloo step 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 benchmarks/sharded.py --depth 20 --expansions 700 --max-states 2000000 --games 2 --net fc_small_bf16 --time-limit 30 2>/dev/null | grep '^{' >> $O/sharded.json
RK_BENCH_BACKEND=gloo step 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29535 benchmarks/sharded.py --depth 20 --expansions 700 --max-states 2000000 --games 2 --net fc_small_bf16 --fused folded --time-limit 30 2>/dev/null | grep '^{' >> $O/sharded.json
find $O -name "*kernel_trace.csv" -size +3M -delete
du -sh $O
