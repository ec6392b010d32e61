set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2n
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
grep -q "exit 0" $O/pytest_gpu.log || exit 1
timeout -k 10 300 python benchmarks/astar_small.py 2>/dev/null | grep '^{' > $O/astar_small.json; cat $O/astar_small.json | cut -c1-130
for a in "" "--bf16 1" "--fused 1" "--bf16 1 --fused 1"; do python benchmarks/search.py astar $a 2>/dev/null | grep '^{' >> $O/search_astar.json; done
cat $O/search_astar.json | cut -c60-330
