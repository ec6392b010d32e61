set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2k
mkdir -p $O
for a in "" "--bf16 1" "--fused 1" "--bf16 1 --fused 1"; do python benchmarks/search.py astar $a 2>/dev/null | grep '^{' >> $O/search_astar.json; done
cat $O/search_astar.json | cut -c1-420
