#!/bin/bash
# Sweep of the paced fan-out's phase length and reader count at the bench's own size (1 M parents): does pipelining the read phase
# against the stores (several short phases) beat one phase?  One bench.py run per point; prints frac and kernel ms.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4/phase
mkdir -p $O
: > $O/sweep.json
for rep in 1 2; do
for phase in 16384 8192 4096; do
	for pull in 128 256 64; do
		for first in 256 128; do
			RK_PACE_PHASE=$phase RK_PACE_PULL=$pull RK_PACE_PULL_FIRST=$first timeout -k 10 120 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-search-legs 2>/dev/null | grep '^{' | \
				python -c "import sys, json; d = json.loads(sys.stdin.read()); print(json.dumps({'phase': $phase, 'pull': $pull, 'pull_first': $first, 'rep': $rep, 'frac': d['roofline']['frac'], 'kernel_ms': d['roofline']['kernel_ms_back_to_back'], 'frac_ring': d['roofline']['frac_ring_same_box']}))" | tee -a $O/sweep.json
		done
	done
done
done
