"""
Dependent memory waits of the A* engine's kernels, counted from the gfx950 ISA (no GPU needed: hipcc cross-compiles).

    python benchmarks/isa_waits.py > profiles/r04_astar_isa_waits.json

For each of the six kernels of an iteration (single-engine instantiations) the script compiles librubiks_amd/csrc/rk_astar.hip to
assembly (`--cuda-device-only -S`) and counts, inside the kernel's body:
  vmem            global / flat loads and atomics (a store does not stall the wave)
  waits           `s_waitcnt vmcnt(...)` instructions that have at least one vmem instruction since the previous such wait: each is a point
                  where the wave stalls for a memory ROUND TRIP before it can go on (an upper bound of the dependent chain of one
                  pass through the code: waits on different branches of an `if` are all counted)
  waits_straight  those of them outside any loop (executed at most once per wave)
  loops_with_vmem loops (backward branches) whose body contains a vmem instruction and a wait: every iteration is one more round
                  trip -- the binary searches (log2 of the run length iterations), the hash probe (usually one), the look-back scan
  barriers        s_barrier instructions (workgroup-wide stalls)
The counts bound the dependent-latency chain of a kernel from below: `waits_straight` round trips plus one per loop iteration.
They are static: how often a loop runs is in the data (DESIGN 3.1 gives the measured kernel times beside them).
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {
	"k_expand_lookup": "_ZN2rk15k_expand_lookupENS_8AstarDevE",
	"k_append<false>": "_ZN2rk8k_appendILb0EEEvNS_8AstarDevEPKh",
	"k_new_rows<4, false>": "_ZN2rk10k_new_rowsILi4ELb0EEEvNS_8AstarDevEPDv4_jjPKh",
	"k_records_sort<256>": "_ZN2rk14k_records_sortILi256EEEvNS_8AstarDevEPKf",
	"k_queue_insert<false>": "_ZN2rk14k_queue_insertILb0EEEvNS_8AstarDevEi",
	"k_end<false>": "_ZN2rk5k_endILb0EEEvNS_8AstarDevEii",
}
VMEM = re.compile(r"\b(global_load|flat_load|buffer_load|global_atomic|flat_atomic|buffer_atomic)")
BRANCH = re.compile(r"\bs_c?branch\w*\s+(\.LBB\d+_\d+)")


def main():
	with tempfile.TemporaryDirectory() as tmp:
		out = os.path.join(tmp, "astar.s")
		subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
		                "--cuda-device-only", "-S", "-o", out, os.path.join(ROOT, "librubiks_amd", "csrc", "rk_astar.hip")],
		               check=True, stderr=subprocess.DEVNULL)
		src = open(out).read().split("\n")
	rec = {"source": "librubiks_amd/csrc/rk_astar.hip", "target": "gfx950", "kernels": {}}
	for name, sym in KERNELS.items():
		i = next(n for n, l in enumerate(src) if l.startswith(sym + ":"))
		j = next(n for n in range(i, len(src)) if src[n].startswith(".Lfunc_end"))
		body = src[i:j]
		labels = {m.group(1): n for n, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
		loops = []
		for n, l in enumerate(body):
			m = BRANCH.search(l)
			if m and m.group(1) in labels and labels[m.group(1)] < n:
				loops.append((labels[m.group(1)], n))
		vm = [n for n, l in enumerate(body) if VMEM.search(l)]
		waits, last = [], -1
		for n, l in enumerate(body):
			if "s_waitcnt" in l and "vmcnt" in l:
				if any(last < v < n for v in vm):
					waits.append(n)
				last = n
		in_loop = lambda n: any(a <= n <= b for a, b in loops)
		rec["kernels"][name] = {
			"instructions": sum(1 for l in body if l.startswith("\t") and not l.strip().startswith((".", ";"))),
			"vmem": len(vm), "waits": len(waits), "waits_straight": sum(1 for w in waits if not in_loop(w)),
			"loops_with_vmem": sum(1 for a, b in loops if any(a <= v <= b for v in vm) and any(a <= w <= b for w in waits)),
			"barriers": sum(1 for l in body if "s_barrier" in l),
		}
	print(json.dumps(rec, indent=1))


if __name__ == "__main__":
	main()
