"""
HBM traffic of the paced one-hot and 6x8x6 fan-out kernels from the PMC counters ("traffic ratios for the two
paced kernels are on record").

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR/pmc_paced_FETCH_SIZE -- python3 benchmarks/paced_pmc.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d DIR/pmc_paced_WRITE_SIZE -- python3 benchmarks/paced_pmc.py
    python benchmarks/paced_pmc.py --summarise DIR --out profiles/r04_paced_pmc.json

The workload: a few launches each of `as_oh` (float32 and bfloat16, 1 Mi states) and of the 6x8x6 fan-out + goal test (200 000
parents), inputs rotating over sets larger than the Infinity Cache together (cache-neutral), outputs over two buffers.  Counters
are collected in two separate passes (never together, never with --stats: MI355X_MICROARCH.md, HBM section); FETCH_SIZE is doubled
per that guide's gfx950 correction, both are KiB.  Both kernels' read phases fetch their inputs once from HBM and the storing
workgroups fetch them again from the Infinity Cache, so fetch = 2 x the input bytes is the design, not waste.
"""
import argparse
import csv
import glob
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
csv.field_size_limit(1 << 30)

N_OH, N_686, LAUNCHES = 1 << 20, 200_000, 6
ALGO = {
	"k_as_oh<float": {"read": 20 * N_OH, "write": 1920 * N_OH},
	"k_as_oh<rk::bf16_tag": {"read": 20 * N_OH, "write": 960 * N_OH},
	"k_fanout686p<true": {"read": 288 * N_686, "write": (3456 + 12) * N_686},
}


def workload():
	import numpy as np
	import torch
	from librubiks_amd import _ffi, cube
	_ffi.check(_ffi.lib().rk_init(0))
	g = torch.Generator(device="cuda")
	g.manual_seed(4)
	ins = [cube.device.apply_sequences(torch.randint(0, 12, (12, N_OH), device="cuda", dtype=torch.uint8, generator=g), False, True) for _ in range(16)]   # 336 MB
	for dtype in (torch.float32, torch.bfloat16):
		outs = [torch.empty((N_OH, 480), dtype=dtype, device="cuda") for _ in range(2)]
		for i in range(LAUNCHES):
			cube.device.as_oh(ins[i % len(ins)], outs[i % 2], dtype)
		torch.cuda.synchronize()
		del outs
	del ins
	cube.set_is2024(False)
	solved = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(cube.get_solved(), (N_686, 6, 8, 6)))).cuda()
	ins = []
	for _ in range(6):                                                                             # 346 MB of distinct parents
		p = solved
		for _ in range(4):
			p = cube.device.multi_rotate(p, torch.randint(0, 12, (N_686,), device="cuda", dtype=torch.uint8, generator=g))
		ins.append(p)
	outs = [(torch.empty((12 * N_686, 6, 8, 6), dtype=torch.int8, device="cuda"), torch.empty(12 * N_686, dtype=torch.uint8, device="cuda")) for _ in range(2)]
	for i in range(LAUNCHES):
		cube.device.expand12(ins[i % len(ins)], *outs[i % 2])
	torch.cuda.synchronize()
	cube.set_is2024(True)


def counter(directory, name):
	hits = sorted(glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True))
	if not hits:
		sys.exit(f"no counter_collection.csv under {directory}")
	per = {}
	with open(hits[-1], newline="") as f:
		for row in csv.DictReader(f):
			if row["Counter_Name"] != name:
				continue
			for key in ALGO:
				if key in row["Kernel_Name"]:
					per.setdefault(key, []).append(float(row["Counter_Value"]))
	return per


def summarise(root, out):
	fetch = counter(os.path.join(root, "pmc_paced_FETCH_SIZE"), "FETCH_SIZE")
	write = counter(os.path.join(root, "pmc_paced_WRITE_SIZE"), "WRITE_SIZE")
	rec = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 benchmarks/paced_pmc.py (two separate passes)",
	       "corrections": "KiB units; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM section)", "kernels": {}}
	for key, algo in ALGO.items():
		if key not in fetch or key not in write:
			continue
		f = 2.0 * statistics.fmean(fetch[key][1:] or fetch[key]) * 1024.0        # the first launch of a kind also pays for cold tables
		w = statistics.fmean(write[key][1:] or write[key]) * 1024.0
		rec["kernels"][key] = {"dispatches": len(fetch[key]), "fetch_bytes_per_launch": f, "write_bytes_per_launch": w,
		                       "algorithmic_read": algo["read"], "algorithmic_write": algo["write"],
		                       "fetch_over_algorithmic_read": f / algo["read"], "write_over_algorithmic_write": w / algo["write"],
		                       "traffic_over_algorithmic": (f + w) / (algo["read"] + algo["write"])}
	with open(out, "w") as fh:
		json.dump(rec, fh, indent=1)
	print(json.dumps(rec))


if __name__ == "__main__":
	ap = argparse.ArgumentParser()
	ap.add_argument("--summarise")
	ap.add_argument("--out", default="paced_pmc.json")
	a = ap.parse_args()
	if a.summarise:
		summarise(a.summarise, a.out)
	else:
		workload()
