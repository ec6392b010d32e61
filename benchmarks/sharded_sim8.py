"""
What ONE rank's engine costs per iteration at world W, measured on one GPU: all W ranks of a hash-sharded search in one process (one
engine per rank, the two collectives done by hand: all-gather = stack the contributions, all-to-all = transpose the send blocks -- as
tests/test_sharded_gpu.py does for parity), HIP events around rank 0's three engine calls.  No xGMI, no RCCL, no concurrency between
ranks: this is the per-rank KERNEL cost that Amdahl's argument in DESIGN.md section 6 needs (select / insert / push at world 8 scan
W x 12 N incoming slots), not a scaling number.

    python benchmarks/sharded_sim8.py [--world 8] [--expansions 700] [--budget 2000000] [--weak] > profiles/r05_sharded_sim8.json
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks.nets import FastStub  # noqa: E402
from librubiks_amd import _ffi, cube  # noqa: E402
from librubiks_amd.solving.sharded import net_rows  # noqa: E402


def run(world, N, budget, depth, lam, seed):
	lib, st = _ffi.lib(), _ffi.stream_ptr
	np.random.seed(seed)
	start, _, _ = cube.scramble(depth, True)
	cap = int(budget / world * 1.5) + 12 * N + 1024
	hs, sends, mines = [], [], []
	for r in range(world):
		h = C.c_void_p()
		_ffi.check(lib.rk_astar_create_sharded(C.byref(h), cap, N, r, world))
		send = torch.zeros((world, int(lib.rk_astar_shard_block_bytes(h))), dtype=torch.uint8, device="cuda")
		mine = torch.zeros(int(lib.rk_astar_shard_gather_len(h)), dtype=torch.float64, device="cuda")
		_ffi.check(lib.rk_astar_shard_bind(h, mine.data_ptr()))
		_ffi.check(lib.rk_astar_shard_reset(h, start.ctypes.data, lam, send.data_ptr(), st()))
		hs.append(h); sends.append(send); mines.append(mine)
	K = 12 * N
	rows = net_rows(K, world)
	oh = torch.zeros((K, 480), device="cuda")
	net = FastStub()
	dec = (C.c_longlong * 8)()
	ev = lambda: torch.cuda.Event(enable_timing=True)
	marks, iters = [], 0
	while True:
		gathered = torch.stack(mines).contiguous()
		e = [ev() for _ in range(6)]
		for r in range(world):
			if r == 0: e[0].record()
			_ffi.check(lib.rk_astar_shard_select(hs[r], gathered.data_ptr(), 1e10, float(budget), sends[r].data_ptr(), st()))
			if r == 0: e[1].record()
		_ffi.check(lib.rk_astar_shard_decision(hs[0], dec, st()))
		if dec[0]:
			break
		recvs = [torch.stack([sends[src][r] for src in range(world)]).contiguous() for r in range(world)]
		for r in range(world):
			if r == 0: e[2].record()
			_ffi.check(lib.rk_astar_shard_insert(hs[r], recvs[r].data_ptr(), sends[r].data_ptr(), oh.data_ptr(), _ffi.OH_F32, st()))
			if r == 0: e[3].record()
			values = net(oh[:rows], policy=False, value=True).reshape(-1).contiguous()
			if r == 0: e[4].record()
			_ffi.check(lib.rk_astar_shard_push_rows(hs[r], values.data_ptr(), rows, recvs[r].data_ptr(), sends[r].data_ptr(), st()))
			if r == 0: e[5].record()
		marks.append(e)
		iters += 1
	torch.cuda.synchronize()
	skip = min(5, len(marks) // 4)
	sel = np.mean([m[0].elapsed_time(m[1]) for m in marks[skip:]]) * 1e3
	ins = np.mean([m[2].elapsed_time(m[3]) for m in marks[skip:]]) * 1e3
	push = np.mean([m[4].elapsed_time(m[5]) for m in marks[skip:]]) * 1e3
	total, stop = int(dec[3]), int(dec[0])
	for h in hs:
		lib.rk_astar_destroy(h)
	return {"world": world, "expansions_all_ranks": N, "budget": budget, "iterations": iters, "total_states": total, "stop": stop, "net_rows_per_rank": rows,
	        "rank0_select_us": sel, "rank0_insert_us": ins, "rank0_push_us": push, "rank0_engine_us": sel + ins + push}


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--world", type=int, default=8)
	ap.add_argument("--expansions", type=int, default=700)
	ap.add_argument("--budget", type=int, default=2_000_000)
	ap.add_argument("--weak", action="store_true")
	ap.add_argument("--skip-world1", action="store_true", help="only the world-W run (for a kernel profile of it)")
	a = ap.parse_args()
	_ffi.check(_ffi.lib().rk_init(0))
	empty = []
	for _ in range(100):
		x, y = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
		x.record(); y.record()
		empty.append((x, y))
	torch.cuda.synchronize()
	pair_us = float(np.median([x.elapsed_time(y) for x, y in empty])) * 1e3
	rows = []
	for world in ([a.world] if a.skip_world1 else sorted({1, a.world})):
		N = a.expansions * (world if a.weak else 1)
		rows.append(run(world, N, a.budget * (world if a.weak else 1), 20, 0.16, 0))
	print(json.dumps({"bench": "sharded_sim", "mode": "weak" if a.weak else "strong", "net": "exact stub (one kernel)", "event_pair_overhead_us": pair_us,
	                  "note": "per-rank engine kernel time at world W on one GPU (ranks run one after the other; each figure contains one event pair's own cost)",
	                  "runs": rows}), flush=True)


if __name__ == "__main__":
	main()
