"""
The N > 1 path on CPU (gloo, world_size 2 and 3): the two collectives of the hash-sharded search (fixed-size all-gather,
equal-split all-to-all of {header, padded payload} blocks) and the path-walk broadcast, the global top-N selection rule
every rank must agree on, and the weak-scaling aggregation of bench.py.  No kernels run here; the GPU side is covered by tests/test_sharded_gpu.py.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from librubiks_amd import _ffi
from librubiks_amd.solving.sharded import Transport, select_pops


def _free_port():
	with socket.socket() as s:
		s.bind(("127.0.0.1", 0))
		return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
	os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
	dist.init_process_group("gloo", rank=rank, world_size=world)
	try:
		tp = Transport()
		assert (tp.world, tp.rank, tp.backend, tp.on_device, tp.shortcut) == (world, rank, "gloo", False, False)
		# collective 1, all-gather of a fixed-size vector: every rank sees every rank's vector, in rank order
		allv = tp.all_gather(torch.tensor([rank, 10.0 * rank + 0.5], dtype=torch.float64))
		assert allv.tolist() == [[r, 10.0 * r + 0.5] for r in range(world)]
		# collective 2, all-to-all of equal fixed-size blocks {header = counts, padded payload}: rank r sends (r + 1) * (d + 1)
		# records to rank d; byte 0 = sender, byte 1 = destination, byte 2 = running number
		block = 32 + 40 * 4
		send = torch.zeros((world, block), dtype=torch.uint8)
		for d in range(world):
			cnt = (rank + 1) * (d + 1)
			send[d, 0] = cnt
			for k in range(cnt):
				send[d, 32 + 4 * k: 32 + 4 * k + 3] = torch.tensor([rank, d, k], dtype=torch.uint8)
		recv = tp.all_to_all(send, torch.zeros_like(send))
		for src in range(world):
			cnt = int(recv[src, 0])
			assert cnt == (src + 1) * (rank + 1)
			rows = recv[src, 32: 32 + 4 * cnt].view(cnt, 4)[:, :3].tolist()
			assert rows == [[src, rank, k] for k in range(cnt)]
		assert tp.collectives == 2
		# broadcast from the last rank (the cross-rank path walk)
		got = tp.broadcast_vec(np.array([rank, 5, 6], dtype=np.int64), world - 1)
		assert got.tolist() == [world - 1, 5, 6]
		# the selection rule gives the same answer on every rank from the gathered heads
		rng = np.random.RandomState(3)
		heads = np.sort(rng.randint(0, 6, (world, 8)).astype(np.float64), axis=1)
		heads[0, 5:] = np.inf
		pops = select_pops(tp.all_gather(torch.from_numpy(heads[rank])).numpy(), 8)
		np.save(os.path.join(out_dir, f"pops{rank}.npy"), pops)
		# bench.py's aggregation: MAX of the ranks' elapsed times
		import bench
		assert bench.max_over_ranks(0.1 * (rank + 1), dist, torch.device("cpu")) == pytest.approx(0.1 * world)
	finally:
		dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_collectives_and_selection_gloo(world, tmp_path):
	mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
	pops = [np.load(tmp_path / f"pops{r}.npy") for r in range(world)]
	assert all((p == pops[0]).all() for p in pops) and pops[0].sum() == 8


def test_select_pops_semantics():
	inf = np.inf
	heads = np.array([[1.0, 2.0, 2.0, 9.0], [2.0, 2.0, 3.0, inf], [0.5, inf, inf, inf]])
	assert select_pops(heads, 4).tolist() == [3, 0, 1]          # ties at 2.0 go to the lower rank first
	assert select_pops(heads, 6).tolist() == [3, 2, 1]
	assert select_pops(heads, 100).tolist() == [4, 3, 1]        # never more than what is open
	assert select_pops(np.full((2, 3), inf), 3).tolist() == [0, 0]
	assert select_pops(heads[:1], 2).tolist() == [2]            # world = 1: the head of the queue, as AStar pops


def test_owner_function_is_balanced_and_host_computable():
	lib = _ffi.lib()
	from oracle import cube_oracle as orc
	np.random.seed(0)
	s = orc.repeat_state(orc.SOLVED, 4000)
	for _ in range(15):
		s = orc.multi_rotate(s, np.random.randint(0, 6, len(s)), np.random.randint(0, 2, len(s)))
	for world in (1, 2, 8):
		owners = np.array([lib.rk_shard_owner(np.ascontiguousarray(x).ctypes.data, world) for x in s[:2000]])
		assert owners.min() >= 0 and owners.max() < world
		counts = np.bincount(owners, minlength=world)
		assert counts.min() > 0.7 * 2000 / world
