"""
The N > 1 path on CPU (gloo, world_size 2 and 3): the collectives of the hash-sharded search (counts / variable-size
record all-to-all / all-gather / broadcast), the global top-N selection every rank must agree on, and the weak-scaling
aggregation of bench.py.  No kernels run here; the GPU side is covered by tests/test_sharded_gpu.py.
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
		assert (tp.world, tp.rank, tp.backend, tp.on_device) == (world, rank, "gloo", False)
		# all-gather: every rank sees every rank's vector, in rank order
		allv = tp.all_gather_vec(np.array([rank, 10.0 * rank + 0.5]))
		assert allv.tolist() == [[r, 10.0 * r + 0.5] for r in range(world)]
		# counts: rank r sends (r + 1) * (d + 1) records to rank d
		send_counts = np.array([(rank + 1) * (d + 1) for d in range(world)], dtype=np.int64)
		recv_counts = tp.exchange_counts(send_counts)
		assert recv_counts.tolist() == [(s + 1) * (rank + 1) for s in range(world)]
		# records: byte 0 = sender, byte 1 = destination, bytes 2.. = running number; grouped by destination
		rows = []
		for d in range(world):
			for k in range(send_counts[d]):
				rows.append([rank, d, k] + [7] * 29)
		send = torch.tensor(rows, dtype=torch.uint8)
		# (Transport moves the result to the GPU only when there is one; patch the device for the CPU test)
		import librubiks_amd.solving.sharded as sh
		sh.gpu = torch.device("cpu")
		recv = tp.exchange_records(send, send_counts, recv_counts).numpy()
		want = [[s, rank, k] for s in range(world) for k in range((s + 1) * (rank + 1))]
		assert recv[:, :3].tolist() == want and (recv[:, 3:] == 7).all()
		# empty exchange
		z = np.zeros(world, np.int64)
		assert tp.exchange_records(torch.zeros((0, 16), dtype=torch.uint8), z, tp.exchange_counts(z)).shape == (0, 16)
		# broadcast from the last rank
		got = tp.broadcast_vec(np.array([rank, 5, 6], dtype=np.int64), world - 1)
		assert got.tolist() == [world - 1, 5, 6]
		# the global selection is the same on every rank
		rng = np.random.RandomState(3)
		heads = np.sort(rng.randint(0, 6, (world, 8)).astype(np.float64), axis=1)
		heads[0, 5:] = np.inf
		pops = select_pops(tp.all_gather_vec(heads[rank]), 8)
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
