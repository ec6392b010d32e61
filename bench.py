"""
Headline benchmark: cube node-expansions/s (12-child fan-out + goal test) on a 1 M-state batch.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: ONE launch of the fan-out kernel (rk_expand12) on 1 000 000
device-resident parent states (depth-20 random walks from solved), producing 12 000 000 children and their solved
flags.  Inputs are in HBM before the timed region; nothing is copied over PCIe inside it.  With N GPUs every rank
expands its own 1 M-state batch (independent units, no data-path collective): weak scaling, value = all ranks'
expansions / max-over-ranks time.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline      algorithmic bytes (272 B per parent) / measured kernel time vs the 8 TB/s HBM peak
  cpu_baseline  the CPU port of the reference's fan-out idiom (oracle/, NumPy, 1 core) timed on a bounded sample,
                and the C/OpenMP restatement on all host cores
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
	sys.path.insert(0, ROOT)

N_PARENTS = 1_000_000
BYTES_PER_PARENT = 20 + 240 + 12          # read parent, write 12 children, write 12 solved flags
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: 8.0 TB/s spec (about 6.3 TB/s achievable)


def make_parents(n: int, seed: int) -> torch.Tensor:
	"""n depth-20 random walks from the solved state, generated on the device (synthetic data)."""
	from librubiks_amd import cube
	g = torch.Generator(device="cuda")
	g.manual_seed(seed)
	acts = torch.randint(0, 12, (20, n), device="cuda", dtype=torch.uint8, generator=g)
	return cube.device.apply_sequences(acts, with_solved=False, only_last=True)


def cpu_baseline(sample: int = 1_000_000):
	"""
	The reference's CPU path, as far as it can travel: the oracle's NumPy port of the fan-out idiom
	`multi_rotate(np.repeat(S, 12, 0), *iter_actions(n))` + `multi_is_solved` (agents.py:277-281, :321) on one
	core, and the C/OpenMP restatement on every host core.  Only this function touches oracle/.
	"""
	from oracle import c_oracle, cube_oracle as orc
	rng = np.random.RandomState(1)
	s = orc.repeat_state(orc.SOLVED, sample)
	for _ in range(20):
		a = rng.randint(0, 12, sample)
		s = c_oracle.multi_rotate(s, a.astype(np.uint8), threads=4)
	t0 = time.perf_counter()
	faces, dirs = orc.iter_actions(sample)
	children = orc.multi_rotate(np.repeat(s, 12, axis=0), faces, dirs)
	flags = orc.multi_is_solved(children)
	t_np = time.perf_counter() - t0
	threads = c_oracle.max_threads()
	out = np.empty((12 * sample, 20), np.int8)
	fl = np.empty(12 * sample, np.uint8)
	c_oracle.expand12(s, threads=threads, out=out, solved=fl)          # warm
	reps, t0 = 5, time.perf_counter()
	for _ in range(reps):
		c_oracle.expand12(s, threads=threads, out=out, solved=fl)
	t_c = (time.perf_counter() - t0) / reps
	assert (out == children).all() and (fl.astype(bool) == flags).all()
	return {
		"value": sample / t_np, "unit": "expansions/s", "cores": 1, "kind": "port",
		"sample": f"{sample} parents (depth-20 walks), NumPy port of the reference fan-out idiom + goal test, {t_np:.2f} s",
		"native": {"value": sample / t_c, "unit": "expansions/s", "cores": threads, "kind": "port",
		           "sample": f"{sample} parents x {reps} reps, C -O3 + OpenMP restatement, {t_c * 1e3:.1f} ms/rep"},
		"host_cpus": os.cpu_count(),
	}


def pmc_traffic():
	"""HBM bytes per launch from the committed rocprofv3 PMC summary (profiles/), if one exists."""
	path = os.path.join(ROOT, "profiles", "expand12_pmc.json")
	try:
		with open(path) as f:
			return json.load(f).get("hbm_bytes_per_launch")
	except (OSError, ValueError):
		return None


def max_over_ranks(elapsed: float, dist, device) -> float:
	"""The slowest rank's time: what divides the whole job's units (weak scaling, no data-path collective)."""
	t = torch.tensor([elapsed], dtype=torch.float64, device=device)
	if dist is not None:
		dist.all_reduce(t, op=dist.ReduceOp.MAX)
	return float(t.item())


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--gpus", type=int, default=1)
	ap.add_argument("--steps", type=int, default=300)
	ap.add_argument("--warmup", type=int, default=30)
	ap.add_argument("--no-cpu-baseline", action="store_true")
	args = ap.parse_args()

	rank = int(os.environ.get("RANK", "0"))
	local_rank = int(os.environ.get("LOCAL_RANK", "0"))
	world = int(os.environ.get("WORLD_SIZE", "1"))
	if world != args.gpus:
		if world == 1 and args.gpus > 1:
			sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
	# One rank per GPU.  RK_BENCH_BACKEND=gloo is a rehearsal mode for a one-GPU box (several ranks share cuda:0 and
	# the only collectives -- barrier and the MAX of the elapsed times -- go through the host); the driver uses nccl (= RCCL).
	backend = os.environ.get("RK_BENCH_BACKEND", "nccl")
	device_index = local_rank % max(1, torch.cuda.device_count())
	torch.cuda.set_device(device_index)
	dist = None
	if world > 1:
		import torch.distributed as dist
		if backend == "nccl":
			dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
		else:
			dist.init_process_group(backend)
	reduce_device = torch.device("cuda") if backend == "nccl" else torch.device("cpu")

	from librubiks_amd import _ffi, cube
	_ffi.check(_ffi.lib().rk_init(device_index))

	parents = make_parents(N_PARENTS, seed=1000 + rank)
	children = torch.empty((12 * N_PARENTS, 20), dtype=torch.int8, device="cuda")
	solved = torch.empty(12 * N_PARENTS, dtype=torch.uint8, device="cuda")
	stats = torch.tensor([0, _ffi.INT64_MAX], dtype=torch.int64, device="cuda")

	def step():
		cube.device.expand12(parents, children, solved, stats)

	def fence():
		if dist is not None:
			dist.barrier()
		torch.cuda.synchronize()

	for _ in range(args.warmup):
		step()
	ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	fence()
	t0 = time.perf_counter()
	ev0.record()
	for _ in range(args.steps):
		step()
	ev1.record()
	fence()
	elapsed = time.perf_counter() - t0

	elapsed_max = max_over_ranks(elapsed, dist, reduce_device)
	kernel_ms = ev0.elapsed_time(ev1) / args.steps          # HIP events on the launch stream: back-to-back launches

	# sanity of the timed work: the children of the last step are a real fan-out (spot check on the device)
	probe = cube.device.multi_rotate(children[:12 * 4096].contiguous(),
	                                 torch.arange(12, dtype=torch.uint8, device="cuda").bitwise_xor(1).repeat(4096))
	assert torch.equal(probe.view(4096, 12, 20), parents[:4096].view(4096, 1, 20).expand(4096, 12, 20))

	if rank == 0:
		value = world * N_PARENTS * args.steps / elapsed_max
		achieved = BYTES_PER_PARENT * N_PARENTS / (kernel_ms * 1e-3) / 1e9
		line = {
			"metric": "cube node-expansions/sec (12-child fan-out) at 1M-state batch",
			"value": value, "unit": "expansions/s",
			"n_gpus": world, "steps": args.steps, "warmup": args.warmup,
			"ms_per_step": elapsed_max / args.steps * 1e3,
			"higher_is_better": True, "scaling": "weak", "vs_baseline": None,
			"dtype": "u8", "data": "synthetic",
			"config": {"workload": f"configs[1]: {world}xMI355X fan-out (12 moves) + is_solved on 1M depth-20 scrambles per GPU, "
			                       "device-resident, one rk_expand12 launch per step",
			           "parents_per_gpu": N_PARENTS, "children_per_step": 12 * N_PARENTS * world,
			           "parallelism": f"independent batches x{world}"},
			"transitions_per_s": 12 * value,
			"roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
			             "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(),
			             "kernel": "rk::k_expand12<true, 1, true, 4, false, 1>", "kernel_ms": kernel_ms,
			             "algorithmic_bytes_per_launch": BYTES_PER_PARENT * N_PARENTS},
		}
		if world == 1 and not args.no_cpu_baseline:
			line["cpu_baseline"] = cpu_baseline()
		print(json.dumps(line), flush=True)
	if dist is not None:
		dist.barrier()
		dist.destroy_process_group()


if __name__ == "__main__":
	main()
