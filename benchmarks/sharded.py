"""
BASELINE.json configs[4]: A* with the open set hash-sharded across the GPUs of one node, RCCL all-to-all frontier exchange
over xGMI, depth-20 scrambles.  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        benchmarks/sharded.py --depth 20 --expansions 700 --max-states 4000000 --games 3

or, without a launcher, `python benchmarks/sharded.py --world 8 ...`: the script then starts its own ranks (benchmarks/spawn.py:
fresh child processes, before torch or the GPU is touched).  With one process and no --world: world = 1, the collectives
short-circuit.  RK_BENCH_BACKEND=gloo rehearses the
protocol with several ranks sharing one GPU (host-staged collectives).  Rank 0 prints one JSON object: the proof that the
collectives saw the ranks and the leg's flat keys (benchmarks/multi_gpu.py -- the very functions `bench.py --gpus N` runs behind its
fan-out region, so the number the driver's scaling run takes and this harness's are one code path).  Real nets run with the first
layer fused + folded; --net stub is the exact integer heuristic (benchmarks/nets.py FastStub, no oracle/ import).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse_args():
	ap = argparse.ArgumentParser()
	ap.add_argument("--world", type=int, default=0, help="ranks to start when no launcher did (WORLD_SIZE unset); 0 = run as launched")
	ap.add_argument("--depth", type=int, default=20)
	ap.add_argument("--expansions", type=int, default=700)          # configs/main_eval.ini:9
	ap.add_argument("--lam", type=float, default=0.16)              # configs/main_eval.ini:8
	ap.add_argument("--max-states", type=int, default=4_000_000)
	ap.add_argument("--capacity", type=int, default=0, help="states per rank (default: max_states / world * 1.5 + slack)")
	ap.add_argument("--games", type=int, default=3)
	ap.add_argument("--time-limit", type=float, default=60.0)
	ap.add_argument("--poll", type=int, default=4)
	ap.add_argument("--net", default="fc_small", choices=["fc_small", "fc_small_bf16", "stub"])
	ap.add_argument("--mcts", type=int, default=0, help="instead of A*: configs[3] with this many trees PER RANK (weak scaling), partitioned "
	                "over the ranks (PartitionedMCTS: no collective in the loop, one all-gather of results at the end)")
	ap.add_argument("--sims", type=int, default=4096)
	ap.add_argument("--weak", action="store_true", help="weak scaling: --expansions and --max-states are per rank (multiplied by the world size)")
	return ap.parse_args()


if __name__ == "__main__":
	_a = parse_args()
	if _a.world > 1 and "WORLD_SIZE" not in os.environ:              # no launcher: start the ranks here, before any GPU call
		from benchmarks import spawn
		sys.exit(spawn.run_ranks(os.path.abspath(__file__), sys.argv[1:], _a.world))

import numpy as np
import torch


def main():
	args = parse_args()

	rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
	if args.world and args.world != world:
		sys.exit(f"--world {args.world} but WORLD_SIZE={world}")
	local = int(os.environ.get("LOCAL_RANK", "0"))
	backend = os.environ.get("RK_BENCH_BACKEND", "nccl")
	dev = local % max(1, torch.cuda.device_count())
	torch.cuda.set_device(dev)
	import torch.distributed as dist
	if world > 1:
		if backend == "nccl":
			dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
		else:
			dist.init_process_group(backend)

	from benchmarks import multi_gpu
	from librubiks_amd import _ffi
	_ffi.check(_ffi.lib().rk_init(dev))
	d = dist if world > 1 else None
	net_kind = {"fc_small": "fc_small", "fc_small_bf16": "fc_small_bf16", "stub": "stub"}[args.net]
	rows = {}
	if world > 1:
		rows.update(multi_gpu.collective_proof(dist, backend))
	if args.mcts:
		rows.update(multi_gpu.partitioned_mcts_leg(d, backend, world, rank, trees_per_rank=args.mcts, sims=args.sims, net_kind=net_kind))
	else:
		# --expansions / --max-states are the totals over all ranks, as before: one scaling mode, named by --weak
		rows.update(multi_gpu.sharded_astar_leg(d, backend, world, rank, weak=args.weak, games=args.games, depth=args.depth, lam=args.lam,
		                                        expansions=args.expansions, budget=args.max_states, poll=args.poll, time_limit=args.time_limit,
		                                        net_kind=net_kind))
	if rank == 0:
		print(json.dumps(rows), flush=True)
	if world > 1:
		dist.barrier()
		dist.destroy_process_group()


if __name__ == "__main__":
	main()
