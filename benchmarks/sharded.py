"""
BASELINE.json configs[4]: A* with the open set hash-sharded across the GPUs of one node, RCCL all-to-all frontier exchange
over xGMI, depth-20 scrambles.  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        benchmarks/sharded.py --depth 20 --expansions 700 --max-states 4000000 --games 3

or, without a launcher, `python benchmarks/sharded.py --world 8 ...`: the script then starts its own ranks (benchmarks/spawn.py:
fresh child processes, before torch or the GPU is touched).  With one process and no --world: world = 1, the collectives
short-circuit.  RK_BENCH_BACKEND=gloo rehearses the
protocol with several ranks sharing one GPU (host-staged collectives).  Rank 0 prints one JSON object per game with the
iteration time split into all-gather / select+expand / all-to-all / insert / net / push (device time between HIP events on the
search stream) and one summary object.  No 8-GPU node was available to the build: the harness exists so that the number can be
taken when one is.
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
	ap.add_argument("--fused", default="", choices=["", "epilogue", "folded"], help="fused first layer mode of the net (not for the stub)")
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

	from benchmarks.nets import FcSmall
	from librubiks_amd import _ffi, cube
	from librubiks_amd.solving.sharded import ShardedAStar
	_ffi.check(_ffi.lib().rk_init(dev))
	if args.net == "stub":
		from oracle.search_oracle import StubNet          # the exact stub heuristic (a net stand-in, not the checker)
		net = StubNet()
	else:
		net = FcSmall(seed=0).cuda().eval()               # same seed on every rank: identical weights
		if args.net.endswith("bf16"):
			net = net.to(torch.bfloat16)
	if args.mcts:
		from librubiks_amd.solving.sharded import PartitionedMCTS
		trees = args.mcts * world
		starts = []
		for i in range(trees):
			np.random.seed(1000 + i)
			starts.append(cube.scramble(14, True)[0])
		agent = PartitionedMCTS(net, 0.6, trees, capacity=12 * args.sims + 16, max_path=4096, **({"fused_first_layer": args.fused} if args.fused else {}))
		agent.search(np.array(starts), max_sims=8, use_graph=False)                    # warm-up
		torch.cuda.synchronize()
		if world > 1:
			dist.barrier()
		t0 = time.perf_counter()
		solved = agent.search(np.array(starts), max_sims=args.sims, use_graph=True, poll=64)
		torch.cuda.synchronize()
		dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
		if world > 1:
			dist.all_reduce(dt, op=dist.ReduceOp.MAX)
		if rank == 0:
			print(json.dumps({"bench": "partitioned_mcts", "config": f"configs[3] per rank: {args.mcts} trees x {args.sims} sims, depth-14 scrambles, c=0.6, "
			                  f"net={args.net}{', fused ' + args.fused if args.fused else ''}, hipGraph=on, world={world}", "world": world,
			                  "backend": backend if world > 1 else "local", "trees": trees, "seconds_max_over_ranks": float(dt[0]),
			                  "tree_sims": int(agent.sims.sum()), "tree_sims_per_s": float(agent.sims.sum()) / float(dt[0]),
			                  "solved": int(solved.sum()), "states": int(agent.states.sum()), "scaling": "weak"}), flush=True)
		if world > 1:
			dist.barrier()
			dist.destroy_process_group()
		return
	cap = args.capacity or int(args.max_states / world * 1.5) + 12 * args.expansions * world + 1024
	agent = ShardedAStar(net, args.lam, args.expansions, capacity=cap, poll=args.poll, profile=True,
	                     fused_first_layer=(args.fused or False) if args.net != "stub" else False)
	# one-time costs (pool allocation, GEMM kernel selection, process-group warm-up) stay out of the timed games
	np.random.seed(12345)
	warm, _, _ = cube.scramble(args.depth, True)
	agent.search(warm, time_limit=args.time_limit, max_states=30 * 12 * args.expansions * world)
	rows = []
	for g in range(args.games):
		np.random.seed(g)
		state, _, _ = cube.scramble(args.depth, True)
		torch.cuda.synchronize()
		if world > 1:
			dist.barrier()
		t0 = time.perf_counter()
		solved = agent.search(state, time_limit=args.time_limit, max_states=args.max_states)
		torch.cuda.synchronize()
		dt = time.perf_counter() - t0
		row = {"bench": "sharded_astar", "game": g, "world": world, "backend": backend if world > 1 else "local", "solved": bool(solved),
		       "stop": agent.stop_reason, "iterations": agent.iterations, "total_states": agent.total_states, "seconds": dt,
		       "ms_per_iteration": dt / max(agent.iterations, 1) * 1e3, "states_per_s": agent.total_states / dt,
		       "solution_length": len(agent.action_queue) if solved else None, "collectives": agent.tp.collectives,
		       "net_rows_per_iteration": agent.net_rows_total / max(agent.iterations, 1), "net_rows_max": agent.net_rows_max,
		       "net_rows_bound_12N": 12 * args.expansions, "phase_ms": agent.phase_ms}
		rows.append(row)
		if rank == 0:
			print(json.dumps(row), flush=True)
	if rank == 0:
		it = sum(r["iterations"] for r in rows)
		print(json.dumps({"bench": "sharded_astar summary", "config": f"configs[4]: depth-{args.depth} scrambles, lambda={args.lam}, N={args.expansions}, "
		                  f"max_states={args.max_states}, net={args.net}{', fused ' + args.fused if args.fused else ''}, world={world}", "games": len(rows), "solved": sum(r["solved"] for r in rows),
		                  "states_per_s": sum(r["total_states"] for r in rows) / sum(r["seconds"] for r in rows),
		                  "ms_per_iteration": sum(r["seconds"] for r in rows) / max(it, 1) * 1e3,
		                  "collectives_per_iteration": rows[-1]["collectives"] / max(sum(r["iterations"] for r in rows), 1)}), flush=True)
	if world > 1:
		dist.barrier()
		dist.destroy_process_group()


if __name__ == "__main__":
	main()
