"""
The multi-GPU legs of the benchmark: what `bench.py --gpus N` (N > 1) runs behind its fan-out region, and what
`benchmarks/sharded.py` calls.  One process per GPU, every rank calls the same functions with the same arguments (they are
collective); rank 0 reports.

  collective_proof   that the collectives really saw N ranks: backend name, dist.get_world_size(), an all-reduce SUM of
                     rank + 1 (= N (N + 1) / 2), an all-to-all of rank-stamped blocks verified on EVERY rank (MIN of the verdicts),
                     both on the device the search's collectives use (device buffers with nccl = RCCL, host with gloo)
  sharded_astar_leg  BASELINE.json configs[4]: `ShardedAStar` (open set hash-sharded, one all-gather + one all-to-all per iteration;
                     per-rank semantics ref:librubiks/solving/agents.py:236-331), depth-20 scrambles, lambda 0.16
                     (ref:configs/main_eval.ini:8-9), fc_small bf16 first layer fused + folded;
                       STRONG: N = 700 whatever the world size, 2 M-state budget -- all ranks together pop 700 nodes per iteration
                       WEAK:   N = 700 x world, budget 2 M x world                -- every rank pops about 700
  partitioned_mcts_leg  configs[3] weak: `PartitionedMCTS`, 256 trees per rank x 4096 simulations, no collective in the loop

All results are FLAT dicts (the driver's parser keeps flat extra keys).  RK_BENCH_SEARCH_NET=stub replaces the net by the exact
integer stub heuristic (benchmarks/nets.py FastStub): then the searches are the ones oracle/sharded_oracle.py can replay, which is
how tests/test_bench_contract_gpu.py pins the leg's state counts.
"""
import os
import time

import numpy as np
import torch

STRONG_N, STRONG_BUDGET, LAMBDA, DEPTH = 700, 2_000_000, 0.16, 20


def _device_for(dist, backend):
	return torch.device("cuda") if backend == "nccl" else torch.device("cpu")


def collective_proof(dist, backend: str) -> dict:
	"""dist: torch.distributed with an initialised default group (world > 1)."""
	world, rank = dist.get_world_size(), dist.get_rank()
	dev = _device_for(dist, backend)
	s = torch.tensor([rank + 1], dtype=torch.int64, device=dev)
	dist.all_reduce(s, op=dist.ReduceOp.SUM)
	# block p of rank r carries (r, p) in every element: after the exchange block q must carry (q, r)
	block = 4096
	send = torch.empty((world, block), dtype=torch.int32, device=dev)
	for p in range(world):
		send[p].fill_(rank * 1000 + p)
	recv = torch.full_like(send, -1)
	dist.all_to_all_single(recv, send)
	want = torch.tensor([q * 1000 + rank for q in range(world)], dtype=torch.int32, device=dev).view(world, 1).expand(world, block)
	ok = torch.tensor([int(torch.equal(recv, want))], dtype=torch.int64, device=dev)
	dist.all_reduce(ok, op=dist.ReduceOp.MIN)
	mine = torch.tensor([rank, torch.cuda.current_device()], dtype=torch.int64, device=dev)
	parts = [torch.empty_like(mine) for _ in range(world)]
	dist.all_gather(parts, mine)
	g = torch.stack(parts)
	return {"collective_backend": dist.get_backend(), "collective_world": world, "rank_checksum": int(s.item()),
	        "rank_checksum_expected": world * (world + 1) // 2, "alltoall_verified_on_every_rank": bool(ok.item()),
	        "rank_devices": [int(x) for x in g[:, 1].tolist()], "collective_buffers": "device" if backend == "nccl" else "host (gloo rehearsal)"}


def search_net(kind: str = None):
	"""fc_small, random init (seed 0: identical on every rank), bfloat16 -- or the exact stub (RK_BENCH_SEARCH_NET=stub)."""
	kind = kind or os.environ.get("RK_BENCH_SEARCH_NET", "fc_small_bf16")
	from benchmarks.nets import FastStub, FcSmall
	if kind == "stub":
		return FastStub(), False, "exact stub heuristic"
	net = FcSmall(seed=0).cuda().eval()
	if kind.endswith("bf16"):
		net = net.to(torch.bfloat16)
	return net, "folded", f"{kind} random init, first layer fused + folded, heads' last layer fused"


def _max_over_ranks(seconds: float, dist, backend) -> float:
	if dist is None:
		return seconds
	t = torch.tensor([seconds], dtype=torch.float64, device=_device_for(dist, backend))
	dist.all_reduce(t, op=dist.ReduceOp.MAX)
	return float(t.item())


def sharded_astar_leg(dist, backend: str, world: int, rank: int, *, weak: bool, games: int = 3, depth: int = DEPTH, lam: float = LAMBDA,
                      expansions: int = STRONG_N, budget: int = STRONG_BUDGET, poll: int = 8, time_limit: float = 30.0, net_kind: str = None,
                      prefix: str = None, seeds=None, force_collectives: bool = False) -> dict:
	"""One scaling mode of configs[4].  `expansions` / `budget` are the world-1 figures; weak scaling multiplies both by `world`."""
	from librubiks_amd import cube
	from librubiks_amd.solving.sharded import ShardedAStar
	net, fused, net_note = search_net(net_kind)
	N = expansions * (world if weak else 1)
	total_budget = budget * (world if weak else 1)
	cap = int(total_budget / world * 1.5) + 12 * N + 1024
	pre = prefix or ("sharded_weak_" if weak else "sharded_")
	seeds = list(range(games)) if seeds is None else list(seeds)

	def barrier():
		torch.cuda.synchronize()
		if dist is not None:
			dist.barrier()

	# the iteration replayed as one hipGraph wherever its collectives are on the device (RCCL) or short-circuit (world 1); RK_SHARD_GRAPH=0/1 overrides
	want_graph = os.environ.get("RK_SHARD_GRAPH", "1" if (world == 1 or backend == "nccl") else "0") != "0"
	want_graph = want_graph and (backend == "nccl" or not force_collectives)            # (host-staged collectives cannot be captured)
	agent = ShardedAStar(net, lam, N, capacity=cap, poll=poll, fused_first_layer=fused, use_hipgraph=want_graph, force_collectives=force_collectives)
	np.random.seed(12345)
	agent.search(cube.scramble(depth, True)[0], time_limit=time_limit, max_states=min(total_budget, 30 * 12 * N))      # pools, GEMM selection, group warm-up
	starts = []
	for g in seeds:
		np.random.seed(g)
		starts.append(cube.scramble(depth, True)[0])
	def play(st):
		barrier()
		t0 = time.perf_counter()
		ok = agent.search(st, time_limit=time_limit, max_states=total_budget)
		torch.cuda.synchronize()
		return ok, _max_over_ranks(time.perf_counter() - t0, dist, backend)      # the same number on every rank

	# Game 0 both ways first -- the captured iteration replayed (one launch per iteration) and the same iteration issued eagerly (about
	# thirty launches) -- then the timed games in the faster form: which one wins depends on whether the host's launch bill or the
	# GPU bounds the iteration at THIS world size (world 1, N = 700: the GPU, and the replay is 5-9 % slower; 1 344-row nets at world 8:
	# unknown until a node runs it).  The times are max-over-ranks, so every rank takes the same decision.
	graph_ms = eager_ms = None
	can_graph = False
	if want_graph:
		_, t = play(starts[0])
		can_graph = agent.use_hipgraph and agent.graph_error is None
		if can_graph:
			graph_ms = t / max(agent.iterations, 1) * 1e3
	graph_error = agent.graph_error
	agent.use_hipgraph = False
	_, t = play(starts[0])
	eager_ms = t / max(agent.iterations, 1) * 1e3
	use_graph = can_graph and graph_ms < eager_ms
	agent.use_hipgraph = use_graph
	secs = states = iters = solved = rows = launches = 0
	stops, game_secs, game_iters = [], [], []
	for st in starts:
		ok, t = play(st)
		game_secs.append(t)
		game_iters.append(agent.iterations)
		secs += t
		states += agent.total_states
		iters += agent.iterations
		solved += bool(ok)
		rows += agent.net_rows_total
		launches += agent.host_launches
		stops.append(agent.stop_reason)
	graph_state = ("replayed" if use_graph else "eager (faster on game 0)" if can_graph else "eager: " + (graph_error or "not requested"))
	repeated, rows_fixed = agent.repeated, agent.net_rows_max
	# the phase split: one more pass over the first game with HIP events between the phases of every iteration (device time on the
	# search stream; the events themselves cost a few microseconds per phase, so the timed passes above run without them)
	agent.profile = True
	barrier()
	agent.search(starts[0], time_limit=time_limit, max_states=total_budget)
	torch.cuda.synchronize()
	ph = dict(agent.phase_ms)
	agent.profile = False
	collectives = agent.tp.collectives
	del agent
	torch.cuda.empty_cache()
	out = {
		pre + "states_per_s": states / secs, pre + "ms_per_iteration": secs / max(iters, 1) * 1e3, pre + "iterations": iters,
		pre + "total_states": states, pre + "games": len(starts), pre + "solved": solved, pre + "stop_reasons": ",".join(stops),
		pre + "expansions_per_iteration": N, pre + "budget": total_budget,
		pre + "net_rows_per_rank": rows_fixed, pre + "net_rows_bound_12N": 12 * N, pre + "row_shortfall_repeats": repeated,
		pre + "hipgraph": graph_state, pre + "graph_launches_per_iteration": launches / max(iters, 1) if use_graph else None,
		pre + "ms_per_iteration_game0_replayed": graph_ms, pre + "ms_per_iteration_game0_eager": eager_ms,
		pre + "allgather_us": ph.get("all_gather", 0.0) * 1e3, pre + "select_us": ph.get("select+expand", 0.0) * 1e3,
		pre + "alltoall_us": ph.get("all_to_all", 0.0) * 1e3, pre + "insert_us": ph.get("insert", 0.0) * 1e3,
		pre + "net_us": ph.get("net", 0.0) * 1e3, pre + "push_us": ph.get("push", 0.0) * 1e3,
		pre + "collectives_seen": collectives,
		pre + "config": f"configs[4] {'weak' if weak else 'strong'}: depth-{depth} scrambles (seeds {seeds}), lambda={lam}, N={N} over all {world} ranks, "
		                f"{total_budget} states, {net_note}, poll={poll}, world={world}, backend={backend if world > 1 else 'local'}; "
		                f"seconds = max over ranks per game; *_us = device time between HIP events, rank 0, second pass over game 0",
	}
	return out


def partitioned_mcts_leg(dist, backend: str, world: int, rank: int, *, trees_per_rank: int = 256, sims: int = 4096, depth: int = 14, c: float = 0.6,
                         net_kind: str = None, force_collectives: bool = False) -> dict:
	"""configs[3], weak: every rank runs its own `trees_per_rank` trees (hipGraph-replayed step); one all-gather of results at the end."""
	from librubiks_amd import cube
	from librubiks_amd.solving.sharded import PartitionedMCTS
	net, fused, net_note = search_net(net_kind)
	trees = trees_per_rank * world
	starts = []
	for i in range(trees):
		np.random.seed(1000 + i)
		starts.append(cube.scramble(depth, True)[0])
	starts = np.array(starts)
	kw = {"fused_first_layer": fused} if fused else {}
	agent = PartitionedMCTS(net, c, trees, capacity=12 * sims + 64, max_path=4096, force_collectives=force_collectives, **kw)
	agent.search(starts, max_sims=16, use_graph=True, poll=8)                                       # pools, GEMM selection, first capture
	torch.cuda.synchronize()
	if dist is not None:
		dist.barrier()
	t0 = time.perf_counter()
	solved = agent.search(starts, max_sims=sims, use_graph=True, poll=64)
	torch.cuda.synchronize()
	dt = _max_over_ranks(time.perf_counter() - t0, dist, backend)
	out = {"pmcts_tree_sims_per_s": float(agent.sims.sum()) / dt, "pmcts_seconds": dt, "pmcts_trees": trees, "pmcts_tree_sims": int(agent.sims.sum()),
	       "pmcts_solved": int(solved.sum()), "pmcts_states": int(agent.states.sum()),
	       "pmcts_config": f"configs[3] weak: {trees_per_rank} trees per rank x {sims} simulations, depth-{depth} scrambles, c={c}, {net_note}, step replayed as a "
	                       f"hipGraph, world={world}; seconds = max over ranks, includes the final all-gather of results"}
	del agent
	torch.cuda.empty_cache()
	return out


def legs(dist, backend: str, world: int, rank: int, *, games: int = 3, sims: int = 4096, trees_per_rank: int = 256, budget: int = STRONG_BUDGET,
         expansions: int = STRONG_N, depth: int = DEPTH, mcts: bool = True, out: dict = None, force_collectives: bool = False) -> dict:
	"""Everything `bench.py --gpus N` adds for N > 1.  A leg that fails says so in `<leg>_error` and the others still run -- but a rank
	that raises inside a collective leaves its peers waiting, so errors are caught per leg on EVERY rank alike (the legs are
	deterministic: what fails on one rank fails on all).  `out` is filled leg by leg, so a caller with a watchdog can report what was finished."""
	out = {} if out is None else out
	for name, fn in (("sharded", lambda: sharded_astar_leg(dist, backend, world, rank, weak=False, games=games, budget=budget, expansions=expansions, depth=depth, force_collectives=force_collectives)),
	                 ("sharded_weak", lambda: sharded_astar_leg(dist, backend, world, rank, weak=True, games=games, budget=budget, expansions=expansions, depth=depth, force_collectives=force_collectives)),
	                 ("pmcts", (lambda: partitioned_mcts_leg(dist, backend, world, rank, trees_per_rank=trees_per_rank, sims=sims, force_collectives=force_collectives)) if mcts else None)):
		if fn is None:
			continue
		try:
			t0 = time.perf_counter()
			out.update(fn())
			out[name + "_leg_seconds"] = time.perf_counter() - t0
		except Exception as e:
			out[name + "_error"] = f"{type(e).__name__}: {e}"[:400]
	return out
