"""
Headline benchmark: cube node-expansions/s (12-child fan-out + goal test) on a 1 M-state batch.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: ONE launch of the fan-out kernel (rk_expand12) on 1 000 000
device-resident parent states (depth-20 random walks from solved), producing 12 000 000 children and their solved
flags.  Inputs are in HBM before the timed region; nothing is copied over PCIe inside it.  With N GPUs every rank
expands its own 1 M-state batch (independent units, no data-path collective): weak scaling, value = all ranks'
expansions / max-over-ranks time.

The timed loop is CACHE-NEUTRAL BY CONSTRUCTION: step i reads parent set i % 32 and writes children/flag set i % 4.
The 32 parent sets are 640 MB of DISTINCT input -- more than twice the 256 MiB Infinity Cache -- so a parent line
cannot still be cached when its set comes round again, whatever the (non-temporal) stores do or do not allocate; the
four output sets are 1 GB, and 756 MB of other stores pass between two writes of a line (MI355X_MICROARCH.md, Infinity
Cache residency rule).  Round 2 rotated the inputs over 4 sets only (80 MB: they could stay cache-resident if the
non-temporal stores bypass the cache); that figure is printed once beside the new one as `kernel_ms_4_input_sets`.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline      algorithmic bytes (272 B per parent) / measured kernel time vs the 8 TB/s HBM peak; `frac` is
                the total-bytes fraction, `frac_read` the read-bytes fraction (20 B per parent: the north star says
                "HBM-read roofline"); kernel time = HIP events around the K back-to-back launches of the timed region,
                plus min / median / mean of a second pass with one event pair per launch
  cpu_baseline  the CPU port of the reference's fan-out idiom (oracle/, NumPy, 1 core) timed on a bounded sample,
                and the C/OpenMP restatement on all host cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
	sys.path.insert(0, ROOT)


def parse_args(argv=None):
	ap = argparse.ArgumentParser()
	ap.add_argument("--gpus", type=int, default=1)
	ap.add_argument("--steps", type=int, default=300)
	ap.add_argument("--warmup", type=int, default=30)
	ap.add_argument("--no-cpu-baseline", action="store_true")
	ap.add_argument("--no-search-legs", action="store_true", help="skip the search legs behind the timed region (N = 1: configs[2] / configs[3]; "
	                "N > 1: configs[4] sharded A* strong + weak and the partitioned MCTS)")
	ap.add_argument("--search-games", type=int, default=3, help="N > 1: games per scaling mode of the sharded A* leg")
	ap.add_argument("--search-budget", type=int, default=0, help="N > 1: state budget of the sharded A* leg at world 1 (default: configs[4]'s 2 M)")
	ap.add_argument("--search-expansions", type=int, default=0, help="N > 1: nodes popped per iteration at world 1 (default: configs[4]'s 700)")
	ap.add_argument("--search-depth", type=int, default=0, help="N > 1: scramble depth of the sharded A* leg (default: configs[4]'s 20)")
	ap.add_argument("--mcts-sims", type=int, default=4096, help="N > 1: simulations per tree of the partitioned MCTS leg (0 skips it)")
	ap.add_argument("--dry-run", action="store_true", help="launch path only: rendezvous, barrier and the MAX over ranks, no GPU work "
	                "(what the CPU test of `--gpus N` without a launcher runs)")
	return ap.parse_args(argv)


if __name__ == "__main__":
	# `python bench.py --gpus N` without a launcher: start the N ranks here, BEFORE torch or anything that touches the GPU is
	# imported (benchmarks/spawn.py: fresh child processes with the launcher's environment, rank 0's JSON line relayed, the
	# exit code of the first rank that fails).  Under torch.distributed.run WORLD_SIZE is set and this is skipped.
	_a = parse_args()
	if _a.gpus > 1 and "WORLD_SIZE" not in os.environ:
		from benchmarks import spawn
		sys.exit(spawn.run_ranks(os.path.abspath(__file__), sys.argv[1:], _a.gpus))

import numpy as np
import torch

N_PARENTS = 1_000_000
BYTES_PER_PARENT = 20 + 240 + 12          # read parent, write 12 children, write 12 solved flags
READ_BYTES_PER_PARENT = 20
N_IN_SETS = 32                            # rotating parent sets: 32 x 20 MB = 640 MB of distinct input (> 2 x 256 MiB)
N_OUT_SETS = 4                            # rotating children/flag sets: 3 x 252 MB pass between two writes of a line
PACED = os.environ.get("RK_PACE", "1") != "0"
KERNEL_PACED, KERNEL_RING = "rk::k_expand12p<true>", "rk::k_expand12r<true, 2, 1, false, 0>"    # the instantiations launch_expand12 picks at 1 M parents (rocprofv3's spelling)
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: 8.0 TB/s spec (about 6.3 TB/s achievable)


def make_parents(n: int, seed: int) -> torch.Tensor:
	"""n depth-20 random walks from the solved state, generated on the device (synthetic data)."""
	from librubiks_amd import cube
	g = torch.Generator(device="cuda")
	g.manual_seed(seed)
	acts = torch.randint(0, 12, (20, n), device="cuda", dtype=torch.uint8, generator=g)
	return cube.device.apply_sequences(acts, with_solved=False, only_last=True)


def _cpu_model() -> str:
	try:
		with open("/proc/cpuinfo") as f:
			for line in f:
				if line.startswith("model name"):
					return line.split(":", 1)[1].strip()
	except OSError:
		pass
	return "unknown"


def _protocol_stats(samples):
	"""The reference's timing protocol (analysis/benchmark.py:38-48, 92-103): drop samples > 2x the mean, then mean and
	the half-width of the 95 % confidence interval of the mean (normal approximation)."""
	x = np.asarray(samples, dtype=np.float64)
	kept = x[x <= 2 * x.mean()]
	mean = float(kept.mean())
	ci = float(1.96 * kept.std(ddof=1) / np.sqrt(len(kept))) if len(kept) > 1 else 0.0
	return mean, ci, len(kept)


def cpu_baseline(sample: int = 1_000_000, numpy_reps: int = 5, native_reps: int = 20):
	"""
	The reference's CPU path, as far as it can travel: the oracle's NumPy port of the fan-out idiom
	`multi_rotate(np.repeat(S, 12, 0), *iter_actions(n))` + `multi_is_solved` (agents.py:277-281, :321) on one
	core, and the C/OpenMP restatement on every host core.  Only this function touches oracle/.
	Protocol of BASELINE.md section 4 / analysis/benchmark.py: fixed batch, repeated calls, input generation
	excluded, samples above twice the mean dropped, mean +- 95 % CI.
	"""
	from oracle import c_oracle, cube_oracle as orc
	rng = np.random.RandomState(1)
	s = orc.repeat_state(orc.SOLVED, sample)
	for _ in range(20):
		a = rng.randint(0, 12, sample)
		s = c_oracle.multi_rotate(s, a.astype(np.uint8), threads=4)
	t_np = []
	for _ in range(numpy_reps):
		t0 = time.perf_counter()
		faces, dirs = orc.iter_actions(sample)
		children = orc.multi_rotate(np.repeat(s, 12, axis=0), faces, dirs)
		flags = orc.multi_is_solved(children)
		t_np.append(time.perf_counter() - t0)
	threads = c_oracle.max_threads()
	out = np.empty((12 * sample, 20), np.int8)
	fl = np.empty(12 * sample, np.uint8)
	c_oracle.expand12(s, threads=threads, out=out, solved=fl)          # warm
	t_c = []
	for _ in range(native_reps):
		t0 = time.perf_counter()
		c_oracle.expand12(s, threads=threads, out=out, solved=fl)
		t_c.append(time.perf_counter() - t0)
	assert (out == children).all() and (fl.astype(bool) == flags).all()
	m_np, ci_np, k_np = _protocol_stats(t_np)
	m_c, ci_c, k_c = _protocol_stats(t_c)
	# flat on purpose: nested objects were dropped by the driver's parser in round 1
	return {
		"value": sample / m_np, "unit": "expansions/s", "cores": 1, "kind": "port",
		"sample": f"{sample} parents (depth-20 walks) x {numpy_reps} reps, NumPy port of the reference fan-out idiom + goal test",
		"seconds_mean": m_np, "seconds_ci95": ci_np, "reps_kept": k_np,
		"value_ci95": [sample / (m_np + ci_np), sample / max(m_np - ci_np, 1e-12)],
		"native_value": sample / m_c, "native_unit": "expansions/s", "native_cores": threads, "native_kind": "port",
		"native_sample": f"{sample} parents x {native_reps} reps, C -O3 + OpenMP restatement",
		"native_seconds_mean": m_c, "native_seconds_ci95": ci_c, "native_reps_kept": k_c,
		"native_value_ci95": [sample / (m_c + ci_c), sample / max(m_c - ci_c, 1e-12)],
		"cpu_model": _cpu_model(), "host_cpus": os.cpu_count(),
		"protocol": "fixed batch, input generation excluded, samples > 2x mean dropped, mean +- 1.96 s/sqrt(n) (analysis/benchmark.py:38-48,92-103)",
	}



def search_legs():
	"""
	BASELINE.json configs[2] and configs[3] on the driver's clock, run once behind the timed region on rank 0 of a
	1-GPU run.  Random-init fc_small in bfloat16, first layer fused with its epilogue, BatchNorm folded (benchmarks/nets.py,
	librubiks_amd/oh_linear.py); one-time costs (pools, GEMM selection, graph capture) stay outside the clocks.
	  A*:   depth-14 scrambles, lambda 0.16, N = 1000, 150 000 states per game, 5 games after one warm-up game.
	        astar_states_per_s / astar_ms_per_iteration: wall clock around the games (synchronised);
	        astar_engine_us_per_iteration / astar_net_share: a second pass over the same games with HIP events between the
	        three parts of every iteration (engine: expand + lookup + append + rows | net | engine: sort + push + bookkeeping) --
	        engine = the two engine parts, net share = net part / whole iteration on the GPU's timeline.
	  MCTS: 256 trees x 4096 simulations, depth-14 scrambles, c = 0.6, step replayed as a hipGraph.
	        mcts_tree_sims_per_s / mcts_ms_per_step: wall clock around the search;
	        mcts_select_us: mean HIP-event time of the backup + select (+ expand ahead) launch over a second, eager run of
	        the same search (events cannot be recorded inside a replayed graph).
	Returns a FLAT dict: the driver's parser keeps flat extra keys and drops nested ones.
	"""
	from benchmarks.nets import FcSmall
	from librubiks_amd import cube
	from librubiks_amd.solving.agents import AStar, MCTSBatch
	net = FcSmall().cuda().eval().to(torch.bfloat16)
	out = {}
	# what an event pair with NOTHING between its two records reads on this box: the part of every event-based figure below
	# that is not kernel time (rocprofv3's kernel averages of the same legs do not contain it)
	pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
	torch.cuda.synchronize()
	for a, b in pairs:
		a.record()
		b.record()
	torch.cuda.synchronize()
	empty_us = sorted(a.elapsed_time(b) for a, b in pairs)[len(pairs) // 2] * 1e3
	out["event_pair_overhead_us"] = empty_us
	# ---- the cube part of one ADI rollout (ref:train.py:277-292) at the reference's size, 7 500 games x 30: walks, goal tests and fan-out
	#      in ONE launch (rk_rollout_fanout; before / after against the three launches it replaced: benchmarks/adi_cube.py, profiles/r05_adi_cube.json)
	acts = torch.randint(0, 12, (30, 7500), device="cuda", dtype=torch.uint8)
	for _ in range(3):
		cube.device.rollout_fanout(acts, True)
	ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	torch.cuda.synchronize()
	ea.record()
	for _ in range(50):
		cube.device.rollout_fanout(acts, True)
	eb.record()
	torch.cuda.synchronize()
	out["adi_cube_us"] = ea.elapsed_time(eb) / 50 * 1e3
	out["adi_cube_config"] = "7 500 games x 30 rows (225 000 states, 2.7 M children, 61 MB written): rk_rollout_fanout through cube.device (torch allocates the outputs inside the clock)"
	del acts
	# ---- configs[2]: A* ------------------------------------------------------------------------------------------
	lam, N, depth, budget, games = 0.16, 1000, 14, 150_000, 5
	agent = AStar(net, lam, N, fused_first_layer="folded")
	np.random.seed(12345)
	agent.search(cube.scramble(depth, True)[0], time_limit=None, max_states=40 * N)           # warm-up
	starts = []
	for g in range(games):
		np.random.seed(g)
		starts.append(cube.scramble(depth, True)[0])
	tot_t = tot_states = tot_iter = solved = 0
	for st in starts:
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		solved += bool(agent.search(st, time_limit=None, max_states=budget))
		torch.cuda.synchronize()
		tot_t += time.perf_counter() - t0
		tot_states += len(agent)
		tot_iter += agent.iterations
	eng = netp = whole = 0.0
	n_it = 0
	for st in starts:
		agent.profile_events = []
		agent.search(st, time_limit=None, max_states=budget)
		torch.cuda.synchronize()
		for a, b, c, d in agent.profile_events:
			eng += a.elapsed_time(b) + c.elapsed_time(d)
			netp += b.elapsed_time(c)
			whole += a.elapsed_time(d)
		n_it += len(agent.profile_events)
	agent.profile_events = None
	out.update({
		"astar_states_per_s": tot_states / tot_t, "astar_ms_per_iteration": tot_t / max(tot_iter, 1) * 1e3,
		"astar_engine_us_per_iteration": eng / max(n_it, 1) * 1e3, "astar_net_share": netp / max(whole, 1e-12),
		"astar_engine_us_per_iteration_less_event_overhead": eng / max(n_it, 1) * 1e3 - 2 * empty_us,       # two event pairs per iteration
		"astar_iterations": tot_iter, "astar_states": tot_states, "astar_games": games, "astar_solved": solved,
		"astar_config": f"configs[2]: depth-{depth} scrambles, lambda={lam}, N={N}, {budget} states per game, fc_small bf16 random init, first layer fused + folded, heads' last layer fused (rk_tail_linear)",
	})
	del agent
	# ---- configs[3]: MCTS -----------------------------------------------------------------------------------------
	T, sims, c = 256, 4096, 0.6
	starts = []
	for g in range(T):
		np.random.seed(g)
		starts.append(cube.scramble(depth, True)[0])
	starts = np.array(starts)
	cap = 12 * sims + 64
	def run_mcts(overlap):
		trees = MCTSBatch(net, c, T, capacity=cap, max_path=16384, fused_first_layer="folded", overlap_halves=overlap)
		trees.search(starts, max_states=cap, max_sims=16, use_graph=True, poll=8)                     # pools, GEMM selection, first capture
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		ok = trees.search(starts, max_states=cap, max_sims=sims, use_graph=True, poll=64)
		torch.cuda.synchronize()
		return trees, ok, time.perf_counter() - t0

	# the step on ONE stream (round 4's form) ...
	trees, ok, dt = run_mcts(False)
	status = trees.status
	steps = trees.simulations
	trees.profile_events = []
	trees.search(starts, max_states=cap, max_sims=sims, use_graph=False, poll=64)
	torch.cuda.synchronize()
	sel = [a.elapsed_time(b) for a, b in trees.profile_events]
	trees.profile_events = None
	del trees
	torch.cuda.empty_cache()
	# ... and as two halves on two streams, skewed by half a step: one half's backup + descent under the other half's net forward
	trees2, ok2, dt2 = run_mcts(True)
	status2, steps2 = trees2.status, trees2.simulations
	del trees2
	best_dt, best_status, best_steps, best_ok = (dt2, status2, steps2, ok2) if dt2 / max(steps2, 1) < dt / max(steps, 1) else (dt, status, steps, ok)
	out.update({
		"mcts_tree_sims_per_s": float(best_status[:, 3].sum()) / best_dt, "mcts_ms_per_step": best_dt / max(best_steps, 1) * 1e3,
		"mcts_ms_per_step_one_stream": dt / max(steps, 1) * 1e3, "mcts_ms_per_step_two_halves": dt2 / max(steps2, 1) * 1e3,
		"mcts_step_form": "two halves on two streams" if best_dt is dt2 else "one stream",
		"mcts_select_us": sum(sel) / max(len(sel), 1) * 1e3, "mcts_select_us_less_event_overhead": sum(sel) / max(len(sel), 1) * 1e3 - empty_us,
		"mcts_steps": best_steps, "mcts_tree_sims": int(best_status[:, 3].sum()),
		"mcts_solved": int(best_ok.sum()),
		"mcts_config": f"configs[3]: {T} trees x {sims} simulations, depth-{depth} scrambles, c={c}, fc_small bf16 random init, first layer fused + folded, heads' last layer fused (rk_tail_linear), step replayed as a hipGraph "
		               "(mcts_ms_per_step = the faster of the one-stream step and the two-halves step, both timed here: mcts_ms_per_step_one_stream / _two_halves)",
	})
	return out


PMC_FILE = os.path.join("profiles", "r05_expand12_pmc.json")


def pmc_traffic(kernel):
	"""
	HBM bytes per launch from the committed rocprofv3 PMC passes of THIS command (profiles/, written by
	benchmarks/pmc_summary.py from separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs; FETCH_SIZE doubled per the
	guide's gfx950 correction).  PMC counters cannot be collected inside a timed run, so the figure is a record with its
	provenance; it is null when the record is missing or was taken for a different kernel instantiation.
	"""
	try:
		with open(os.path.join(ROOT, PMC_FILE)) as f:
			rec = json.load(f)
	except (OSError, ValueError):
		return None, None
	if rec.get("kernel") != kernel:
		return None, {"source": PMC_FILE, "stale": f"recorded for {rec.get('kernel')}"}
	return rec.get("hbm_bytes_per_launch"), {"source": PMC_FILE, "kernel": rec.get("kernel"), "commit": rec.get("commit"),
	                                         "fetch_bytes": rec.get("fetch_bytes_per_launch"), "write_bytes": rec.get("write_bytes_per_launch"),
	                                         "reading": "fabric-side bytes (FETCH_SIZE / WRITE_SIZE count Infinity-Cache hits like HBM accesses): the paced kernel "
	                                                    "fetches every parent twice over the fabric by design -- its read phase from HBM, then the expanding wave "
	                                                    "from the Infinity Cache -- so fetch = 2 x 20 B per parent where HBM serves 1 x 20 B; writes are exact"}


def max_over_ranks(elapsed: float, dist, device) -> float:
	"""The slowest rank's time: what divides the whole job's units (weak scaling, no data-path collective)."""
	t = torch.tensor([elapsed], dtype=torch.float64, device=device)
	if dist is not None:
		dist.all_reduce(t, op=dist.ReduceOp.MAX)
	return float(t.item())


def dry_run(args, rank, world):
	"""The launch path without the GPU: process group over gloo, the barrier and the MAX-over-ranks reduction of the timed
	region, one JSON line from rank 0.  No kernel runs, `value` is null: this checks that N ranks start, meet and report."""
	dist = None
	if world > 1:
		import torch.distributed as dist
		dist.init_process_group("gloo")
		dist.barrier()
	t = max_over_ranks(0.001 * (rank + 1), dist, torch.device("cpu"))
	if rank == 0:
		print(json.dumps({"metric": "cube node-expansions/sec (12-child fan-out) at 1M-state batch", "value": None, "dry_run": True,
		                  "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "max_over_ranks_s": t,
		                  "self_spawned": os.environ.get("RK_SELF_SPAWNED") == "1"}), flush=True)
	if dist is not None:
		dist.barrier()
		dist.destroy_process_group()


def main():
	args = parse_args()

	rank = int(os.environ.get("RANK", "0"))
	local_rank = int(os.environ.get("LOCAL_RANK", "0"))
	world = int(os.environ.get("WORLD_SIZE", "1"))
	if world != args.gpus:
		sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node and --gpus must agree")
	if args.dry_run:
		return dry_run(args, rank, world)
	# One rank per GPU.  RK_BENCH_BACKEND=gloo is a rehearsal mode for a one-GPU box (several ranks share cuda:0 and
	# the only collectives -- barrier and the MAX of the elapsed times -- go through the host); the driver uses nccl (= RCCL).
	backend = os.environ.get("RK_BENCH_BACKEND", "nccl")
	device_index = local_rank % max(1, torch.cuda.device_count())
	torch.cuda.set_device(device_index)
	# RK_BENCH_FORCE_MULTI=1 with --gpus 1: a process group of ONE rank over RCCL, and the multi-GPU legs with their collectives forced --
	# the N > 1 code path of this file (device-buffer collectives, the captured sharded iteration, the final all-gather of the
	# partitioned MCTS) rehearsed on a one-GPU box, where two nccl ranks cannot share the GPU.  n_gpus stays 1.
	force_multi = world == 1 and os.environ.get("RK_BENCH_FORCE_MULTI", "") == "1"
	if force_multi:
		backend = "nccl"
	dist = None
	if world > 1 or force_multi:
		import torch.distributed as dist
		if force_multi:
			import tempfile
			store = os.path.join(tempfile.mkdtemp(prefix="rk_bench_"), "rendezvous")
			dist.init_process_group("nccl", init_method=f"file://{store}", rank=0, world_size=1, device_id=torch.device("cuda", device_index))
		elif backend == "nccl":
			dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
		else:
			dist.init_process_group(backend)
	reduce_device = torch.device("cuda") if backend == "nccl" else torch.device("cpu")

	from librubiks_amd import _ffi, cube
	_ffi.check(_ffi.lib().rk_init(device_index))

	ins = [make_parents(N_PARENTS, seed=1000 + 64 * rank + k) for k in range(N_IN_SETS)]
	outs = [(torch.empty((12 * N_PARENTS, 20), dtype=torch.int8, device="cuda"),
	         torch.empty(12 * N_PARENTS, dtype=torch.uint8, device="cuda")) for _ in range(N_OUT_SETS)]
	stats = torch.tensor([0, _ffi.INT64_MAX], dtype=torch.int64, device="cuda")
	counter = [0]
	last_in = {}                               # output set -> parent set last expanded into it (for the sanity check)
	in_sets = [N_IN_SETS]                      # the comparison pass after the timed region narrows the rotation to round 2's 4 sets

	def step():
		i = counter[0]
		counter[0] += 1
		children, solved = outs[i % N_OUT_SETS]
		last_in[i % N_OUT_SETS] = i % in_sets[0]
		cube.device.expand12(ins[i % in_sets[0]], children, solved, stats)

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

	def back_to_back(n_launches):
		e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
		torch.cuda.synchronize()
		e0.record()
		for _ in range(n_launches):
			step()
		e1.record()
		torch.cuda.synchronize()
		return e0.elapsed_time(e1) / n_launches

	# for the record (outside the timed region): round 2's rotation -- 4 parent sets = 80 MB, which the Infinity Cache could hold
	in_sets[0] = 4
	back_to_back(8)
	kernel_ms_4in = back_to_back(max(40, min(args.steps, 300)))
	in_sets[0] = N_IN_SETS

	# for the record (outside the timed region): the UNPACED ring form on this very box, same rotation -- a box on which the
	# paced form's 2.10 ns/tile schedule does not hold shows up as a ratio near 1 instead of as an unexplained lower fraction
	kernel_ms_ring = None
	if PACED:
		_ffi.check(_ffi.lib().rk_set_pacing(0))
		back_to_back(8)
		kernel_ms_ring = back_to_back(max(40, min(args.steps, 300)))
		_ffi.check(_ffi.lib().rk_set_pacing(-1))

	# per-launch distribution (outside the timed region): one event pair per launch, same rotation
	per = []
	evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(args.steps, 200))]
	for e0, e1 in evs:
		e0.record()
		step()
		e1.record()
	torch.cuda.synchronize()
	per = sorted(e0.elapsed_time(e1) for e0, e1 in evs)

	# The timed work, verified IN FULL (outside the timed region): for every output set, ALL 12 M children of the last launch
	# into it are moved back by the inverse action and compared with the 1 M parents that launch read -- on the device, in
	# slices of 1 M children (one rk_multi_rotate + one comparison each); the flags are checked against a goal test of the children.
	undo_slice = torch.arange(12, dtype=torch.uint8, device="cuda").bitwise_xor(1).repeat(1 << 16)
	verified_children = 0
	for o, k in sorted(last_in.items()):
		parents, (children, solved) = ins[k], outs[o]
		for lo in range(0, N_PARENTS, 1 << 16):
			hi = min(lo + (1 << 16), N_PARENTS)
			back = cube.device.multi_rotate(children[12 * lo:12 * hi], undo_slice[:12 * (hi - lo)])
			if not torch.equal(back.view(hi - lo, 12, 20), parents[lo:hi].view(hi - lo, 1, 20).expand(hi - lo, 12, 20)):
				raise AssertionError(f"output set {o}: children of parents {lo}..{hi} are not the fan-out of parent set {k}")
			del back
		if not torch.equal(cube.device.multi_is_solved(children), solved):
			raise AssertionError(f"output set {o}: solved flags differ from a goal test of the children")
		verified_children += 12 * N_PARENTS
	assert int(stats[0]) == 0 or int(stats[1]) < 12 * N_PARENTS
	import ctypes
	tau_c, src_c, us_c = ctypes.c_uint(0), ctypes.c_int(0), (ctypes.c_float * 5)()
	_ffi.check(_ffi.lib().rk_get_pacing(tau_c, src_c, us_c))                    # the store schedule rk_init settled on for this device
	paced = PACED and tau_c.value > 0
	kernel = KERNEL_PACED if paced else KERNEL_RING

	if world > 1 or force_multi:
		del ins, outs
		torch.cuda.empty_cache()

	if rank == 0:
		value = world * N_PARENTS * args.steps / elapsed_max
		achieved = BYTES_PER_PARENT * N_PARENTS / (kernel_ms * 1e-3) / 1e9
		achieved_read = READ_BYTES_PER_PARENT * N_PARENTS / (kernel_ms * 1e-3) / 1e9
		traffic, traffic_src = pmc_traffic(kernel)
		line = {
			"metric": "cube node-expansions/sec (12-child fan-out) at 1M-state batch",
			"value": value, "unit": "expansions/s",
			"n_gpus": world, "steps": args.steps, "warmup": args.warmup,
			"ms_per_step": elapsed_max / args.steps * 1e3,
			"higher_is_better": True, "scaling": "weak", "vs_baseline": None,
			"dtype": "u8", "data": "synthetic",
			"config": {"workload": f"configs[1]: {world}xMI355X fan-out (12 moves) + is_solved on 1M depth-20 scrambles per GPU, "
			                       f"device-resident, one rk_expand12 launch per step, parents rotating over {N_IN_SETS} sets "
			                       f"({N_IN_SETS * READ_BYTES_PER_PARENT * N_PARENTS // 1_000_000} MB of distinct input, > 2 x the 256 MiB Infinity Cache) and "
			                       f"children/flags over {N_OUT_SETS} sets ({N_OUT_SETS * (BYTES_PER_PARENT - READ_BYTES_PER_PARENT) * N_PARENTS // 1_000_000} MB)",
			           "parents_per_gpu": N_PARENTS, "children_per_step": 12 * N_PARENTS * world,
			           "input_sets": N_IN_SETS, "output_sets": N_OUT_SETS,
			           "parallelism": f"independent batches x{world}"},
			"transitions_per_s": 12 * value,
			"verified_children": verified_children,          # every child of the last launch into each output set, undone and compared on the device
			"pace_tau_ps": int(tau_c.value), "pace_source": {0: "compiled default", 1: "calibrated on this device", 2: "environment"}.get(int(src_c.value), "?"),
			"pace_calibration_us": {"ring": us_c[0], "2000": us_c[1], "2100": us_c[2], "2200": us_c[3], "2400": us_c[4]},
			"roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
			             "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_provenance": traffic_src,
			             "achieved_read": achieved_read, "frac_read": achieved_read / HBM_PEAK_GBS,
			             "kernel": kernel, "kernel_form": f"paced: a read phase (parents -> Infinity Cache), then one 64-parent tile per wave stored on a {tau_c.value / 1000:.2f} ns/tile "
			                                    "schedule (DESIGN 3 step 3; RK_PACE=0 runs the unpaced ring form)" if paced else "ring form (RK_PACE=0, or the calibration chose it)",
			             "kernel_ms_back_to_back": kernel_ms,
			             "per_launch_event_pairs_ms": {"min": per[0], "median": per[len(per) // 2], "mean": sum(per) / len(per)},
			             "kernel_ms_4_input_sets": kernel_ms_4in, "frac_4_input_sets": BYTES_PER_PARENT * N_PARENTS / (kernel_ms_4in * 1e-3) / 1e9 / HBM_PEAK_GBS,
			             "kernel_ms_ring_same_box": kernel_ms_ring,
			             "frac_ring_same_box": (BYTES_PER_PARENT * N_PARENTS / (kernel_ms_ring * 1e-3) / 1e9 / HBM_PEAK_GBS) if kernel_ms_ring else None,
			             "algorithmic_bytes_per_launch": BYTES_PER_PARENT * N_PARENTS,
			             "algorithmic_read_bytes_per_launch": READ_BYTES_PER_PARENT * N_PARENTS,
			             "cache_neutral": True},
		}
		if kernel_ms_ring:
			# flat copies: the driver's parser keeps flat extra keys
			line["frac_ring_same_box"] = line["roofline"]["frac_ring_same_box"]
			line["paced_over_ring_same_box"] = kernel_ms_ring / kernel_ms
		if world == 1 and not force_multi and not args.no_search_legs:
			del ins, outs                                     # 1.6 GB back to the allocator before the pools of the search legs
			torch.cuda.empty_cache()
			try:
				line.update(search_legs())
			except Exception as e:                            # the headline line must not be lost to a leg: say what failed and go on
				line["search_legs_error"] = f"{type(e).__name__}: {e}"[:400]
		if world == 1 and not args.no_cpu_baseline:
			try:
				line["cpu_baseline"] = cpu_baseline()
			except Exception as e:
				line["cpu_baseline"] = {"value": None, "unit": "expansions/s", "cores": 0, "kind": "port", "sample": "failed", "error": f"{type(e).__name__}: {e}"[:400]}
	else:
		line = None

	# N > 1: the collectives' proof and the search legs are COLLECTIVE -- every rank runs them, rank 0 reports.  They run behind the
	# headline's line, under a watchdog: should a leg hang (a collective that never completes on hardware the build never saw), every
	# rank leaves after RK_BENCH_LEGS_TIMEOUT seconds and rank 0 still prints the line with what was finished -- the fan-out number of
	# the scaling run is never lost to a leg.
	if world > 1 or force_multi:
		import threading
		from benchmarks import multi_gpu
		multi, finished = {}, threading.Event()
		limit = float(os.environ.get("RK_BENCH_LEGS_TIMEOUT", "240"))

		def bail():
			if finished.is_set():
				return
			if rank == 0:
				line.update(multi)
				line["multi_gpu_legs_error"] = f"the multi-GPU legs did not finish within {limit:.0f} s (RK_BENCH_LEGS_TIMEOUT); keys above are the legs that did"
				print(json.dumps(line), flush=True)
			os._exit(0)

		watchdog = threading.Timer(limit, bail)
		watchdog.daemon = True
		watchdog.start()
		try:
			multi.update(multi_gpu.collective_proof(dist, backend))
		except Exception as e:
			multi["collective_proof_error"] = f"{type(e).__name__}: {e}"[:400]
		if not args.no_search_legs:
			multi_gpu.legs(dist, backend, world, rank, games=args.search_games, sims=args.mcts_sims or 4096, mcts=args.mcts_sims > 0,
			               budget=args.search_budget or multi_gpu.STRONG_BUDGET, expansions=args.search_expansions or multi_gpu.STRONG_N,
			               depth=args.search_depth or multi_gpu.DEPTH, out=multi, force_collectives=force_multi)
		finished.set()
		watchdog.cancel()
		if rank == 0:
			line.update(multi)
	if rank == 0:
		print(json.dumps(line), flush=True)
	if dist is not None:
		dist.barrier()
		dist.destroy_process_group()


if __name__ == "__main__":
	main()
