"""bench.py's one-line JSON contract and __graft_entry__.smoke(), run as the driver runs them."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
	out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--no-cpu-baseline"],
	                     capture_output=True, text=True, timeout=300, cwd=ROOT)
	assert out.returncode == 0, out.stderr[-2000:]
	lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
	assert len(lines) == 1
	r = json.loads(lines[0])
	for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
	            "dtype", "data", "config", "roofline"):
		assert key in r, key
	assert r["n_gpus"] == 1 and r["steps"] == 20 and r["warmup"] == 3 and r["higher_is_better"] is True
	assert r["scaling"] == "weak" and r["vs_baseline"] is None and r["dtype"] == "u8" and r["data"] == "synthetic"
	assert "workload" in r["config"] and "model" not in r["config"]
	rf = r["roofline"]
	assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
	assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.2 < rf["frac"] < 1.0
	assert rf["traffic"] is None or 0.9 < rf["traffic"] / 272e6 < 1.5
	# value is consistent with the step time: 1 M parents per step
	assert abs(r["value"] - 1e6 / (r["ms_per_step"] * 1e-3)) / r["value"] < 1e-6
	assert r["value"] > 1e8 / 12            # the north star's floor, in expansions/s
	# every child of the last launch into each of the four output sets was undone and compared on the device
	assert r["verified_children"] == 48_000_000
	# the store schedule in force on this box, and where it came from
	assert r["pace_tau_ps"] in (0, 2000, 2100, 2200, 2400) and r["pace_source"] in ("calibrated on this device", "environment")


def test_smoke_entry_point():
	sys.path.insert(0, ROOT)
	import __graft_entry__
	__graft_entry__.smoke()


def test_bench_two_ranks_without_a_launcher():
	"""VERDICT r3 #1a + r4 #1: `python bench.py --gpus 2` with no launcher starts its own two ranks (benchmarks/spawn.py) -- here sharing
	the one GPU of the box, rendezvous over gloo -- and prints ONE line: the fan-out value is both ranks' work over the slower one's
	time; behind it the line carries the proof that the collectives saw both ranks and the multi-GPU search legs (configs[4] sharded
	A* strong + weak, configs[3] partitioned MCTS).  With the exact stub net and a small budget the sharded leg's searches are ones
	the CPU oracle of the protocol replays: iterations and state counts must be the oracle's at world 2."""
	import numpy as np
	from oracle import cube_oracle as orc
	from oracle.search_oracle import StubNet
	from oracle.sharded_oracle import ShardedAStarOracle
	env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
	env["RK_BENCH_BACKEND"] = "gloo"
	env["RK_BENCH_SEARCH_NET"] = "stub"
	games, budget, N, depth = 2, 60_000, 100, 12
	out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--search-games", str(games),
	                      "--search-budget", str(budget), "--search-expansions", str(N), "--search-depth", str(depth), "--mcts-sims", "64"],
	                     capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
	assert out.returncode == 0, out.stderr[-3000:]
	lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
	assert len(lines) == 1
	r = json.loads(lines[0])
	assert r["n_gpus"] == 2 and r["steps"] == 10 and r["scaling"] == "weak" and "cpu_baseline" not in r
	assert abs(r["value"] - 2 * 1e6 / (r["ms_per_step"] * 1e-3)) / r["value"] < 1e-6
	assert r["config"]["parallelism"] == "independent batches x2"
	assert r["verified_children"] == 48_000_000
	# the collectives saw two ranks
	assert r["collective_backend"] == "gloo" and r["collective_world"] == 2 and r["rank_checksum"] == 3 == r["rank_checksum_expected"]
	assert r["alltoall_verified_on_every_rank"] is True and len(r["rank_devices"]) == 2
	# the legs ran (no *_error key) and carry their keys
	assert not [k for k in r if k.endswith("_error")], {k: r[k] for k in r if k.endswith("_error")}
	for pre in ("sharded_", "sharded_weak_"):
		for key in ("states_per_s", "ms_per_iteration", "allgather_us", "alltoall_us", "net_rows_per_rank", "total_states", "iterations"):
			assert r[pre + key] > 0, pre + key
	assert r["pmcts_tree_sims_per_s"] > 0 and r["pmcts_trees"] == 512 and r["pmcts_tree_sims"] == 512 * 64
	# ... and searched what the protocol's oracle searches at world 2: strong N, weak 2 N and twice the budget
	for pre, n, b in (("sharded_", N, budget), ("sharded_weak_", 2 * N, 2 * budget)):
		states = iters = 0
		for g in range(games):
			np.random.seed(g)
			start, _, _ = orc.scramble(depth, True)
			o = ShardedAStarOracle(StubNet(), 0.16, n, 2)
			o.search(start, b)
			states += o.total_states
			iters += o.iterations
		assert (r[pre + "total_states"], r[pre + "iterations"]) == (states, iters), pre


def test_a_hanging_leg_does_not_lose_the_headline():
	"""The multi-GPU legs run under a watchdog: when they do not finish in time (here: a limit of half a second) every rank leaves and
	rank 0 still prints the ONE line -- the fan-out value, the contract fields -- with `multi_gpu_legs_error` saying what happened."""
	env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
	env.update(RK_BENCH_BACKEND="gloo", RK_BENCH_SEARCH_NET="stub", RK_BENCH_LEGS_TIMEOUT="0.5")
	out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"],
	                     capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
	assert out.returncode == 0, out.stderr[-3000:]
	lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
	assert len(lines) == 1
	r = json.loads(lines[0])
	assert r["n_gpus"] == 2 and r["value"] > 0 and r["verified_children"] == 48_000_000 and "roofline" in r
	assert "did not finish within" in r["multi_gpu_legs_error"] and "sharded_weak_states_per_s" not in r


def test_the_multi_gpu_code_path_over_rccl_with_one_rank():
	"""Two nccl ranks cannot share the one GPU of the test box, so the N > 1 code path of bench.py -- device-buffer collectives, the
	sharded iteration with its RCCL collectives inside the captured graph, the partitioned MCTS's final all-gather -- is rehearsed with
	a process group of ONE rank and the collectives forced (RK_BENCH_FORCE_MULTI=1).  With the exact stub net the sharded leg's searches
	are the protocol oracle's at world 1 (= the reference's A*): same iteration and state counts."""
	import numpy as np
	from oracle import cube_oracle as orc
	from oracle.search_oracle import StubNet
	from oracle.sharded_oracle import ShardedAStarOracle
	env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "RK_BENCH_BACKEND")}
	env.update(RK_BENCH_FORCE_MULTI="1", RK_BENCH_SEARCH_NET="stub")
	games, budget, N, depth = 2, 60_000, 100, 12
	out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--search-games", str(games),
	                      "--search-budget", str(budget), "--search-expansions", str(N), "--search-depth", str(depth), "--mcts-sims", "64"],
	                     capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
	assert out.returncode == 0, out.stderr[-3000:]
	lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
	assert len(lines) == 1
	r = json.loads(lines[0])
	assert r["n_gpus"] == 1 and r["verified_children"] == 48_000_000 and "astar_states_per_s" not in r       # the N = 1 legs give way to the multi-GPU ones
	assert r["collective_backend"] == "nccl" and r["collective_world"] == 1 and r["rank_checksum"] == 1 and r["alltoall_verified_on_every_rank"] is True
	assert r["collective_buffers"] == "device"
	assert not [k for k in r if k.endswith("_error")], {k: r[k] for k in r if k.endswith("_error")}
	assert r["sharded_hipgraph"] in ("replayed", "eager (faster on game 0)") and r["sharded_ms_per_iteration_game0_replayed"] > 0   # RCCL captured
	assert r["sharded_collectives_seen"] > 0 and r["pmcts_tree_sims"] == 256 * 64
	states = iters = 0
	for g in range(games):
		np.random.seed(g)
		start, _, _ = orc.scramble(depth, True)
		o = ShardedAStarOracle(StubNet(), 0.16, N, 1)                     # (= the reference's A* at world 1: tests/test_sharded_oracle_cpu.py)
		o.search(start, budget)
		states += o.total_states
		iters += o.iterations
	assert (r["sharded_total_states"], r["sharded_iterations"]) == (states, iters) == (r["sharded_weak_total_states"], r["sharded_weak_iterations"])
