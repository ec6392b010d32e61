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


def test_smoke_entry_point():
	sys.path.insert(0, ROOT)
	import __graft_entry__
	__graft_entry__.smoke()


def test_bench_two_ranks_without_a_launcher():
	"""VERDICT r3 #1a: `python bench.py --gpus 2` with no launcher starts its own two ranks (benchmarks/spawn.py) -- here sharing
	the one GPU of the box, rendezvous over gloo -- and prints ONE line whose value is both ranks' work over the slower one's time."""
	env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
	env["RK_BENCH_BACKEND"] = "gloo"
	out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2"],
	                     capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
	assert out.returncode == 0, out.stderr[-3000:]
	lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
	assert len(lines) == 1
	r = json.loads(lines[0])
	assert r["n_gpus"] == 2 and r["steps"] == 10 and r["scaling"] == "weak" and "cpu_baseline" not in r
	assert abs(r["value"] - 2 * 1e6 / (r["ms_per_step"] * 1e-3)) / r["value"] < 1e-6
	assert r["config"]["parallelism"] == "independent batches x2"
