"""
`python bench.py --gpus N` and `python benchmarks/sharded.py --world N` without a launcher (VERDICT r3 #1a): the script starts
its own N ranks -- fresh child processes with the launcher's environment, started before torch or the GPU is touched -- relays
rank 0's JSON line and exits non-zero when any rank does.  No GPU here: bench.py runs its `--dry-run` (rendezvous over gloo,
barrier, MAX over ranks -- the launch path without the kernels); tests/test_bench_contract_gpu.py runs the real thing with two
ranks sharing the GPU.
"""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
	env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
	env["RK_BENCH_BACKEND"] = "gloo"
	return env


def test_bench_starts_its_own_ranks():
	out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "7", "--warmup", "2", "--dry-run"],
	                     capture_output=True, text=True, timeout=300, cwd=ROOT, env=_clean_env())
	assert out.returncode == 0, out.stderr[-2000:]
	lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
	assert len(lines) == 1                                                  # rank 0's line, once
	r = json.loads(lines[0])
	assert r["n_gpus"] == 2 and r["steps"] == 7 and r["warmup"] == 2 and r["dry_run"] is True and r["self_spawned"] is True
	assert abs(r["max_over_ranks_s"] - 0.002) < 1e-9                        # the slowest rank's time, through the all-reduce


def test_bench_under_a_launcher_does_not_spawn_again():
	"""With WORLD_SIZE set (torch.distributed.run did the launching) the script is a rank, and a mismatch with --gpus is an error."""
	env = dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
	out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-run"], capture_output=True, text=True,
	                     timeout=300, cwd=ROOT, env=env)
	assert out.returncode == 0 and json.loads(out.stdout.strip().splitlines()[-1])["self_spawned"] is False
	out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], capture_output=True, text=True,
	                     timeout=300, cwd=ROOT, env=env)
	assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr


def test_a_failing_rank_fails_the_launch(tmp_path):
	"""The parent reports the first failing rank's exit code and does not leave its peers waiting."""
	sys.path.insert(0, ROOT)
	from benchmarks import spawn
	script = tmp_path / "rank.py"
	script.write_text(textwrap.dedent("""
		import os, sys, time
		rank = int(os.environ["RANK"])
		assert os.environ["WORLD_SIZE"] == "3" and os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ["LOCAL_RANK"] == str(rank)
		if rank == 1:
			sys.exit(3)
		if rank == 0:
			print("rank 0 speaking", flush=True)
		time.sleep(60)                  # a peer stuck in a collective: the parent must end it
	"""))
	code = subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {ROOT!r}); from benchmarks import spawn; "
	                       f"sys.exit(spawn.run_ranks({str(script)!r}, [], 3))"], capture_output=True, text=True, timeout=50, env=_clean_env())
	assert code.returncode == 3 and "rank 0 speaking" in code.stdout
	assert spawn.launched_by_someone_else() is ("WORLD_SIZE" in os.environ and "RANK" in os.environ)


def test_parent_imports_nothing_that_touches_the_gpu():
	"""The spawning parent must not have imported torch or the package (whose import asks torch.cuda.is_available()) when it starts
	the ranks: on the GPU pool a process that has touched the GPU may not start programs."""
	probe = textwrap.dedent(f"""
		import sys, runpy
		sys.argv = ["bench.py", "--gpus", "2", "--dry-run"]
		import benchmarks.spawn as spawn
		def fake(script, argv, world, timeout=None):
			bad = [m for m in ("torch", "librubiks_amd", "numpy") if m in sys.modules]
			print("IMPORTED:" + ",".join(bad))
			return 0
		spawn.run_ranks = fake
		try:
			runpy.run_path({os.path.join(ROOT, "bench.py")!r}, run_name="__main__")
		except SystemExit as e:
			print("EXIT:" + str(e.code))
	""")
	out = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, timeout=120, cwd=ROOT, env=_clean_env())
	assert "IMPORTED:\n" in out.stdout + "\n" and "EXIT:0" in out.stdout, out.stdout + out.stderr
