"""
One process per GPU without a launcher: `python bench.py --gpus N` (and `python benchmarks/sharded.py --world N`) start their
own N ranks when nobody else did (WORLD_SIZE unset).

The PARENT does nothing but start children: it imports neither torch nor librubiks_amd and makes no HIP call (on the GPU
pool a process that has touched the GPU must not start or replace programs), never re-execs itself, and exits non-zero when
any rank does.  Every child is a fresh interpreter running the same script with the same arguments and the environment
`torch.distributed.run` would have given it (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR = 127.0.0.1, MASTER_PORT), so the
script's rank code path is exactly the one the driver's launcher exercises.  stdout of rank 0 is relayed (that is where the
JSON lines are printed); the other ranks' stdout is dropped, every rank's stderr goes to the parent's stderr.
"""
import os
import socket
import subprocess
import sys
import time


def launched_by_someone_else() -> bool:
	return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port() -> int:
	with socket.socket() as s:
		s.bind(("127.0.0.1", 0))
		return s.getsockname()[1]


def run_ranks(script: str, argv: list, world: int, timeout: float = None) -> int:
	"""Start `world` ranks of `script argv...`, relay rank 0's stdout, return 0 or the first non-zero exit code."""
	port = free_port()
	procs = []
	for rank in range(world):
		env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
		           MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RK_SELF_SPAWNED="1")
		env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: RCCL across processes needs it on this pool
		procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env,
		                              stdout=None if rank == 0 else subprocess.DEVNULL))
	t0, rc = time.monotonic(), 0
	live = list(procs)
	while live:
		for p in list(live):
			code = p.poll()
			if code is None:
				continue
			live.remove(p)
			if code != 0 and rc == 0:
				rc = code if 0 < code < 256 else 1
				for q in live:                                      # a dead rank leaves its peers in a collective for ever
					q.terminate()
		if timeout is not None and time.monotonic() - t0 > timeout:
			for q in live:
				q.kill()
			return rc or 124
		time.sleep(0.05)
	return rc
