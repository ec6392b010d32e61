"""
The per-row kernels on a row count that is NOT a multiple of the 256-row tile (250 000) against one that is (261 120), cache-neutral:
what the ragged last tile costs.  Run under `rocprofv3 --kernel-trace` and summarise with kernel_trace_by_grid.py --segments.
Round 4 found the ragged tile's twenty predicated dword loads compiled to twenty dependent round trips (multi_is_solved: 6.8 us on
250 000 rows against 3.4 us on 261 120); with clamped addresses they are issued back to back.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librubiks_amd import _ffi, cube  # noqa: E402

_ffi.check(_ffi.lib().rk_init(0))
g = torch.Generator(device="cuda")
g.manual_seed(1)
POOL = 33_600_000
pool = torch.empty((POOL, 20), dtype=torch.int8, device="cuda")
for lo in range(0, POOL, 4_200_000):
	pool[lo:lo + 4_200_000] = cube.device.apply_sequences(torch.randint(0, 12, (6, 4_200_000), device="cuda", dtype=torch.uint8, generator=g), False, True)
acts = torch.randint(0, 12, (POOL,), device="cuda", dtype=torch.uint8, generator=g)
flags = torch.empty(300_000, dtype=torch.uint8, device="cuda")
out = torch.empty((300_000, 20), dtype=torch.int8, device="cuda")
stats = torch.tensor([0, _ffi.INT64_MAX], dtype=torch.int64, device="cuda")
at = 0
for n in (261_120, 250_000, 100_000, 99_840):
	for fn in (lambda a, n: cube.device.multi_is_solved(pool[a:a + n], flags[:n]),
	           lambda a, n: cube.device.multi_rotate(pool[a:a + n], acts[a:a + n], out=out[:n]),
	           lambda a, n: cube.device.multi_rotate_solved(pool[a:a + n], acts[a:a + n], out=out[:n], flags=flags[:n], stats=stats)):
		for i in range(120):
			if at + n > POOL:
				at = 0
			fn(at, n)
			at += n
		torch.cuda.synchronize()
