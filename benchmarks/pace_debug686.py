"""Tuning aid: per-group time stamps of the paced 6x8x6 fan-out (tuning build; RK_PACE_PHASE=65536 keeps 200 k parents in one phase)."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librubiks_amd import _ffi, cube  # noqa: E402

TUNE_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "librubiks_hip_tune.so")
_ffi.LIB_PATH = TUNE_LIB
_ffi._lib = None
lib = _ffi.lib()
lib.rkx_pace_debug.restype = C.c_int
lib.rkx_pace_debug.argtypes = [C.c_void_p]
_ffi.check(lib.rk_init(0))
cube.set_is2024(False)
n = 200_000
solved = torch.from_numpy(cube.get_solved()).cuda()
g = torch.Generator(device="cuda"); g.manual_seed(0)
ins = []
for k in range(12):
	s = solved.unsqueeze(0).repeat(n, 1, 1, 1).contiguous()
	for _ in range(6):
		s = cube.device.multi_rotate(s, torch.randint(0, 12, (n,), device="cuda", dtype=torch.uint8, generator=g))
	ins.append(s)
children = [torch.empty((12 * n, 6, 8, 6), dtype=torch.int8, device="cuda") for _ in range(2)]
flags = torch.empty(12 * n, dtype=torch.uint8, device="cuda")
n_groups = (n + 3) // 4
dbg = torch.zeros(n_groups * 4, dtype=torch.int64, device="cuda")
for rep in range(6):
	if rep == 5:
		_ffi.check(lib.rkx_pace_debug(dbg.data_ptr()))
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	cube.device.expand12(ins[rep], children[rep % 2], flags)
	e1.record(); torch.cuda.synchronize()
	print("launch", rep, "ms", round(e0.elapsed_time(e1), 4), flush=True)
_ffi.check(lib.rkx_pace_debug(None))
d = dbg.cpu().view(-1, 4)
t0 = int(d[:, 1].min())
for t in list(range(0, 4)) + list(range(2046, 2050)) + list(range(4096, 4098)) + list(range(10000, 10002)) + list(range(30000, 30002)) + list(range(n_groups - 2, n_groups)):
	b, now, due, after = (int(x) for x in d[t])
	print(json.dumps({"group": t, "base-t0": b - t0, "ready-t0": now - t0, "due-t0": due - t0, "released-t0": after - t0}))
late = (d[:, 1] - d[:, 2]).float() / 100.0
print(json.dumps({"late_us_mean": float(late.mean()), "late_us_max": float(late.max()), "late_us_min": float(late.min()), "frac_late": float((late > 0).float().mean()),
                  "span_us": (int(d[:, 3].max()) - t0) / 100.0}))
