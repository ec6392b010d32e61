"""Tuning aid: the per-tile time stamps of the paced fan-out kernel (tuning build only): base read, time at the hold, due time, time after the hold."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librubiks_amd import _ffi, cube  # noqa: E402

N = int(os.environ.get("RK_TUNE_N", "1000000"))
TUNE_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "librubiks_hip_tune.so")
_ffi.LIB_PATH = TUNE_LIB
_ffi._lib = None
lib = _ffi.lib()
lib.rkx_expand12_variant.restype = C.c_int
lib.rkx_expand12_variant.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]
lib.rkx_pace_debug.restype = C.c_int
lib.rkx_pace_debug.argtypes = [C.c_void_p]


def main(code):
	_ffi.check(lib.rk_init(0))
	g = torch.Generator(device="cuda"); g.manual_seed(1)
	ins = [cube.device.apply_sequences(torch.randint(0, 12, (20, N), device="cuda", dtype=torch.uint8, generator=g), False, True) for _ in range(max(6, 640_000_000 // (20 * N)))]
	c, f = cube.device.expand12(ins[0])
	outs = [(torch.empty_like(c), torch.empty_like(f)) for _ in range(4 if N <= 2_000_000 else 2)]
	cell = torch.zeros(4, dtype=torch.int64, device="cuda")
	n_tiles = (N + 63) // 64
	dbg = torch.zeros(n_tiles * 4, dtype=torch.int64, device="cuda")
	for rep in range(6):
		if rep == 5:
			_ffi.check(lib.rkx_pace_debug(dbg.data_ptr()))
		e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
		e0.record()
		cc, ff = outs[rep % len(outs)]
		_ffi.check(lib.rkx_expand12_variant(300, ins[rep].data_ptr(), cc.data_ptr(), ff.data_ptr(), None, N, cell.data_ptr(), code, _ffi.stream_ptr()))
		e1.record(); torch.cuda.synchronize()
		print("launch", rep, "ms", round(e0.elapsed_time(e1), 4), flush=True)
	_ffi.check(lib.rkx_pace_debug(None))
	d = dbg.cpu().view(-1, 4)
	t0 = int(d[:, 1].min())
	rows = []
	phase = int(os.environ.get("RK_PACE_PHASE", "0")) // 4 * 4
	probe = list(range(0, 8)) + list(range(2044, 2052)) + list(range(4096, 4100)) + list(range(8000, 8004))
	if phase and phase < n_tiles:
		for k in (-4096, -2048, -4, 0, 4, 1024, 2044, 2052, 4096, 8192, 16384):
			probe += list(range(phase + k, phase + k + 4))
	probe += list(range(n_tiles - 4, n_tiles))
	for t in probe:
		b, now, due, after = (int(x) for x in d[t])
		rows.append({"tile": t, "base-t0": b - t0, "ready-t0": now - t0, "due-t0": due - t0, "released-t0": after - t0})
	for r in rows:
		print(json.dumps(r))
	late = (d[:, 1] - d[:, 2]).float() / 100.0
	print(json.dumps({"late_us_mean": float(late.mean()), "late_us_max": float(late.max()), "late_us_min": float(late.min()), "frac_late": float((late > 0).float().mean()),
	                  "span_us": (int(d[:, 3].max()) - t0) / 100.0}))


if __name__ == "__main__":
	main(int(sys.argv[1]))
