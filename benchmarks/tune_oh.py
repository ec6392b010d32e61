"""A/B of the one-hot kernel's tiling (states per workgroup step, persistent vs one-shot grid) and of its paced form (RK_OH_TAU_PS, RK_OH_LEAD,
RK_OH_NT in the environment; RK_OH_TILES / RK_OH_CAPS select the shapes); tuning hook rkx_as_oh_variant."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librubiks_amd import _ffi, cube  # noqa: E402
from benchmarks.kernels import timed  # noqa: E402

TUNE_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "librubiks_hip_tune.so")   # python -m librubiks_amd.build --tune
_ffi.LIB_PATH = TUNE_LIB if os.path.exists(TUNE_LIB) else sys.exit("build the tuning library first: python -m librubiks_amd.build --tune")
_ffi._lib = None          # drop the shipped library that importing the package loaded
lib = _ffi.lib()
lib.rkx_as_oh_variant.restype = C.c_int
lib.rkx_as_oh_variant.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]

n = 500_000
N_IN = int(os.environ.get("RK_OH_IN_SETS", "1"))       # > 1: the states rotate over that many sets (64 sets = 640 MB: every state from HBM)
g = torch.Generator(device="cuda")
g.manual_seed(1)
ins = [cube.device.apply_sequences(torch.randint(0, 12, (20, n), device="cuda", dtype=torch.uint8, generator=g), False, True) for _ in range(N_IN)]
states = ins[0]
for dt, code, width in ((torch.float32, 0, 1920), (torch.bfloat16, 2, 960)):
	ref = cube.device.as_oh(states, dtype=dt)
	bufs = [torch.empty_like(ref) for _ in range(3)]          # rotate outputs: 2.9 GB (f32) written between reuses
	for tile in [int(x) for x in os.environ.get("RK_OH_TILES", "64,32,16,8").split(",")]:
		for cap in [int(x) for x in os.environ.get("RK_OH_CAPS", "2048,0").split(",")]:
			i = [0]
			def run():
				i[0] += 1
				_ffi.check(lib.rkx_as_oh_variant(tile, cap, ins[i[0] % N_IN].data_ptr(), bufs[i[0] % 3].data_ptr(), code, n, _ffi.stream_ptr()))
			i[0] = N_IN - 1
			run()
			ok = bool(torch.equal(bufs[i[0] % 3], ref))
			t = timed(run, 30)
			print(json.dumps({"dtype": str(dt), "states_per_workgroup": tile, "grid": "persistent 2048" if cap else "one workgroup per tile",
			                  "input_sets": N_IN, "threads_per_workgroup": int(os.environ.get("RK_OH_THREADS", "256")),
			                  "pace_tau_ps": int(os.environ.get("RK_OH_TAU_PS", "0")) if not cap else 0, "nt_stores": os.environ.get("RK_OH_NT", "0") != "0",
			                  "correct": ok, "ms": t * 1e3, "GB/s": round((20 + width) * n / t / 1e9, 1), "frac_of_8TBs": round((20 + width) * n / t / 8e12, 4)}), flush=True)
