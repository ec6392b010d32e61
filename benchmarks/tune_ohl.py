"""Row groups of the fused first layer (MFMA route with epilogue) at small and medium batches; tuning hook RK_OHL_GROUPS
(tuning build only: python -m librubiks_amd.build --tune)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librubiks_amd import _ffi, cube  # noqa: E402
from benchmarks.kernels import timed  # noqa: E402

TUNE_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "librubiks_hip_tune.so")
_ffi.LIB_PATH = TUNE_LIB if os.path.exists(TUNE_LIB) else sys.exit("build the tuning library first: python -m librubiks_amd.build --tune")
_ffi._lib = None
from librubiks_amd.oh_linear import OhLinear  # noqa: E402

H = 4096
torch.manual_seed(0)
lin = torch.nn.Linear(480, H).cuda().to(torch.bfloat16)
bn = torch.nn.BatchNorm1d(H).cuda().eval()
layer = OhLinear(lin).set_epilogue(torch.nn.ELU(), bn)
g = torch.Generator(device="cuda")
g.manual_seed(0)
for n in (768, 3072, 12_000, 36_000, 120_000, 337_500, 2_700_000):
	states = cube.device.apply_sequences(torch.randint(0, 12, (20, n), device="cuda", dtype=torch.uint8, generator=g), False, True)
	y = torch.empty((n, H), dtype=torch.bfloat16, device="cuda")
	os.environ.pop("RK_OHL_GROUPS", None)
	ref = layer(states).clone()
	for groups in ((0,) if os.environ.get("RK_OHL_RT") else (0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48)):
		if groups:
			os.environ["RK_OHL_GROUPS"] = str(groups)
		else:
			os.environ.pop("RK_OHL_GROUPS", None)
		layer(states, y)
		ok = bool(torch.equal(y, ref))
		t = timed(lambda: layer(states, y), 30)
		print(json.dumps({"rows": n, "groups": groups or "default", "correct": ok, "us": t * 1e6}), flush=True)
