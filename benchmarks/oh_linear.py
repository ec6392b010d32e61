"""
f1 A/B on one MI355X: first layer of fc_small (nn.Linear(480, 4096)) from 20-byte states, three ways, at the batch sizes of
the A* iteration (12 k rows), a large A* / MCTS step (120 k) and one ADI feed-forward slice (2.7 M / 8 = 337 500 rows):
  (a) k_as_oh<bf16> + torch bf16 GEMM (what round 1 argued for)     (b) rk_ohl MFMA route, one-hot synthesised in registers
  (c) rk_ohl GATHER route (exact f32, f32 and bf16 output)          (d) k_as_oh<f32> + torch f32 GEMM (the reference's formulation)

    python benchmarks/oh_linear.py > profiles/r02_oh_linear.json
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks.kernels import timed  # noqa: E402
from librubiks_amd import _ffi, cube  # noqa: E402
from librubiks_amd.oh_linear import OhLinear  # noqa: E402

_ffi.check(_ffi.lib().rk_init(0))
H = 4096
torch.manual_seed(0)
lin32 = torch.nn.Linear(480, H).cuda()
lin16 = torch.nn.Linear(480, H).cuda().to(torch.bfloat16)
f32, f16 = OhLinear(lin32), OhLinear(lin16)
# the layer as it stands in the net (model.py:157-159): Linear, ELU, BatchNorm1d in eval mode
elu, bn16, bn32 = torch.nn.ELU(), torch.nn.BatchNorm1d(H).cuda().to(torch.bfloat16).eval(), torch.nn.BatchNorm1d(H).cuda().eval()
e32, e16 = OhLinear(lin32).set_epilogue(elu, bn32), OhLinear(lin16).set_epilogue(elu, bn16.float())
g = torch.Generator(device="cuda")
g.manual_seed(0)
for n in (12_000, 120_000, 337_500, 2_700_000):
	states = cube.device.apply_sequences(torch.randint(0, 12, (20, n), device="cuda", dtype=torch.uint8, generator=g), False, True)
	oh16 = torch.empty((n, 480), dtype=torch.bfloat16, device="cuda")
	y16 = torch.empty((n, H), dtype=torch.bfloat16, device="cuda")
	rows = {"rows": n, "H": H}
	rows["as_oh bf16 + torch bf16 GEMM"] = timed(lambda: torch.nn.functional.linear(cube.device.as_oh(states, oh16, torch.bfloat16), lin16.weight, lin16.bias), 20) * 1e3
	rows["  of which as_oh bf16"] = timed(lambda: cube.device.as_oh(states, oh16, torch.bfloat16), 20) * 1e3
	rows["rk_ohl MFMA bf16 (one-hot in registers)"] = timed(lambda: f16(states, y16, route="mfma"), 20) * 1e3
	rows["rk_ohl GATHER, bf16 weights, bf16 out"] = timed(lambda: f16(states, y16, route="gather"), 20) * 1e3
	with torch.no_grad():
		rows["as_oh bf16 + torch bf16 GEMM + ELU + BatchNorm (torch)"] = timed(lambda: bn16(elu(torch.nn.functional.linear(cube.device.as_oh(states, oh16, torch.bfloat16), lin16.weight, lin16.bias))), 20) * 1e3
		rows["rk_ohl MFMA bf16 + torch ELU + BatchNorm"] = timed(lambda: bn16(elu(f16(states, y16, route="mfma"))), 20) * 1e3
	rows["rk_ohl MFMA bf16 with ELU + BatchNorm epilogue"] = timed(lambda: e16(states, y16, route="mfma"), 20) * 1e3
	if n <= 337_500:
		oh32 = torch.empty((n, 480), dtype=torch.float32, device="cuda")
		y32 = torch.empty((n, H), dtype=torch.float32, device="cuda")
		rows["as_oh f32 + torch f32 GEMM"] = timed(lambda: torch.nn.functional.linear(cube.device.as_oh(states, oh32), lin32.weight, lin32.bias), 20) * 1e3
		rows["rk_ohl GATHER f32 (exact)"] = timed(lambda: f32(states, y32, route="gather"), 20) * 1e3
		with torch.no_grad():
			rows["rk_ohl GATHER f32 + torch ELU + BatchNorm"] = timed(lambda: bn32(elu(f32(states, y32, route="gather"))), 20) * 1e3
		rows["rk_ohl GATHER f32 with ELU + BatchNorm epilogue"] = timed(lambda: e32(states, y32, route="gather"), 20) * 1e3
		del oh32, y32
	rows["unit"] = "ms"
	rows["output_bytes_bf16"] = n * H * 2
	rows["TB/s of bf16 output, MFMA route"] = n * H * 2 / (rows["rk_ohl MFMA bf16 (one-hot in registers)"] * 1e-3) / 1e12
	rows["TFLOP/s dense-equivalent, MFMA route"] = 2 * n * 480 * H / (rows["rk_ohl MFMA bf16 (one-hot in registers)"] * 1e-3) / 1e12
	print(json.dumps(rows), flush=True)
	del oh16, y16
