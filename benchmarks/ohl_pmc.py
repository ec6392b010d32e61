"""The fused first layer (MFMA route, no epilogue) at 2.7 M rows, a few launches: the dispatch rocprofv3's PMC passes look at
(GRBM_GUI_ACTIVE -> effective clock under load, SQ_VALU_MFMA_BUSY_CYCLES -> matrix-pipe busy cycles)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librubiks_amd import cube  # noqa: E402
from librubiks_amd.oh_linear import OhLinear  # noqa: E402

torch.manual_seed(0)
lin = torch.nn.Linear(480, 4096).cuda().to(torch.bfloat16)
layer = OhLinear(lin)
n = 2_700_000
g = torch.Generator(device="cuda")
g.manual_seed(0)
states = cube.device.apply_sequences(torch.randint(0, 12, (20, n), device="cuda", dtype=torch.uint8, generator=g), False, True)
y = torch.empty((n, 4096), dtype=torch.bfloat16, device="cuda")
for _ in range(6):
	layer(states, y, route="mfma")
torch.cuda.synchronize()
