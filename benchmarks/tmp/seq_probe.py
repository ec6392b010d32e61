import torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librubiks_amd import cube, _ffi
_ffi.check(_ffi.lib().rk_init(0))
for games, depth in ((7500, 30), (1, 999), (1, 100), (96, 100)):
    acts = torch.randint(0, 12, (depth, games), device="cuda", dtype=torch.uint8)
    for only_last in (False, True):
        for _ in range(60):
            cube.device.apply_sequences(acts, depth <= 64 and not only_last, only_last)
torch.cuda.synchronize()
