"""
Kernel times of the scramblers' walks (rk_apply_sequences) at the shapes the reference uses them: sequence_scrambler of a rollout
(7 500 games x 30 rows: a lane per game walks its moves, k_apply_sequences) and single deep scrambles of the evaluation loop (1 game x
100 / 999 moves, 96 games x 100: a wave per game, 64-move chunks composed by a prefix scan, k_apply_sequences_scan).  Run under rocprofv3:

    rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 benchmarks/scramblers.py
    python benchmarks/kernel_trace_by_grid.py DIR --segments --min-calls 20 --skip 5 > profiles/r05_scramblers.csv
"""
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
