"""
A short, seeded stretch of the randomised differential run (tests/differential/soak.py): every engine against the CPU oracle, bit for
bit, on randomly drawn cases -- sizes around the paced forms' thresholds and ragged tiles, both representations, the drop-in
NumPy surface and the scramblers' seed parity, A* / MCTS single and batched with stub / misleading / policy nets, pools that
grow on the way, eager and hipGraph-replayed steps, the sharded search on 2-8 simulated ranks.  The long form ran for minutes
(profiles/r04_soak.json); this is its first few cases per kind, so that the draw logic itself stays tested.
"""
import numpy as np
import pytest

from tests.differential import soak

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", list(soak.CASES))
def test_randomly_drawn_cases(kind):
	rng = np.random.RandomState(20260 + sorted(soak.CASES).index(kind))
	for _ in range(3):
		soak.CASES[kind](rng)
