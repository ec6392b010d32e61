"""
ADI data generation on the device (librubiks_amd/adi.py) against a NumPy restatement of the reference's
`Train.ADI_traindata` (librubiks/train.py:256-339) built on the oracle, with the exact-integer stub net.
"""
import numpy as np
import pytest
import torch

from librubiks_amd.adi import adi_traindata
from oracle import cube_oracle as orc
from oracle.search_oracle import StubNet

pytestmark = pytest.mark.gpu


def _expected(games, depth, alpha, method):
	"""train.py:277-332 on the CPU oracle."""
	net = StubNet()
	states, oh_states = orc.sequence_scrambler(games, depth, method == "lapanfix")
	solved_scrambled = orc.multi_is_solved(states)
	sub = orc.expand12(states)
	solved_sub = orc.multi_is_solved(sub)
	rewards = np.where(solved_sub, 0.0 if method == "reward0" else 1.0, -1.0).astype(np.float32)
	values = net(orc.as_oh(sub), policy=False, value=True).reshape(-1) + rewards
	values = values.reshape(-1, 12)
	policy = values.argmax(axis=1)
	value = values[np.arange(len(values)), policy].copy()
	if method == "lapanfix":
		value[solved_scrambled] = 0
	elif method == "schultzfix":
		value[np.arange(0, len(states), depth)] = 0
	w = np.tile(1 / np.arange(1, depth + 1), games)
	u = np.ones_like(w)
	lw = ((1 - alpha) * w / w.sum() + alpha * u / len(u)) * (w.sum() + len(u))
	return oh_states, policy, value, lw.astype(np.float32)


@pytest.mark.parametrize("method", ["lapanfix", "paper", "schultzfix", "reward0"])
def test_adi_matches_reference_algorithm(method):
	games, depth, alpha = 37, 9, 0.3
	np.random.seed(12)
	want = _expected(games, depth, alpha, method)
	np.random.seed(12)
	oh, policy, value, lw = adi_traindata(StubNet(), games, depth, alpha, method, ff_batches=3)
	assert oh.is_cuda and (oh.cpu().numpy() == want[0]).all()
	assert (policy.numpy() == want[1]).all()
	assert (value.numpy() == want[2]).all()
	assert np.allclose(lw.numpy(), want[3], rtol=1e-6, atol=0)


def test_adi_at_training_scale():
	"""configs/main_train.ini: 7 500 games x depth 30 = 225 000 states, 2.7 M children per rollout."""
	np.random.seed(1)
	oh, policy, value, lw = adi_traindata(StubNet(), 7500, 30, 0.5, "lapanfix", ff_batches=8)
	assert oh.shape == (225_000, 480) and policy.shape == (225_000,) and value.shape == (225_000,)
	assert (oh.sum(dim=1) == 20).all()
	# the first state of every game is solved under lapanfix -> target 0; one move away -> reward +1 dominates
	assert (value.view(7500, 30)[:, 0] == 0).all() and (value.view(7500, 30)[:, 1] == 1).all()
