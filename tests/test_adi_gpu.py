"""
ADI data generation on the device (librubiks_amd/adi.py) against outputs of the UNMODIFIED reference
`Train.ADI_traindata` (librubiks/train.py:256-339) captured by oracle/gen_golden.py with the exact-integer stub net
(tests/golden/adi_trace.npz), and against the oracle restatement at further shapes.
"""
import numpy as np
import pytest
import torch

from librubiks_amd.adi import adi_traindata
from oracle.search_oracle import StubNet, adi_traindata_oracle
from tests.helpers import sha

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("method", ["lapanfix", "paper", "schultzfix", "reward0"])
def test_adi_matches_reference(golden, method):
	"""Against tests/golden/adi_trace.npz: outputs of the UNMODIFIED reference `Train.ADI_traindata` (gen_golden.py)."""
	t = golden["adi_trace"]
	seed, games, depth, ff = (int(x) for x in t[f"{method}_params"])
	np.random.seed(seed)
	oh, policy, value, lw = adi_traindata(StubNet(), games, depth, float(t[f"{method}_alpha"]), method, ff_batches=ff)
	assert oh.is_cuda and oh.dtype == torch.float32
	assert sha(oh.cpu().numpy()) == str(t[f"{method}_oh_sha256"])
	assert (policy.numpy() == t[f"{method}_policy"]).all() and policy.dtype == torch.int64
	assert (value.numpy() == t[f"{method}_value"]).all()
	assert np.allclose(lw.numpy(), t[f"{method}_loss_weights"], rtol=1e-6, atol=0)


@pytest.mark.parametrize("method,games,depth", [("lapanfix", 301, 17), ("schultzfix", 64, 30)])
def test_adi_matches_oracle_at_other_shapes(method, games, depth):
	"""Shapes the fixture does not hold, against the oracle restatement (itself pinned to the fixture on the CPU)."""
	np.random.seed(5)
	want = adi_traindata_oracle(StubNet(), games, depth, 0.5, method)
	np.random.seed(5)
	oh, policy, value, lw = adi_traindata(StubNet(), games, depth, 0.5, method, ff_batches=2)
	assert (oh.cpu().numpy() == want[0]).all() and (policy.numpy() == want[1]).all() and (value.numpy() == want[2]).all()
	assert np.allclose(lw.numpy(), want[3], rtol=1e-6, atol=0)


def test_adi_at_training_scale():
	"""configs/main_train.ini: 7 500 games x depth 30 = 225 000 states, 2.7 M children per rollout."""
	np.random.seed(1)
	oh, policy, value, lw = adi_traindata(StubNet(), 7500, 30, 0.5, "lapanfix", ff_batches=8)
	assert oh.shape == (225_000, 480) and policy.shape == (225_000,) and value.shape == (225_000,)
	assert (oh.sum(dim=1) == 20).all()
	# the first state of every game is solved under lapanfix -> target 0; one move away -> reward +1 dominates
	assert (value.view(7500, 30)[:, 0] == 0).all() and (value.view(7500, 30)[:, 1] == 1).all()


@pytest.mark.parametrize("mode", [False, True, "folded"])
def test_adi_with_a_bf16_net(mode):
	"""A bfloat16 net gets bfloat16 one-hot rows (or the states, with the fused first layer): same one-hot targets as the
	float32 net, values within bf16 rounding of it (tolerance 0.05 on values of order 1)."""
	from benchmarks.nets import FcSmall
	net32 = FcSmall(seed=2).cuda().eval()
	net16 = FcSmall(seed=2).cuda().eval().to(torch.bfloat16)
	np.random.seed(9)
	oh32, p32, v32, _ = adi_traindata(net32, 40, 10, 0.5, "lapanfix", ff_batches=2)
	np.random.seed(9)
	oh16, p16, v16, _ = adi_traindata(net16, 40, 10, 0.5, "lapanfix", ff_batches=3, fused_first_layer=mode)
	assert oh16.dtype == torch.float32 and torch.equal(oh32, oh16)
	assert torch.allclose(v16, v32, atol=0.05) and (p16 == p32).float().mean() > 0.8
