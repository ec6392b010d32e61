"""
Host logic of the fused net (librubiks_amd/oh_linear.py): eval-mode BatchNorm1d as an affine map and its folding into
the following Linear layers, checked on the CPU against the torch modules (reference layer order: Linear, activation,
BatchNorm1d; librubiks/model.py:148-161).  No GPU, no HIP call.
"""
import pytest
import torch

from librubiks_amd.oh_linear import _fold, batchnorm_affine


def _bn(width, seed):
	g = torch.Generator().manual_seed(seed)
	bn = torch.nn.BatchNorm1d(width)
	with torch.no_grad():
		bn.running_mean.copy_(torch.randn(width, generator=g))
		bn.running_var.copy_(torch.rand(width, generator=g) + 0.3)
		bn.weight.copy_(torch.rand(width, generator=g) + 0.5)
		bn.bias.copy_(torch.randn(width, generator=g))
	return bn.eval()


def test_batchnorm_affine_is_the_eval_mode_layer():
	bn = _bn(33, 0).double()
	x = torch.randn(17, 33, dtype=torch.float64)
	scale, shift = batchnorm_affine(bn)
	assert scale.dtype == torch.float32
	assert torch.allclose(bn(x), x * scale.double() + shift.double(), rtol=1e-6, atol=1e-6)
	with pytest.raises(ValueError):
		batchnorm_affine(_bn(4, 1).train())


def test_fold_removes_batchnorm_and_keeps_the_function():
	torch.manual_seed(2)
	shared = [torch.nn.ELU(), _bn(24, 3), torch.nn.Linear(24, 16), torch.nn.ELU(), _bn(16, 4)]          # behind the first Linear
	value = [torch.nn.Linear(16, 8), torch.nn.ELU(), _bn(8, 5), torch.nn.Linear(8, 1)]
	policy = [torch.nn.Linear(16, 8), torch.nn.ELU(), _bn(8, 6), torch.nn.Linear(8, 12)]
	x = torch.randn(50, 24)
	with torch.no_grad():
		h = torch.nn.Sequential(*shared)(x)
		want_v, want_p = torch.nn.Sequential(*value)(h), torch.nn.Sequential(*policy)(h)
		mods, pending = _fold(shared, None)
		assert pending is not None and not any(isinstance(m, torch.nn.BatchNorm1d) for m in mods)
		v, left_v = _fold(value, pending)
		p, left_p = _fold(policy, pending)
		assert left_v is None and left_p is None
		assert not any(isinstance(m, torch.nn.BatchNorm1d) for m in v + p)
		h2 = torch.nn.Sequential(*mods)(x)
		assert torch.allclose(torch.nn.Sequential(*v)(h2), want_v, rtol=1e-4, atol=1e-5)
		assert torch.allclose(torch.nn.Sequential(*p)(h2), want_p, rtol=1e-4, atol=1e-5)
	# the original modules are untouched
	assert isinstance(shared[1], torch.nn.BatchNorm1d) and value[0].weight.shape == (8, 16)


def test_fold_leaves_training_mode_batchnorm_alone():
	bn = _bn(8, 7).train()
	mods, pending = _fold([bn, torch.nn.Linear(8, 4)], None)
	assert pending is None and mods[0] is bn


def test_merged_heads_are_the_two_heads_side_by_side():
	from librubiks_amd.oh_linear import _Affine, _merge_heads
	torch.manual_seed(5)
	pol = [torch.nn.Linear(16, 8), torch.nn.ELU(), torch.nn.Linear(8, 12)]
	val = [torch.nn.Linear(16, 8), torch.nn.ELU(), torch.nn.Linear(8, 1)]
	merged, width = _merge_heads(pol, val)
	x = torch.randn(9, 16)
	with torch.no_grad():
		both = merged(x)
		assert both.shape == (9, 13) and width == 12
		assert torch.allclose(both[:, :12], torch.nn.Sequential(*pol)(x), rtol=1e-5, atol=1e-6)
		assert torch.allclose(both[:, 12:], torch.nn.Sequential(*val)(x), rtol=1e-5, atol=1e-6)
		# the block-diagonal layer has exact zeros off the diagonal
		w = merged[2].weight
		assert (w[:12, 8:] == 0).all() and (w[12:, :8] == 0).all()
	# heads that do not match stay separate
	assert _merge_heads(pol, val[:2]) is None
	assert _merge_heads(pol, [torch.nn.Linear(16, 8), torch.nn.ReLU(), torch.nn.Linear(8, 1)]) is None
	assert _merge_heads(pol, [torch.nn.Linear(20, 8), torch.nn.ELU(), torch.nn.Linear(8, 1)]) is None


def test_tail_fusion_is_for_bfloat16_gpu_stacks_only_and_the_knobs_ship_neutral():
	"""The heads' last layer becomes an `rk_tail_linear` launch only where that kernel is built (bfloat16 on the GPU, K in 512 / 1024 /
	2048, at most 16 outputs, ELU / ReLU in front): any other stack comes back as it is.  The two benchmark knobs ship in their
	neutral positions."""
	import torch
	from librubiks_amd import oh_linear
	assert oh_linear.FUSE_TAIL is True and oh_linear.MFMA_FORM is None
	for mods in ([torch.nn.Linear(2048, 512), torch.nn.ELU(), torch.nn.Linear(512, 12)],                               # float32 on the CPU
	             [torch.nn.Linear(2048, 512).to(torch.bfloat16), torch.nn.ELU(), torch.nn.Linear(512, 12).to(torch.bfloat16)],   # bf16, CPU
	             [torch.nn.Linear(64, 480), torch.nn.Tanh(), torch.nn.Linear(480, 12)],
	             [torch.nn.Linear(512, 12)], []):
		out = oh_linear._fuse_tail(mods)
		assert len(out) == len(mods) and all(a is b for a, b in zip(out, mods))
	assert not oh_linear.TailLinear.fits(torch.nn.ELU(), torch.nn.Linear(512, 12))
	assert oh_linear._act_code(torch.nn.ELU(alpha=0.5)) == (1, 0.5) and oh_linear._act_code(None) == (0, 1.0) and oh_linear._act_code(torch.nn.Tanh()) is None
