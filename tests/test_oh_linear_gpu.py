"""
Fused one-hot -> first Linear (rk_ohl_*, librubiks_amd/oh_linear.py) against `torch.nn.functional.linear(as_oh(x), W, b)`
in float32 (reference: cube.py:265-277 + model.py:127,150).

Tolerances (floating point, stated here as the scope contract asks):
  * GATHER route, float32: the kernel sums ((b + w_0) + w_1) + ... + w_19 with float32 adds in that order, so it is
    BIT-EXACT against the same sequence of float32 adds done by torch; against F.linear (whose GEMM adds the 20 non-zero
    terms in another order) the difference is rounding of a 21-term float32 sum: |diff| <= 2e-6 * (|b| + sum |w_i|).
  * MFMA route, bf16: weights rounded to bf16, float32 accumulation, result rounded to bf16 (nearest even): within one
    bf16 unit in the last place of the float32 reference computed from the SAME bf16 weights: rtol 2^-7, atol 1e-6.
"""
import numpy as np
import pytest
import torch

from librubiks_amd import cube
from librubiks_amd.oh_linear import OhLinear, fuse_first_linear
from oracle import cube_oracle as orc
from tests.helpers import random_walk

pytestmark = pytest.mark.gpu


def _layer(H, dtype=torch.float32, seed=0):
	torch.manual_seed(seed)
	lin = torch.nn.Linear(480, H)
	torch.nn.init.xavier_uniform_(lin.weight)
	torch.nn.init.uniform_(lin.bias, -0.5, 0.5)
	return lin.cuda().to(dtype)


@pytest.mark.parametrize("n,H", [(1, 64), (5, 128), (16, 4096), (17, 4096), (1000, 4096), (12_000, 4096), (4099, 512)])
def test_gather_route_is_exact_and_matches_linear(n, H):
	lin = _layer(H)
	states = torch.from_numpy(random_walk(n, 15, seed=n)).cuda()
	y = OhLinear(lin, route="gather")(states)
	assert y.shape == (n, H) and y.dtype == torch.float32
	# bit-exact against the same float32 adds in the same order
	wt = lin.weight.detach().t().contiguous()                      # (480, H)
	acc = lin.bias.detach().expand(n, H).clone()
	idx = states.long() + 24 * torch.arange(20, device="cuda")
	for i in range(20):
		acc = acc + wt[idx[:, i]]
	assert torch.equal(y, acc)
	# and against the reference formulation within float32 summation-order rounding
	ref = torch.nn.functional.linear(cube.device.as_oh(states), lin.weight, lin.bias)
	bound = 2e-6 * (lin.bias.detach().abs() + wt.abs()[idx].sum(dim=1))
	assert ((y - ref).abs() <= bound + 1e-7).all()


@pytest.mark.parametrize("n,H", [(1, 64), (31, 128), (128, 4096), (129, 4096), (1000, 4096), (12_000, 4096), (4099, 512)])
def test_mfma_route_matches_bf16_linear(n, H):
	lin = _layer(H, torch.bfloat16, seed=1)
	states = torch.from_numpy(random_walk(n, 15, seed=100 + n)).cuda()
	y = OhLinear(lin, route="mfma")(states)
	assert y.shape == (n, H) and y.dtype == torch.bfloat16
	ref = torch.nn.functional.linear(cube.device.as_oh(states), lin.weight.float(), lin.bias.float())       # f32 math on the bf16 weights
	assert torch.allclose(y.float(), ref, rtol=2.0 ** -7, atol=1e-6)
	# the one-hot is exact in bf16, so torch's own bf16 layer is the same computation up to its accumulation order
	tb = torch.nn.functional.linear(cube.device.as_oh(states).to(torch.bfloat16), lin.weight, lin.bias)
	assert torch.allclose(y.float(), tb.float(), rtol=2.0 ** -6, atol=1e-6)
	# gather route from the same bf16 weights, bf16 output: same numbers up to the final rounding
	g = OhLinear(lin, route="gather")(states)
	assert g.dtype == torch.bfloat16 and torch.allclose(g.float(), ref, rtol=2.0 ** -7, atol=1e-6)


def test_fused_net_in_astar_and_adi():
	"""The fused first layer behind the A* engine (states instead of one-hot rows) and ADI: same search / same targets."""
	from benchmarks.nets import FcSmall
	from librubiks_amd.adi import adi_traindata
	from librubiks_amd.solving.agents import AStar
	net = FcSmall(seed=0).cuda().eval()
	f = fuse_first_linear(net)
	states = torch.from_numpy(random_walk(300, 12, seed=9)).cuda()
	with torch.no_grad():
		want_p, want_v = net(cube.device.as_oh(states))
		got_p, got_v = f(states)
	assert torch.allclose(got_v, want_v, rtol=1e-4, atol=1e-4) and torch.allclose(got_p, want_p, rtol=1e-4, atol=1e-4)
	np.random.seed(21)
	start, _, _ = orc.scramble(9, True)
	plain, fused = AStar(net, 0.2, 50), AStar(net, 0.2, 50, fused_first_layer=True)
	a, b = plain.search(start, None, 20_000), fused.search(start, None, 20_000)
	# float32 values differ in the last bits between the two first layers, so the searches may diverge late; both must be
	# valid searches of the same budget, and the first iterations (root and its children) are identical
	assert len(fused) <= 20_000 and (fused.states[1:14] == plain.states[1:14]).all()
	for agent, ok in ((plain, a), (fused, b)):
		if ok:
			s = start
			for act in agent.action_queue:
				s = orc.rotate(s, act // 2, 1 - act % 2)
			assert orc.is_solved(s)
	np.random.seed(4)
	oh1, p1, v1, w1 = adi_traindata(net, 20, 8, 0.5, "lapanfix", ff_batches=2)
	np.random.seed(4)
	oh2, p2, v2, w2 = adi_traindata(net, 20, 8, 0.5, "lapanfix", ff_batches=2, fused_first_layer=True)
	assert torch.equal(oh1, oh2) and torch.allclose(v1, v2, rtol=1e-4, atol=1e-4) and (p1 == p2).float().mean() > 0.95


def test_fused_net_in_mcts():
	"""MCTSBatch with the first layer fused: the tree's P and V equal a fresh forward of the net (reference invariant,
	tests/test_agents.py:84-90, atol 1e-4 because the first layer sums in another order) and the graph replay works."""
	from benchmarks.nets import FcSmall
	from librubiks_amd.solving.agents import MCTSBatch
	net = FcSmall(seed=3).cuda().eval()
	starts = []
	for i in range(6):
		np.random.seed(50 + i)
		starts.append(orc.scramble(8, True)[0])
	starts = np.array(starts)
	agent = MCTSBatch(net, 1.0, 6, capacity=1500, fused_first_layer=True)
	agent.search(starts, max_states=1500, max_sims=100, use_graph=True, poll=32)
	for t in (0, 5):
		a = agent.tree_arrays(t)
		n = a["n"]
		assert n > 500 and (a["states"][1] == starts[t]).all()
		with torch.no_grad():
			p, v = net(cube.as_oh(a["states"][1:n + 1]))
		assert np.allclose(a["P"][1:n + 1], p.softmax(dim=1).cpu().numpy(), atol=1e-4)
		assert np.allclose(a["V"][1:n + 1], v.reshape(-1).cpu().numpy(), atol=1e-4)
		i, j = np.nonzero(a["neighbors"][1:n + 1])
		moved = orc.multi_rotate(a["states"][i + 1], j // 2, 1 - j % 2)
		assert (moved == a["states"][a["neighbors"][i + 1, j]]).all()
