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


@pytest.mark.parametrize("n,H", [(1, 64), (12, 4096), (31, 128), (33, 192), (256, 4096), (768, 4096), (769, 4096), (1000, 4096), (1536, 4096), (3072, 4096), (3073, 4096), (6000, 2048), (4099, 512)])
@pytest.mark.parametrize("act,affine", [(None, False), ("elu", True), ("relu", False)])
def test_mfma_forms_give_the_same_bits(n, H, act, affine):
	"""Round 5: the MFMA route has two forms -- a wave per 32 x 32 output tile with its weights straight from global memory (few rows:
	a search step's batch) and the LDS-resident weight tile (many rows).  Same instruction, same k order, same epilogue: every
	output equal bit for bit, whatever the batch (ragged row tiles, a last column group of two tiles at H = 192, both sides of the
	768-row switch; 3 072 rows take three tiles per wave and pass in the tiled form, 1 536 two), and `route="mfma"` is one of the two."""
	lin = _layer(H, dtype=torch.bfloat16, seed=n + H)
	states = torch.from_numpy(random_walk(n, 15, seed=n)).cuda()
	layer = OhLinear(lin, route="mfma")
	if act or affine:
		bn = None
		if affine:
			bn = torch.nn.BatchNorm1d(H).cuda().eval()
			with torch.no_grad():
				bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0); bn.weight.normal_(); bn.bias.normal_()
		layer.set_epilogue({"elu": torch.nn.ELU(), "relu": torch.nn.ReLU(), None: None}[act], bn)
	# guard rows behind the outputs: neither form writes past row n
	outs = {r: torch.full((n + 2, H), 7.0, dtype=torch.bfloat16, device="cuda") for r in ("mfma_direct", "mfma_tiled", "mfma")}
	for r, o in outs.items():
		layer(states, out=o[:n], route=r)
	torch.cuda.synchronize()
	for r, o in outs.items():
		assert (o[n:] == 7.0).all(), r
	assert torch.equal(outs["mfma_direct"][:n].view(torch.int16), outs["mfma_tiled"][:n].view(torch.int16))
	assert torch.equal(outs["mfma"][:n].view(torch.int16), outs["mfma_tiled"][:n].view(torch.int16))
	if not act and not affine:                     # and the layer itself, as test_mfma_route_matches_bf16_linear states it
		ref = torch.nn.functional.linear(cube.device.as_oh(states).to(torch.bfloat16), lin.weight, lin.bias).float()
		got = outs["mfma_direct"][:n].float()
		assert ((got - ref).abs() <= 2.0 ** -6 * ref.abs() + 2.0 ** -6).all()


def test_mfma_forms_agree_on_random_shapes():
	"""The tiled form's launcher chooses row groups, rows per group and two or three tiles per wave from (n, H); the direct form has one
	geometry.  Sixty seeded random shapes -- every output of one form against the other's, the buffers pre-filled with DIFFERENT values so
	that a row or column neither form wrote cannot compare equal -- plus the float32 statement of the layer on a sample of rows."""
	rng = np.random.RandomState(20240505)
	layers = {}
	for case in range(60):
		H = int(rng.choice([64, 128, 192, 320, 512, 1024, 2048, 4096]))
		n = int(rng.choice([rng.randint(1, 400), rng.randint(400, 4000), rng.randint(4000, 20_000)]))
		if H not in layers:
			layers[H] = _layer(H, torch.bfloat16, seed=H)
		lin = layers[H]
		states = torch.from_numpy(random_walk(n, 12, seed=case)).cuda()
		layer = OhLinear(lin, route="mfma")
		a = torch.full((n, H), 1.0, dtype=torch.bfloat16, device="cuda")
		b = torch.full((n, H), 2.0, dtype=torch.bfloat16, device="cuda")
		layer(states, out=a, route="mfma_direct")
		layer(states, out=b, route="mfma_tiled")
		assert torch.equal(a.view(torch.int16), b.view(torch.int16)), (case, n, H)
		pick = torch.from_numpy(rng.randint(0, n, size=min(n, 64))).cuda()
		ref = torch.nn.functional.linear(cube.device.as_oh(states[pick]), lin.weight.float(), lin.bias.float())
		assert torch.allclose(a[pick].float(), ref, rtol=2.0 ** -7, atol=1e-6), (case, n, H)


@pytest.mark.parametrize("K", [512, 1024, 2048])
@pytest.mark.parametrize("M", [1, 12, 13, 16])
@pytest.mark.parametrize("act", ["elu", "elu0.7", "relu", None])
def test_tail_linear_is_activation_plus_linear(K, M, act):
	"""rk_tail_linear (round 5): a head's last Linear(K, M <= 16) with the activation in front of it in one launch (model.py:124-129).
	Against float64 on the same bf16 numbers: the activation in float32 rounded to bf16 (what torch's activation kernel stores), then
	exact products and sums -- the kernel may differ by its float32 accumulation (1e-5 of the sum of magnitudes) and the final
	rounding to bf16 (2^-8 relative).  And against the two torch modules it replaces (bf16 GEMM: another summation order).  Ragged
	row counts, a strided input view, rows behind the output untouched."""
	from librubiks_amd.oh_linear import TailLinear
	torch.manual_seed(K + M)
	lin = torch.nn.Linear(K, M).cuda().to(torch.bfloat16)
	module = {"elu": torch.nn.ELU(), "elu0.7": torch.nn.ELU(alpha=0.7), "relu": torch.nn.ReLU(), None: None}[act]
	tail = TailLinear(module, lin)
	for n in (1, 15, 16, 17, 3072, 4099):
		wide = (torch.randn(n, K + 8, device="cuda") * 1.5).to(torch.bfloat16)
		for x in (wide[:, :K].contiguous(), wide[:, :K]):                      # row stride K, row stride K + 8
			y = tail(x)
			assert y.shape == (n, M) and y.dtype == torch.bfloat16
			a = (module(x.float()) if module is not None else x.float()).to(torch.bfloat16).double()
			W, bias = lin.weight.detach().double(), lin.bias.detach().double()
			ref = a @ W.t() + bias
			mag = a.abs() @ W.abs().t() + bias.abs()
			assert ((y.double() - ref).abs() <= 2.0 ** -8 * ref.abs() + 1e-5 * mag + 1e-30).all(), (n, x.stride(0))
			with torch.no_grad():
				tb = lin(module(x) if module is not None else x)
			assert torch.allclose(y.float(), tb.float(), rtol=2.0 ** -6, atol=2.0 ** -6 * float(mag.max()) * 0.05 + 1e-3)
	# the kernel writes n rows of M values and nothing else
	out_rows = 37
	x = torch.randn(out_rows, K, device="cuda").to(torch.bfloat16)
	y = tail(x)
	assert torch.isfinite(y.float()).all()
	with pytest.raises(ValueError):
		tail(x.float())
	with pytest.raises(ValueError):
		tail(x[:, : K - 8])
	with pytest.raises(ValueError):
		TailLinear(torch.nn.ELU(), torch.nn.Linear(480, 12).cuda().to(torch.bfloat16))


def test_folded_net_ends_in_tail_linear_and_equals_the_torch_tail():
	"""`fuse_first_linear(net, epilogue=True, fold_batchnorm=True)` on a bfloat16 net of the reference's shape: every stack of heads ends
	in a TailLinear (policy 12, value 1, both 13 columns), and its outputs equal the same folded net with the torch tail
	(`oh_linear.FUSE_TAIL = False`) within bf16 rounding -- on a search step's batch and on one row."""
	from benchmarks.nets import FcSmall
	from librubiks_amd import oh_linear
	net = _randomise_batchnorm(FcSmall(seed=5)).cuda().eval().to(torch.bfloat16)
	fused = fuse_first_linear(net, epilogue=True, fold_batchnorm=True)
	assert isinstance(fused.merged_heads[-1], oh_linear.TailLinear) and fused.merged_heads[-1].out_features == 13
	assert isinstance(fused.modules[1][-1], oh_linear.TailLinear) and isinstance(fused.modules[2][-1], oh_linear.TailLinear)
	oh_linear.FUSE_TAIL = False
	try:
		plain = fuse_first_linear(net, epilogue=True, fold_batchnorm=True)
	finally:
		oh_linear.FUSE_TAIL = True
	assert isinstance(plain.merged_heads[-1], torch.nn.Linear)
	for n in (1, 3072):
		states = torch.from_numpy(random_walk(n, 14, seed=n)).cuda()
		with torch.no_grad():
			for kw in ({}, {"policy": False}, {"value": False}):
				got, want = fused(states, **kw), plain(states, **kw)
				got, want = (got if isinstance(got, list) else [got]), (want if isinstance(want, list) else [want])
				for g, w in zip(got, want):
					assert g.shape == w.shape and g.dtype == w.dtype
					assert torch.allclose(g.float(), w.float(), rtol=2.0 ** -5, atol=2.0 ** -5 * float(w.float().abs().max()))


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


def _randomise_batchnorm(net, seed=0):
	"""fresh BatchNorm layers are (almost) the identity in eval mode: give them statistics and affine parameters worth folding"""
	g = torch.Generator().manual_seed(seed)
	for m in net.modules():
		if isinstance(m, torch.nn.BatchNorm1d):
			with torch.no_grad():
				m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.3)
				m.running_var.copy_(torch.rand(m.num_features, generator=g) * 1.5 + 0.25)
				m.weight.copy_(torch.rand(m.num_features, generator=g) + 0.5)
				m.bias.copy_(torch.randn(m.num_features, generator=g) * 0.2)
	return net


@pytest.mark.parametrize("route", ["gather", "mfma"])
@pytest.mark.parametrize("act", ["elu", "elu0.7", "relu", None])
@pytest.mark.parametrize("affine", [False, True])
def test_epilogue_activation_and_batchnorm(route, act, affine):
	"""y = BatchNorm_eval(act(x W^T + b)) out of the layer's kernel (model.py:157-159) against the torch modules in float32.
	Tolerances: float32 route 1e-5 relative (exp and the affine map round differently from torch's kernels), bf16 route 2^-7."""
	H, n = 512, 777
	dtype = torch.float32 if route == "gather" else torch.bfloat16
	lin = _layer(H, dtype, seed=5)
	states = torch.from_numpy(random_walk(n, 15, seed=77)).cuda()
	module = {"elu": torch.nn.ELU(), "elu0.7": torch.nn.ELU(alpha=0.7), "relu": torch.nn.ReLU(), None: None}[act]
	bn = None
	if affine:
		bn = _randomise_batchnorm(torch.nn.Sequential(torch.nn.BatchNorm1d(H)), seed=3)[0].cuda().eval()
	layer = OhLinear(lin, route=route).set_epilogue(module, bn)
	y = layer(states)
	with torch.no_grad():
		ref = torch.nn.functional.linear(cube.device.as_oh(states), lin.weight.float(), lin.bias.float())
		if module is not None:
			ref = module(ref)
		if bn is not None:
			ref = bn(ref)
	if route == "gather":
		assert y.dtype == torch.float32 and torch.allclose(y, ref, rtol=1e-5, atol=2e-6)
	else:
		assert y.dtype == torch.bfloat16 and torch.allclose(y.float(), ref, rtol=2.0 ** -7, atol=2e-3)
	# and the epilogue can be taken off again
	plain = layer.set_epilogue(None, None)(states)
	assert torch.equal(plain, OhLinear(lin, route=route)(states))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", ["epilogue", "folded"])
def test_fused_net_with_epilogue_and_folded_batchnorm(dtype, mode):
	"""fc_small with the first layer's ELU + BatchNorm in the kernel ("epilogue") and the other BatchNorm layers folded into
	the following Linear layers ("folded"): the same function as the net on one-hot rows up to float rounding."""
	from benchmarks.nets import FcSmall
	from librubiks_amd.oh_linear import fused_net
	from librubiks_amd.solving.agents import AStar
	net = _randomise_batchnorm(FcSmall(seed=1), seed=8).cuda().eval()
	states = torch.from_numpy(random_walk(500, 14, seed=19)).cuda()
	with torch.no_grad():
		want_p, want_v = net(cube.device.as_oh(states))                 # float32 truth
		use = net if dtype == torch.float32 else _randomise_batchnorm(FcSmall(seed=1), seed=8).cuda().eval().to(dtype)
		f = fused_net(use, mode)
		got_p, got_v = f(states)
	if mode == "folded":
		assert not any(isinstance(m, torch.nn.BatchNorm1d) for seq in f.modules for m in seq)
	tol = dict(rtol=2e-4, atol=2e-4) if dtype == torch.float32 else dict(rtol=0.06, atol=0.06 * float(want_v.abs().max()))
	assert torch.allclose(got_v.float(), want_v, **tol) and torch.allclose(got_p.float(), want_p, **tol)
	if dtype == torch.float32:
		np.random.seed(33)
		start, _, _ = orc.scramble(8, True)
		agent = AStar(net, 0.2, 40, fused_first_layer=mode)
		ok = agent.search(start, None, 15_000)
		assert len(agent) <= 15_000 and (agent.states[1] == start).all()
		if ok:
			s = start
			for a in agent.action_queue:
				s = orc.rotate(s, a // 2, 1 - a % 2)
			assert orc.is_solved(s)


def test_bf16_net_outputs_go_to_the_engines_as_they_are():
	"""A bfloat16 net's values (A*) and raw logits + values (MCTS: softmax inside the backup kernel) are taken without
	conversion kernels.  bf16 -> float32 is exact, so A* must build the very same search as with the values converted
	by torch first; the MCTS tree's P and V must equal softmax / value of a fresh forward (bf16 rounding: atol 1e-2)."""
	from benchmarks.nets import FcSmall
	from librubiks_amd.solving.agents import AStar, MCTSBatch
	net = FcSmall(seed=6).cuda().eval().to(torch.bfloat16)

	class AsFloat32:                                   # the same net, values handed over as float32 (the old path)
		def eval(self): return self
		def parameters(self): return net.parameters()
		def __call__(self, x, policy=True, value=True):
			out = net(x, policy=policy, value=value)
			return [o.float() for o in out] if isinstance(out, list) else out.float()

	np.random.seed(12)
	start, _, _ = orc.scramble(9, True)
	a, b = AStar(net, 0.2, 60), AStar(AsFloat32(), 0.2, 60)
	ra, rb = a.search(start, None, 25_000), b.search(start, None, 25_000)
	n = len(a)
	assert ra == rb and n == len(b) and n > 5_000
	assert (a.states[1:n + 1] == b.states[1:n + 1]).all() and (a.G[1:n + 1] == b.G[1:n + 1]).all() and (a.parents[1:n + 1] == b.parents[1:n + 1]).all()

	starts = []
	for i in range(4):
		np.random.seed(80 + i)
		starts.append(orc.scramble(7, True)[0])
	for fused in (False, "folded"):
		agent = MCTSBatch(net, 1.0, 4, capacity=1200, fused_first_layer=fused)
		agent.search(np.array(starts), max_states=1200, max_sims=80, use_graph=True, poll=16)
		t = agent.tree_arrays(2)
		m = t["n"]
		assert m > 400
		with torch.no_grad():
			p, v = net(cube.as_oh(t["states"][1:m + 1]).to(torch.bfloat16))
		assert np.allclose(t["P"][1:m + 1], p.float().softmax(dim=1).cpu().numpy(), atol=1e-2)
		assert np.allclose(t["V"][1:m + 1], v.float().reshape(-1).cpu().numpy(), atol=2e-2)
		assert np.allclose(t["P"][1:m + 1].sum(axis=1), 1.0, atol=1e-6)


@pytest.mark.parametrize("mode", [True, "folded"])
def test_fused_copy_follows_the_net(mode):
	"""the fused first layer COPIES weights (and, folded, BatchNorm statistics and heads) while the reference trains
	its net in place and reassigns agent.net (train.py:134, :214).  The agent re-copies at the start of a search whenever the
	net changed -- in place (optimizer step) or by assignment -- and only then."""
	from benchmarks.nets import FcSmall
	from librubiks_amd.solving.agents import AStar, MCTSBatch
	net = FcSmall(seed=4).cuda().eval()
	np.random.seed(31)
	start, _, _ = orc.scramble(9, True)

	def same_search(x, y):
		n = len(x)
		return n == len(y) and (x.states[1:n + 1] == y.states[1:n + 1]).all() and (x.G[1:n + 1] == y.G[1:n + 1]).all() \
		       and (x.parents[1:n + 1] == y.parents[1:n + 1]).all()

	a = AStar(net, 0.2, 50, fused_first_layer=mode)
	a.search(start, None, 8_000)
	first = a._fs
	a.search(start, None, 8_000)
	assert a._fs is first                                          # nothing changed: no new copy
	before = (a.states.copy(), a.G.copy())
	with torch.no_grad():                                          # an optimizer step: parameters AND BatchNorm statistics move in place
		for prm in net.parameters():
			prm.mul_(0.5).add_(0.01)
		for m in net.modules():
			if isinstance(m, torch.nn.BatchNorm1d):
				m.running_mean.add_(0.05)
	a.search(start, None, 8_000)
	assert a._fs is not first
	fresh = AStar(net, 0.2, 50, fused_first_layer=mode)
	fresh.search(start, None, 8_000)
	assert same_search(a, fresh)
	assert len(a) != len(before[0]) - 1 or not (a.states == before[0]).all() or not (a.G == before[1]).all()      # the new weights do search differently
	# an update through `.data` does not bump the tensors' version counters; the checksum of the first layer catches it
	again = a._fs
	first_layer = next(net.parameters())
	first_layer.data.mul_(1.5)
	a.search(start, None, 8_000)
	assert a._fs is not again
	fresh = AStar(net, 0.2, 50, fused_first_layer=mode)
	fresh.search(start, None, 8_000)
	assert same_search(a, fresh)
	net2 = FcSmall(seed=9).cuda().eval()                           # the net is swapped (train.py:214: agent.net = net)
	a.net = net2
	a.search(start, None, 8_000)
	other = AStar(net2, 0.2, 50, fused_first_layer=mode)
	other.search(start, None, 8_000)
	assert same_search(a, other)
	# MCTS drives its batch engine with `self._batch.net = self.net`: the batch agent follows too
	mb = MCTSBatch(net, 1.0, 2, capacity=600, fused_first_layer=mode)
	starts = np.array([start, start])
	mb.search(starts, max_states=600, max_sims=30)
	copy1 = mb._fs
	mb.net = net
	mb.search(starts, max_states=600, max_sims=30)
	assert mb._fs is copy1                                         # re-assigning the SAME unchanged module costs nothing
	mb.net = net2
	mb.search(starts, max_states=600, max_sims=30)
	assert mb._fs is not copy1


def test_nan_stays_nan_through_the_bf16_outputs():
	"""f32 -> bf16 by integer arithmetic on the bits turns some NaNs into +0 or +inf (0xFFFFFFFF -> +0,
	0x7F800001 -> +inf; MI355X_MICROARCH.md, correctness boundaries).  Both places that round to bf16 -- the weight copy of
	the MFMA route and the GATHER route's bf16 output -- use the hardware convert now: a NaN in, a NaN out."""
	lin = torch.nn.Linear(480, 128).cuda()
	nan_bits = torch.tensor([-1, 0x7F800001, 0x7FC00000, -4194304], dtype=torch.int32, device="cuda").view(torch.float32)   # 0xFFFFFFFF, sNaN, qNaN, 0xFFC00000
	assert torch.isnan(nan_bits).all()
	with torch.no_grad():
		lin.bias[:4] = nan_bits                                # columns 0..3 of every output row: NaN + finite = NaN
		lin.weight[8, 0] = nan_bits[0]                         # one-hot column 0 = "cubie 0 has code 0"
		lin.weight[9, 24] = nan_bits[1]                        # one-hot column 24 = "cubie 1 has code 0"
	states = torch.from_numpy(random_walk(512, 15, seed=5)).cuda()
	picked8, picked9 = states[:, 0] == 0, states[:, 1] == 0
	assert picked8.any() and not picked8.all() and picked9.any() and not picked9.all()
	first = OhLinear(lin)
	outs = {}
	for route in ("gather", "mfma"):
		outs[route] = torch.empty((512, 128), dtype=torch.bfloat16, device="cuda")
		first(states, out=outs[route], route=route)
		nan = torch.isnan(outs[route])
		assert nan[:, :4].all() and not nan[:, 16:].any() and not torch.isinf(outs[route]).any(), route
	# a NaN WEIGHT: the gather-sum meets it only in the rows that select it; the matrix product multiplies it with the
	# one-hot's zeros in every row (0 * NaN = NaN), exactly as torch's dense Linear on the one-hot rows does
	g, m = torch.isnan(outs["gather"]), torch.isnan(outs["mfma"])
	assert torch.equal(g[:, 8], picked8) and torch.equal(g[:, 9], picked9)
	dense = torch.isnan(torch.nn.functional.linear(cube.device.as_oh(states), lin.weight, lin.bias))
	assert dense[:, 8].all() and dense[:, 9].all() and m[:, 8].all() and m[:, 9].all()
