"""
GPU parity of the batched MCTS engine (rk_mcts_*, librubiks_amd.solving.agents.MCTS / MCTSBatch):
  * replays the traces of the UNMODIFIED reference MCTS (tests/golden/mcts_trace.npz, exact-integer stub net);
  * a batch of trees advanced in lock-step equals the CPU oracle run on every start state alone, eager and as a
    replayed hipGraph;
  * with a float net: the structural invariants of the reference's tests/test_agents.py:49-94.
"""
import numpy as np
import pytest
import torch

from librubiks_amd import cube
from librubiks_amd.solving.agents import MCTS, MCTSBatch
from oracle import cube_oracle as orc
from oracle.search_oracle import MCTSOracle, PolicyStubNet, StubNet

pytestmark = pytest.mark.gpu


def _same_tree(arrs: dict, ref: MCTSOracle):
	n = len(ref)
	assert arrs["n"] == n
	assert (arrs["states"][1:n + 1] == ref.states[1:n + 1]).all()
	assert (arrs["neighbors"][1:n + 1] == ref.neighbors[1:n + 1]).all()
	assert (arrs["leaves"][1:n + 1] == ref.leaves[1:n + 1]).all()
	assert (arrs["N"][1:n + 1] == ref.N[1:n + 1]).all()
	assert (arrs["W"][1:n + 1] == ref.W[1:n + 1]).all()
	assert (arrs["L"][1:n + 1] == ref.L[1:n + 1]).all()
	assert (arrs["V"][1:n + 1] == ref.V[1:n + 1]).all()
	assert (arrs["P"][1:n + 1] == ref.P[1:n + 1]).all()


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e", "f", "g", "h"])
def test_reference_traces(golden, tag):
	t = golden["mcts_trace"]
	_, _, search_graph, max_states = (int(x) for x in t[f"{tag}_params"])
	start = t[f"{tag}_start"]
	net = PolicyStubNet() if f"{tag}_P" in t else StubNet()          # trace f: non-uniform priors (exactly 0, 1/8, 1/4)
	agent = MCTS(net, float(t[f"{tag}_c"]), bool(search_graph), use_hipgraph=tag in ("b", "d", "h"))
	solved = agent.search(start, time_limit=None, max_states=max_states)
	n = int(t[f"{tag}_n"])
	assert solved == bool(t[f"{tag}_solved"]) and len(agent) == n
	assert int(agent._batch.status[0, 3]) == int(t[f"{tag}_sims"])
	assert (agent.states[1:n + 1] == t[f"{tag}_states"]).all()
	assert (agent.neighbors[1:n + 1] == t[f"{tag}_neighbors"]).all() and agent.neighbors.dtype == np.int64
	assert (agent.leaves[1:n + 1] == t[f"{tag}_leaves"]).all()
	assert (agent.N[1:n + 1] == t[f"{tag}_N"]).all()
	assert (agent.W[1:n + 1] == t[f"{tag}_W"]).all() and agent.W.dtype == np.float64
	assert (agent.L[1:n + 1] == t[f"{tag}_L"]).all()
	assert (agent.V[1:n + 1] == t[f"{tag}_V"]).all()
	if f"{tag}_P" in t:
		assert (agent.P[1:n + 1] == t[f"{tag}_P"]).all()
	assert list(agent.action_queue) == t[f"{tag}_action_queue"].tolist()
	if solved:
		s = start
		for a in agent.action_queue:
			s = cube.rotate(s, *cube.action_space[a])
		assert cube.is_solved(s)


@pytest.mark.parametrize("use_graph", [False, True])
def test_batch_equals_oracle_per_tree(use_graph):
	"""12 trees with different depths and budgets in one engine; each must equal the oracle run alone."""
	T, c = 12, 5.0
	starts, budgets = [], []
	for i in range(T):
		np.random.seed(200 + i)
		s, _, _ = orc.scramble(2 + i % 5, True)
		starts.append(s)
		budgets.append(600 + 250 * i)
	starts = np.array(starts)
	starts[7] = orc.SOLVED                       # a start that is already solved (agents.py:468)
	agent = MCTSBatch(StubNet(), c, T, capacity=4000)
	solved = agent.search(starts, max_states=np.array(budgets), use_graph=use_graph, poll=32)
	n_solved = 0
	for i in range(T):
		ref = MCTSOracle(StubNet(), c, False)
		ref_solved = ref.search(starts[i], budgets[i])
		assert bool(solved[i]) == ref_solved, i
		assert list(agent.action_queue_of(i)) == list(ref.action_queue), i
		if i != 7:
			_same_tree(agent.tree_arrays(i), ref)
			assert int(agent.status[i, 3]) == ref.sims
		n_solved += ref_solved
	assert 0 < n_solved <= T and solved[7]


class TinyNet(torch.nn.Module):
	def __init__(self):
		super().__init__()
		torch.manual_seed(1)
		self.body = torch.nn.Sequential(torch.nn.Linear(480, 128), torch.nn.ELU(), torch.nn.Linear(128, 64), torch.nn.ELU())
		self.p, self.v = torch.nn.Linear(64, 12), torch.nn.Linear(64, 1)

	def forward(self, x, policy=True, value=True):
		h = self.body(x)
		out = ([self.p(h)] if policy else []) + ([self.v(h)] if value else [])
		return out if len(out) > 1 else out[0]


def test_real_net_invariants():
	"""tests/test_agents.py:49-94 of the reference with a float net on the GPU."""
	net = TinyNet().cuda().eval()
	np.random.seed(4)
	for depth, sg in ((50, False), (3, False), (3, True)):
		state, _, _ = cube.scramble(depth)
		agent = MCTS(net, c=1, search_graph=sg)
		solved = agent.search(state, time_limit=None, max_states=3000)
		idx = agent.indices
		n = len(agent)
		assert idx[state.tobytes()] == 1 and sorted(idx.values()) == list(range(1, n + 1))
		assert (agent.states[1] == state).all()
		used = np.arange(1, n + 1)
		if not sg:
			nb = agent.neighbors
			for i in np.random.randint(1, n + 1, 200):
				for j in range(12):
					if nb[i, j]:
						assert (agent.states[nb[i, j]] == orc.rotate(agent.states[i], j // 2, 1 - j % 2)).all()
			assert (agent.neighbors[used].all(axis=1) != agent.leaves[used]).all()
		with torch.no_grad():
			p, v = net(cube.as_oh(agent.states[used]))
		p, v = p.softmax(dim=1).cpu().numpy(), v.squeeze().cpu().numpy()
		assert np.isclose(agent.P[used], p, atol=1e-5).all()
		assert np.isclose(agent.V[used], v, atol=1e-5).all()
		assert agent.W[used].all()
		s = state
		for a in agent.action_queue:
			s = cube.rotate(s, *cube.action_space[a])
		assert cube.is_solved(s) == solved


class ExactLogitNet:
	"""Logits and values that are EXACT in float32 whatever the batch shape or summation order (one-hot rows times a
	table of multiples of 1/8, twenty terms each), so a fresh forward reproduces bit for bit what the engine was fed;
	the logits spread over about +-6, i.e. real (non-uniform, non-dyadic) softmax work."""
	def __init__(self, dtype=torch.float32, seed=3):
		g = torch.Generator().manual_seed(seed)
		self.wp = (torch.randint(-12, 13, (480, 12), generator=g).float() / 8).cuda()
		self.wv = (torch.randint(-8, 9, (480, 1), generator=g).float() / 8).cuda()
		self.dtype = dtype

	def eval(self):
		return self

	def __call__(self, x, policy=True, value=True):
		x = x.float()
		out = ([(x @ self.wp).to(self.dtype)] if policy else []) + ([(x @ self.wv).to(self.dtype)] if value else [])
		return out if len(out) > 1 else out[0]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_priors_against_torch_softmax(dtype):
	"""The priors are softmax(logits) (agents.py:472, :551-552).  On real (non-dyadic) logits, three ways to get them:
	  * priors="torch"   -> the stored priors ARE torch.softmax of the same logits on the device, bit for bit;
	  * priors="kernel"  -> the backup kernel's own softmax (exp(x - max) / sum in float32 on the raw logits, the sum taken in
	                        the order torch's kernel adds): the SAME bits as torch.softmax on this device --
	                        asserted here on every stored row, and on a million random rows in the next test;
	  * priors="reference" -> the reference's own computation: root by the device's softmax, every other node by the HOST's
	                        (`p.cpu().softmax(dim=1)`, agents.py:551-552), bit for bit."""
	net = ExactLogitNet(dtype)
	starts = []
	for i in range(3):
		np.random.seed(500 + i)
		starts.append(orc.scramble(6 + i, True)[0])
	starts = np.array(starts)
	for priors in ("torch", "kernel", "reference"):
		agent = MCTSBatch(net, 1.0, 3, capacity=1500, priors=priors)
		agent.search(starts, max_states=1500, max_sims=100)
		for tree in range(3):
			t = agent.tree_arrays(tree)
			n = t["n"]
			assert n > 300
			logits, v = net(cube.as_oh(t["states"][1:n + 1]))
			got = t["P"][1:n + 1]
			assert (t["V"][1:n + 1] == v.float().double().reshape(-1).cpu().numpy()).all()
			on_device = logits.float().softmax(dim=1).double().cpu().numpy()
			if priors in ("torch", "kernel"):
				assert (got == on_device).all(), (priors, np.abs(got - on_device).max())
			else:
				on_host = logits.cpu().softmax(dim=1).double().numpy()
				assert (got[0] == on_device[0]).all()                            # the root: agents.py:472
				assert (got[1:] == on_host[1:]).all()                            # everybody else: agents.py:551-552
			# (a bfloat16 net's logits softmaxed on the host stay bfloat16 there, as `p.cpu().softmax(dim=1)` leaves them: rows sum to 1 within bf16)
			assert np.abs(got.sum(axis=1) - 1).max() < (1e-6 if priors != "reference" or dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_kernel_softmax_equals_torch_bitwise_on_random_logits(dtype):
	"""The in-kernel softmax against torch.softmax on this device, bit for bit, on 12 x 87 000 random rows -- wide and narrow
	ranges, ties, a few -inf -- driven through the engine's own entry (rk_mcts_backup_select_logits) on one-simulation trees."""
	import ctypes as C
	from librubiks_amd import _ffi
	lib, st = _ffi.lib(), _ffi.stream_ptr
	T = 87_000
	h = C.c_void_p()
	_ffi.check(lib.rk_mcts_create(C.byref(h), T, 13, 4))
	np.random.seed(1)
	start = orc.scramble(9, True)[0]
	starts = np.ascontiguousarray(np.broadcast_to(start, (T, 20)))
	_ffi.check(lib.rk_mcts_reset(h, starts.ctypes.data, None, 1.0, 100.0, st()))
	g = torch.Generator(device="cuda").manual_seed(5)
	scale = torch.tensor([0.01, 0.5, 3.0, 20.0], device="cuda")[torch.randint(0, 4, (12 * T, 1), device="cuda", generator=g)]
	logits = (torch.randn((12 * T, 12), device="cuda", generator=g) * scale)
	logits[::97, 3] = logits[::97, 7]                                         # ties
	logits[::1013, 5] = -float("inf")
	logits = logits.to(dtype)
	values = torch.randn(12 * T, device="cuda", generator=g).to(dtype)
	root_p = torch.full((T, 12), 1 / 12, device="cuda")
	root_v = torch.zeros(T, device="cuda")
	_ffi.check(lib.rk_mcts_set_root_pv(h, root_p.data_ptr(), root_v.data_ptr(), st()))
	_ffi.check(lib.rk_mcts_expand(h, st()))
	_ffi.check(lib.rk_mcts_backup_select_logits(h, logits.data_ptr(), 12, values.data_ptr(), 1, _ffi.OH_F32 if dtype == torch.float32 else _ffi.OH_BF16, st()))
	want = logits.float().softmax(dim=1).double().cpu().numpy().reshape(T, 12, 12)
	P = np.zeros((12, 12))
	nb = np.zeros((1, 12), np.int64)
	bad = 0
	for t in range(0, T, 613):                                               # a sample of trees: all 12 new children each
		_ffi.check(lib.rk_mcts_export(h, t, 1, 1, None, nb.ctypes.data, None, None, None, None, None, None, st()))
		assert (nb[0] == np.arange(2, 14)).all()
		_ffi.check(lib.rk_mcts_export(h, t, 2, 12, None, None, None, P.ctypes.data, None, None, None, None, st()))
		bad += int((P != want[t]).sum())
	assert bad == 0
	_ffi.check(lib.rk_mcts_destroy(h))


def test_default_priors():
	"""`MCTS(net, c, search_graph)` with default arguments computes P as agents.py:472 / :551-552 do."""
	assert MCTS(StubNet(), 1.0, False).priors == "reference" and MCTS(StubNet(), 1.0, False, use_hipgraph=True).priors == "kernel"
	assert MCTSBatch(StubNet(), 1.0, 2).priors == "kernel" and MCTSBatch(StubNet(), 1.0, 2, torch_softmax=True).priors == "torch"
	net = ExactLogitNet()
	np.random.seed(77)
	start = orc.scramble(8, True)[0]
	agent = MCTS(net, 1.0, False)
	agent.search(start, None, 2000)
	n = len(agent)
	logits, v = net(cube.as_oh(agent.states[1:n + 1]))
	assert (agent.P[1] == logits[:1].softmax(dim=1).double().cpu().numpy()[0]).all()
	assert (agent.P[2:n + 1] == logits[1:].cpu().softmax(dim=1).double().numpy()).all()


@pytest.mark.parametrize("use_graph", [False, True])
def test_search_graph_on_the_device(use_graph):
	"""agents.py:483-486 + :597-633 for a BATCH of trees: graph completion and the breadth-first shortening run as
	kernels (rk_mcts_search_graph); every solved tree's `neighbors` and action queue equal the oracle's MCTS(search_graph=True)
	run alone -- the same path, not merely one of the same length --, unsolved trees are left as they are."""
	# a strongly non-uniform (exact) policy and a large c drive the descents round in circles: the solution is then found at the
	# end of a path that the graph shortcuts (seeds picked with the oracle: 1002 5 -> 3 moves, 1010 9 -> 5, 1032 9 -> 3, ...)
	seeds, c = [1002, 1010, 1004, 1014, 1000, 1032, 1023, 1001, 1003, 1005], 50.0
	T = len(seeds)
	starts, budgets = [], []
	for seed in seeds:
		np.random.seed(seed)
		starts.append(orc.scramble(3 + seed % 4, True)[0])
		budgets.append(4000)
	starts = np.array(starts)
	budgets[4] = 40                                                         # one tree that cannot solve
	batch = MCTSBatch(PolicyStubNet(), c, T, capacity=4000, search_graph=True)
	solved = batch.search(starts, max_states=np.array(budgets), use_graph=use_graph, poll=16)
	shorter = 0
	for i in range(T):
		ref = MCTSOracle(PolicyStubNet(), c, True)
		assert ref.search(starts[i], budgets[i]) == bool(solved[i]), i
		plain = MCTSOracle(PolicyStubNet(), c, False)
		plain.search(starts[i], budgets[i])
		a = batch.tree_arrays(i)
		n = len(ref)
		assert a["n"] == n and (a["neighbors"][1:n + 1] == ref.neighbors[1:n + 1]).all(), i
		assert list(batch.action_queue_of(i)) == list(ref.action_queue), i
		shorter += len(ref.action_queue) < len(plain.action_queue)
		if solved[i]:
			s = starts[i]
			for act in batch.action_queue_of(i):
				s = orc.rotate(s, act // 2, 1 - act % 2)
			assert orc.is_solved(s)
	assert solved.sum() >= 6 and not solved[4] and shorter >= 5             # the shortening did shorten something
	# the single-tree agent goes through the same kernels
	one = MCTS(PolicyStubNet(), c, True)
	for i in (1, 5, 8):
		ref = MCTSOracle(PolicyStubNet(), c, True)
		assert one.search(starts[i], None, budgets[i]) == ref.search(starts[i], budgets[i])
		assert list(one.action_queue) == list(ref.action_queue) and (one.neighbors[1:len(ref) + 1] == ref.neighbors[1:len(ref) + 1]).all()


class _CaptureNet:
	"""StubNet's numbers without host-side allocations inside a capture."""
	def __init__(self):
		self.sol = torch.from_numpy(orc.as_oh(orc.SOLVED)[0]).cuda()

	def __call__(self, x):
		return torch.zeros(len(x), 12, device="cuda"), -(20 - (x * self.sol).sum(dim=1))


@pytest.mark.parametrize("captured_at", [0, 3])
def test_a_step_captured_at_any_point_replays_correctly(captured_at):
	"""a C-ABI caller captures ONE step -- expand, children one-hot, net, backup + select with expand-ahead
	on -- into a hipGraph, straight after the reset or in the middle of a search, and replays it.  Whether the path's leaf still
	needs expanding is decided on the device at every replay, so both graphs give the oracle's trees; round 3 took the
	decision on the host at capture time and a graph captured at step 0 re-expanded leaves that were expanded ahead."""
	import ctypes as C
	from librubiks_amd import _ffi
	lib, st = _ffi.lib(), _ffi.stream_ptr
	T, sims, cap = 5, 60, 2000
	starts = []
	for i in range(T):
		np.random.seed(900 + i)
		starts.append(orc.scramble(7 + i, True)[0])
	starts = np.array(starts)
	h = C.c_void_p()
	_ffi.check(lib.rk_mcts_create(C.byref(h), T, cap, 512))
	_ffi.check(lib.rk_mcts_reset(h, starts.ctypes.data, None, 2.0, 100.0, st()))
	net = _CaptureNet()
	root_oh = torch.empty((T, 480), device="cuda")
	_ffi.check(lib.rk_mcts_roots_oh(h, root_oh.data_ptr(), _ffi.OH_F32, st()))
	p, v = net(root_oh)
	p = p.softmax(dim=1).contiguous()
	_ffi.check(lib.rk_mcts_set_root_pv(h, p.data_ptr(), v.contiguous().data_ptr(), st()))
	_ffi.check(lib.rk_mcts_set_expand_ahead(h, -1))
	oh = torch.empty((12 * T, 480), device="cuda")
	keep = []

	def step():
		_ffi.check(lib.rk_mcts_expand(h, st()))
		_ffi.check(lib.rk_mcts_children_oh(h, oh.data_ptr(), _ffi.OH_F32, st()))
		logits, values = net(oh)
		keep.append((logits, values))
		_ffi.check(lib.rk_mcts_backup_select_logits(h, logits.data_ptr(), 12, values.data_ptr(), 1, _ffi.OH_F32, st()))

	side = torch.cuda.Stream()
	with torch.cuda.stream(side):
		step() if captured_at else net(oh)                                  # (the allocator is warm either way)
		for _ in range(max(0, captured_at - 1)):
			step()
		torch.cuda.synchronize()
		graph = torch.cuda.CUDAGraph()
		with torch.cuda.graph(graph, stream=side):
			step()
	for _ in range(sims - captured_at):
		graph.replay()
	_ffi.check(lib.rk_mcts_set_expand_ahead(h, 0))
	step()                                                                  # completes the leaves expanded ahead
	status = np.zeros((T, 6), np.int64)
	_ffi.check(lib.rk_mcts_status(h, status.ctypes.data, st()))
	assert not status[:, 5].any(), status
	for t in range(T):
		ref = MCTSOracle(StubNet(), 2.0, False)
		ref_solved = ref.search(starts[t], cap, max_sims=sims + 1)
		n = len(ref)
		assert status[t, 2] == n and bool(status[t, 1]) == ref_solved and status[t, 3] == ref.sims, (t, status[t], n, ref.sims)
		nb, N = np.zeros((n, 12), np.int64), np.zeros((n, 12), np.int64)
		W = np.zeros((n, 12))
		_ffi.check(lib.rk_mcts_export(h, t, 1, n, None, nb.ctypes.data, None, None, None, N.ctypes.data, W.ctypes.data, None, st()))
		assert (nb == ref.neighbors[1:n + 1]).all() and (N == ref.N[1:n + 1]).all() and (W == ref.W[1:n + 1]).all(), t
	_ffi.check(lib.rk_mcts_destroy(h))


def test_a_backup_without_an_expansion_is_refused():
	"""A step that leaves rk_mcts_expand out is only right while every backup expands ahead; used on a fresh engine it must stop
	the trees with error 3 instead of backing up stale children."""
	import ctypes as C
	from librubiks_amd import _ffi
	lib, st = _ffi.lib(), _ffi.stream_ptr
	h = C.c_void_p()
	_ffi.check(lib.rk_mcts_create(C.byref(h), 2, 500, 64))
	np.random.seed(5)
	starts = np.array([orc.scramble(6, True)[0] for _ in range(2)])
	_ffi.check(lib.rk_mcts_reset(h, starts.ctypes.data, None, 1.0, 100.0, st()))
	probs = torch.full((24, 12), 1 / 12, device="cuda")
	values = torch.zeros(24, device="cuda")
	_ffi.check(lib.rk_mcts_backup_select(h, probs.data_ptr(), values.data_ptr(), st()))
	status = np.zeros((2, 6), np.int64)
	_ffi.check(lib.rk_mcts_status(h, status.ctypes.data, st()))
	assert (status[:, 5] == 3).all() and (status[:, 0] == 1).all() and (status[:, 2] == 1).all()
	_ffi.check(lib.rk_mcts_destroy(h))


def test_hipgraph_is_kept_from_search_to_search():
	"""The captured step holds addresses and by-value scalars, nothing of the trees: searches on an unchanged engine and net
	replay ONE graph (single-tree agent and batch); another net or a grown pool capture once more; all equal the oracle."""
	tree = MCTS(PolicyStubNet(), 1.5, True, capacity=1_500, use_hipgraph=True)
	tree.max_capacity = 6_000
	def run(net_of, seed, depth, budget):
		np.random.seed(seed)
		start, _, _ = orc.scramble(depth, True)
		ref = MCTSOracle(net_of(), 1.5, True)
		assert tree.search(start, None, budget) == ref.search(start, budget), (seed, depth)
		n = len(ref)
		a = tree._export()
		assert a["n"] == n
		for k in ("states", "neighbors", "N", "W", "L", "V", "P"):
			assert (a[k][1:n + 1] == getattr(ref, k)[1:n + 1]).all(), (k, seed)
		assert list(tree.action_queue) == list(ref.action_queue)
	for seed, depth, budget in ((1, 4, 1_500), (2, 9, 900), (3, 2, 1_500), (4, 12, 1_200)):
		run(PolicyStubNet, seed, depth, budget)
	assert tree._batch.captures == 1
	tree.net = StubNet()
	run(StubNet, 5, 8, 1_000)
	run(StubNet, 6, 10, 1_500)
	assert tree._batch.captures == 2
	run(StubNet, 7, 14, 5_000)                                        # grows on the way (unless solved early): one capture per growth
	grown = tree.grown
	assert tree._batch.captures == 2 + grown
	run(StubNet, 8, 14, 5_000)
	assert tree._batch.captures == 2 + grown + tree.grown
	# a batch, searched three times with different starts and budgets
	T = 5
	batch = MCTSBatch(StubNet(), 2.0, T, capacity=3_400)
	for rep in range(3):
		starts = []
		for i in range(T):
			np.random.seed(100 * rep + i)
			starts.append(orc.scramble(3 + 2 * i + rep, True)[0])
		starts = np.array(starts)
		budgets = np.array([900, 3000, 1500, 2500, 600]) + 100 * rep
		got = batch.search(starts, max_states=budgets, use_graph=True, poll=7)
		for i in range(T):
			r = MCTSOracle(StubNet(), 2.0, False)
			assert r.search(starts[i], int(budgets[i])) == bool(got[i]), (rep, i)
			a = batch.tree_arrays(i)
			n = len(r)
			assert a["n"] == n and (a["N"][1:n + 1] == r.N[1:n + 1]).all() and (a["W"][1:n + 1] == r.W[1:n + 1]).all(), (rep, i)
			assert list(batch.action_queue_of(i)) == list(r.action_queue), (rep, i)
	assert batch.captures == 1


class _ExactModel(torch.nn.Module):
	"""A net of the reference's structure (shared_net / policy_net / value_net, ref:model.py:117-141) whose arithmetic is EXACT in
	float32 whatever the batch shape or the order of any sum: small-integer weights, ReLU, no normalisation.  The fused first layer
	and the GEMMs of a half batch therefore produce the very bits the full batch produces."""
	def __init__(self):
		super().__init__()
		g = torch.Generator().manual_seed(9)
		ints = lambda *shape: torch.randint(-1, 2, shape, generator=g).float()
		self.shared_net = torch.nn.Sequential(torch.nn.Linear(480, 64), torch.nn.ReLU(), torch.nn.Linear(64, 8), torch.nn.ReLU())   # (the fused layer wants a multiple of 64 outputs)
		self.policy_net = torch.nn.Sequential(torch.nn.Linear(8, 12))
		self.value_net = torch.nn.Sequential(torch.nn.Linear(8, 1))
		with torch.no_grad():
			for lin in (self.shared_net[0], self.shared_net[2], self.policy_net[0], self.value_net[0]):
				lin.weight.copy_(ints(*lin.weight.shape))
				lin.bias.copy_(ints(*lin.bias.shape))
			self.policy_net[0].weight.mul_(0.125)                        # logits in eighths: distinct priors, still exact
			self.value_net[0].weight.mul_(0.25)

	def forward(self, x, policy=True, value=True):
		x = self.shared_net(x)
		out = ([self.policy_net(x)] if policy else []) + ([self.value_net(x)] if value else [])
		return out if len(out) > 1 else out[0]


@pytest.mark.parametrize("n_trees", [7, 16])
def test_two_halves_on_two_streams_change_no_tree(n_trees):
	"""the captured step that advances the batch as two halves on two streams, skewed by half a step (one half's
	backup + descent under the other half's net forward), against the single-stream captured step and the eager step: every tree's
	states, neighbours, leaves, P, V, N, W, L, status and action queue are the same -- also when the budget ends trees at different
	simulations, when the pools grow in place on the way, and when the kept graph is replayed by a second search."""
	net = _ExactModel().cuda().eval()
	starts = []
	for i in range(n_trees):
		np.random.seed(500 + i)
		starts.append(orc.scramble(3 + i % 9, True)[0])
	starts = np.array(starts)
	budgets = np.array([1500 + 700 * (i % 4) for i in range(n_trees)])

	def run(overlap, use_graph, **kw):
		agent = MCTSBatch(net, 0.8, n_trees, capacity=kw.pop("capacity", 5000), fused_first_layer=True, overlap_halves=overlap, **kw)
		out = []
		for again in range(2):                                           # the second search replays the kept graph
			solved = agent.search(starts, max_states=budgets, max_sims=400, use_graph=use_graph, poll=7)
			out.append((solved.copy(), agent.status.copy(), [agent.tree_arrays(t) for t in range(n_trees)], [list(agent.action_queue_of(t)) for t in range(n_trees)]))
		return agent, out

	_, eager = run(False, False)
	_, single = run(False, True)
	halves_agent, halves = run(True, True)
	assert halves_agent.captures == 1 and halves_agent._graph_cache[0][0] == "halves"
	grow_agent, grown = run(True, True, capacity=600, max_capacity=8000)
	assert grow_agent.capacity > 600                                         # the pools grew in place during the first search
	for other in (single, halves, grown):
		for (s0, st0, tr0, q0), (s1, st1, tr1, q1) in zip(eager, other):
			assert (s0 == s1).all() and (st0[:, :5] == st1[:, :5]).all() and q0 == q1
			for a, b in zip(tr0, tr1):
				assert a["n"] == b["n"]
				for k in ("states", "neighbors", "leaves", "P", "V", "N", "W", "L"):
					assert (a[k][:a["n"] + 1] == b[k][:a["n"] + 1]).all(), k
	assert 0 < eager[0][0].sum() < n_trees or n_trees == 7                   # solved and unsolved trees in the batch
