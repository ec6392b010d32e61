"""
BASELINE.json configs[2] and configs[3] at FULL size on the GPU (the other GPU tests pin the same engines at sizes the
oracle replays in a second or two):

  configs[2]  A* expand loop, depth-14 scramble, open set beyond 100 000 nodes
              * exact-integer stub net: states, G, parents, actions, open queue bit-identical to the CPU oracle
                (itself pinned to traces of the unmodified reference, tests/test_search_oracle.py);
              * random-init fc_small-shaped torch net (benchmarks/nets.py, the reference's default architecture,
                model.py:17): the invariants of the reference's tests/test_agents.py:100-145.
  configs[3]  MCTS, 256 parallel scrambles x 4096 simulations each, simulation step replayed as a hipGraph
              * per-tree equality with the oracle on a sample of trees; status: 4096 simulations on every unsolved tree;
              * pool invariants of tests/test_agents.py:49-94 on all 256 trees.
"""
import numpy as np
import pytest
import torch

from librubiks_amd import cube
from librubiks_amd.solving.agents import AStar, MCTSBatch
from oracle import c_oracle, cube_oracle as orc
from oracle.search_oracle import AStarOracle, MCTSOracle, StubNet

pytestmark = pytest.mark.gpu


def _apply(state, queue):
	for a in queue:
		state = cube.rotate(state, *cube.action_space[a])
	return state


# ---- configs[2] ----------------------------------------------------------------------------------------------------
def test_astar_depth14_open_set_100k_matches_oracle():
	np.random.seed(114)
	start, _, _ = orc.scramble(14, True)
	lam, n, budget = 0.16, 1000, 140_000                     # lambda of configs/main_eval.ini:8
	ref = AStarOracle(StubNet(), lam, n)
	ref_solved = ref.search(start, budget)
	assert not ref_solved and len(ref.open) > 100_000        # the open set passes 100 k nodes
	agent = AStar(StubNet(), lam, n)
	assert agent.search(start, None, budget) == ref_solved
	states, G, parents, pact = ref.arrays()
	k = len(states)
	assert len(agent) == k
	assert (agent.states[1:k + 1] == states).all() and (agent.G[1:k + 1] == G).all()
	assert (agent.parents[2:k + 1] == parents).all() and (agent.parent_actions[2:k + 1] == pact).all()
	want, got = sorted(ref.open), agent.open_queue
	assert len(got) == len(want) > 100_000
	assert [i for _, i in got] == [i for _, i in want] and [c for c, _ in got] == [c for c, _ in want]


def test_astar_depth14_open_set_100k_fc_small_net():
	from benchmarks.nets import FcSmall
	net = FcSmall(seed=0).cuda().eval()
	np.random.seed(14)
	start, _, _ = cube.scramble(14, force_not_solved=True)
	agent = AStar(net, lambda_=0.16, expansions=1000)
	solved = agent.search(start, time_limit=None, max_states=150_000)
	q = agent.open_queue
	k = len(agent)
	if solved:
		assert cube.is_solved(_apply(start, agent.action_queue))
	else:
		assert len(q) > 100_000 and k + 12 * 1000 > 150_000
	assert q == sorted(q) and len({i for _, i in q}) == len(q)
	st, G, par, act = agent.states, agent.G, agent.parents, agent.parent_actions
	# tests/test_agents.py:122-134: the root is node 1 with G 0, its 12 children have G 1 and parent 1
	assert (st[1] == start).all() and G[1] == 0
	idx = {st[i].tobytes(): i for i in range(1, k + 1)}
	assert len(idx) == k                                      # states <-> indices is a bijection
	for a in range(12):
		i = idx[cube.rotate(start, *cube.action_space[a]).tobytes()]
		assert G[i] == 1 and par[i] == 1
	# every parent link is a real move and G never undercuts the parent's G + 1 (relaxation may lower a parent later)
	pick = np.arange(2, k + 1)
	moved = c_oracle.multi_rotate(st[par[pick]], act[pick].astype(np.uint8), threads=8)
	assert (moved == st[pick]).all() and (G[pick] >= G[par[pick]] + 1).all()
	# cost of the queue entries = lambda * G - value of a fresh forward (agents.py:369-383), float32 net on the same GPU
	probe = np.array([i for _, i in q[:2000]])
	with torch.no_grad():
		v = net(cube.as_oh(st[probe]), policy=False, value=True).float().cpu().numpy().reshape(-1)
	assert np.allclose([c for c, _ in q[:2000]], 0.16 * G[probe] - v, rtol=0, atol=2e-3)


# ---- configs[3] ----------------------------------------------------------------------------------------------------
T, SIMS, C_EXPL = 256, 4096, 5.0
CAP = 12 * SIMS + 16


@pytest.fixture(scope="module")
def mcts_run():
	starts = []
	for i in range(T):
		np.random.seed(1000 + i)
		starts.append(orc.scramble(10 + i % 11, True)[0])            # depths 10..20
	starts = np.array(starts)
	agent = MCTSBatch(StubNet(), C_EXPL, T, capacity=CAP)
	solved = agent.search(starts, max_states=CAP, max_sims=SIMS, use_graph=True, poll=256)
	return starts, agent, solved


def test_mcts_256x4096_status(mcts_run):
	starts, agent, solved = mcts_run
	st = agent.status
	assert agent.simulations == SIMS
	assert (st[:, 5] == 0).all()
	unsolved = ~solved.astype(bool)
	assert unsolved.sum() >= T // 2                                   # deep scrambles: most trees use the whole budget
	assert (st[unsolved, 3] == SIMS).all()                            # 4096 simulations each
	assert (st[~unsolved, 3] <= SIMS).all() and (st[:, 2] <= CAP).all()
	for t in np.flatnonzero(solved)[:8]:
		assert orc.is_solved(_apply(starts[t], agent.action_queue_of(int(t))))


@pytest.mark.parametrize("tree", [0, 77, 130, 201, 255])
def test_mcts_256x4096_tree_equals_oracle(mcts_run, tree):
	starts, agent, solved = mcts_run
	ref = MCTSOracle(StubNet(), C_EXPL, False)
	ref_solved = ref.search(starts[tree], CAP, max_sims=SIMS)
	assert bool(solved[tree]) == ref_solved and int(agent.status[tree, 3]) == ref.sims
	a = agent.tree_arrays(tree)
	n = len(ref)
	assert a["n"] == n
	for name in ("states", "neighbors", "leaves", "N", "W", "L", "V", "P"):
		assert (a[name][1:n + 1] == getattr(ref, name)[1:n + 1]).all(), name
	assert list(agent.action_queue_of(tree)) == list(ref.action_queue)


def test_mcts_256x4096_pool_invariants_all_trees(mcts_run):
	"""tests/test_agents.py:49-94 on every tree: dense indices, bijection, neighbour links are real moves, leaf flags, W, V."""
	starts, agent, solved = mcts_run
	net = StubNet()
	p_uniform = float(torch.zeros(1, 12).softmax(dim=1)[0, 0])          # float32 softmax of the stub's zero logits
	for t in range(T):
		a = agent.tree_arrays(t)
		n, st, nb = a["n"], a["states"], a["neighbors"]
		assert (st[1] == starts[t]).all()
		assert len(np.unique(st[1:n + 1].view("V20"))) == n              # states[i] <-> index is a bijection
		assert nb[1:n + 1].min() >= 0 and nb[1:n + 1].max() <= n
		i, j = np.nonzero(nb[1:n + 1])
		i = i + 1
		moved = c_oracle.multi_rotate(st[i], j.astype(np.uint8), threads=8)
		assert (moved == st[nb[i, j]]).all()                              # neighbors[i, j] = rotate(states[i], action j)
		assert (nb[nb[i, j], j ^ 1] == i).all()                           # and the reverse link points back
		assert (nb[1:n + 1].all(axis=1) != a["leaves"][1:n + 1]).all()
		assert a["W"][1:n + 1].all() or bool(solved[t])
		if t % 16 == 0:                                                   # V and P of a fresh forward (exact with the stub)
			v = np.asarray(net(orc.as_oh(st[1:n + 1]), policy=False, value=True)).reshape(-1)
			assert (a["V"][1:n + 1] == v).all() and (a["P"][1:n + 1] == p_uniform).all()
