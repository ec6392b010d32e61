"""
GPU parity of the batched A* engine (rk_astarb_*, AStarBatch): every search of a batch must equal the reference's
AStar run on that start state alone -- checked against the captured reference traces (batch of one) and against the
CPU oracle for batches with mixed depths and budgets, eager and as a replayed hipGraph.
"""
import numpy as np
import pytest
import torch

from librubiks_amd import cube
from librubiks_amd.solving.agents import AStarBatch
from oracle import cube_oracle as orc
from oracle.search_oracle import AStarOracle, StubNet

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["a", "b", "d"])
def test_batch_of_one_reproduces_reference_traces(golden, tag):
	t = golden["astar_trace"]
	_, _, expansions, max_states = (int(x) for x in t[f"{tag}_params"])
	agent = AStarBatch(StubNet(), float(t[f"{tag}_lambda"]), expansions, 1, capacity=max_states)
	solved = agent.search(t[f"{tag}_start"][None], max_states=max_states, poll=1)
	n = int(t[f"{tag}_n"])
	assert bool(solved[0]) == bool(t[f"{tag}_solved"]) and int(agent.status[0, 2]) == n
	assert int(agent.status[0, 3]) == len(t[f"{tag}_pop_lens"])
	states, G, parents, pact = agent.arrays_of(0)
	assert (states[1:] == t[f"{tag}_states"]).all() and (G[1:] == t[f"{tag}_G"]).all()
	assert (parents[2:] == t[f"{tag}_parents"]).all() and (pact[2:] == t[f"{tag}_parent_actions"]).all()
	assert list(agent.action_queue_of(0)) == t[f"{tag}_action_queue"].tolist()


@pytest.mark.parametrize("use_graph,n,exact", [(False, 10, False), (True, 10, False), (False, 200, False), (True, 200, False),
                                               (False, 10, True), (False, 200, True), (False, 1000, True)])
def test_batch_equals_oracle_per_search(use_graph, n, exact):
	"""exact: the net's batch compacted to the searches' new rows (rk_astarb_step_expand_compact) instead of the padded batch."""
	S, lam = 10, 0.3
	starts, budgets = [], []
	for i in range(S):
		np.random.seed(500 + i)
		starts.append(orc.scramble(3 + i % 6, True)[0])
		budgets.append(3000 + 2500 * i)
	starts = np.array(starts)
	starts[4] = orc.SOLVED
	agent = AStarBatch(StubNet(), lam, n, S, capacity=max(budgets))
	solved = agent.search(starts, max_states=np.array(budgets), use_graph=use_graph, poll=4, exact_batch=exact)
	if exact:                                                            # fewer rows than the padded batch went through the net
		assert 0 < agent.net_rows_total < agent.iterations * S * 12 * n
	n_solved = 0
	for i in range(S):
		ref = AStarOracle(StubNet(), lam, n)
		ref_solved = ref.search(starts[i], budgets[i])
		assert bool(solved[i]) == ref_solved, i
		assert list(agent.action_queue_of(i)) == list(ref.action_queue), i
		if i != 4:
			states, G, parents, pact = agent.arrays_of(i)
			rs, rG, rp, ra = ref.arrays()
			assert (states[1:] == rs).all() and (G[1:] == rG).all() and (parents[2:] == rp).all() and (pact[2:] == ra).all(), i
			assert int(agent.status[i, 3]) == len(ref.pops), i
		n_solved += ref_solved
	assert n_solved >= 3


@pytest.mark.parametrize("fused", [False, "folded"])
def test_batch_with_a_bf16_net(fused):
	"""The batch with a real bfloat16 net (values enter as bf16: rk_astarb_set_values_dtype; optionally the first layer reads
	the states): not comparable bit for bit with single runs (the GEMMs see another batch shape), so the reference's
	invariants (tests/test_agents.py:100-145) per search: budgets kept, root first, every node one move from its parent,
	G = G[parent] + 1 at insertion or better, found paths solve the cube."""
	from benchmarks.nets import FcSmall
	net = FcSmall(seed=11).cuda().eval().to(torch.bfloat16)
	S, n = 5, 30
	starts = []
	for i in range(S):
		np.random.seed(900 + i)
		starts.append(orc.scramble(2 + i, True)[0])
	starts = np.array(starts)
	budgets = np.array([4000 + 1000 * i for i in range(S)])
	agent = AStarBatch(net, 0.3, n, S, capacity=int(budgets.max()), fused_first_layer=fused)
	for use_graph in (True, False):                                      # replayed on the padded batch; eager on the compacted new rows (the default for a real net)
		solved = agent.search(starts, max_states=budgets, use_graph=use_graph, poll=4)
		assert solved[:2].all()                                          # two and three moves from solved
		_check_invariants(agent, starts, budgets, solved)


def _check_invariants(agent, starts, budgets, solved):
	S = len(starts)
	for i in range(S):
		states, G, parents, pact = agent.arrays_of(i)
		m = len(states) - 1
		assert 1 <= m <= budgets[i] and (states[1] == starts[i]).all() and G[1] == 0
		kids = np.arange(2, m + 1)
		moved = orc.multi_rotate(states[parents[kids]], pact[kids] // 2, 1 - pact[kids] % 2)
		assert (moved == states[kids]).all() and (G[kids] == G[parents[kids]] + 1).all()
		assert len({s.tobytes() for s in states[1:]}) == m
		if solved[i]:
			s = starts[i]
			for a in agent.action_queue_of(i):
				s = orc.rotate(s, a // 2, 1 - a % 2)
			assert orc.is_solved(s)
