import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from librubiks_amd import cube
from librubiks_amd.solving.agents import AStar, AStarBatch, MCTSBatch
from oracle import cube_oracle as orc
from oracle.search_oracle import StubNet
def check_pool(states, G, par, act, n, tag):
    assert len({states[i].tobytes() for i in range(1, n + 1)}) == n, tag + ": duplicate states"
    pick = np.arange(2, n + 1)
    moved = orc.multi_rotate(states[par[pick]], act[pick] // 2, 1 - act[pick] % 2)
    assert (moved == states[pick]).all(), tag + ": parent link is not a move"
    assert (G[pick] >= G[par[pick]] + 1).all(), tag + ": G inconsistent"   # relaxation lowers a parent without touching its children
    assert G[1] == 0
np.random.seed(123); start, _, _ = orc.scramble(30, True)
t0 = time.time(); a = AStar(StubNet(), 0.3, 2000); ok = a.search(start, None, 2_000_000); print("AStar 2M:", ok, len(a), a.iterations, round(time.time() - t0, 2), "s")
check_pool(a.states, a.G, a.parents, a.parent_actions, len(a), "AStar")
q = a.open_queue; assert q == sorted(q) and len({i for _, i in q}) == len(q)
print("open queue", len(q), "ok")
S = 128; starts = np.array([orc.scramble(25, True)[0] for _ in range(S)])
t0 = time.time(); b = AStarBatch(StubNet(), 0.5, 50, S, capacity=40_000); sol = b.search(starts, max_states=40_000, use_graph=True, poll=32); print("AStarBatch:", int(sol.sum()), "solved", len(b), "states", round(time.time() - t0, 2), "s")
for i in (0, 17, 127):
    st, G, par, act = b.arrays_of(i); check_pool(st, G, par, act, int(b.status[i, 2]), f"batch[{i}]")
t0 = time.time(); m = MCTSBatch(StubNet(), 2.0, 512, capacity=6000); sol = m.search(starts.repeat(4, axis=0), max_states=6000, use_graph=True, poll=64); print("MCTSBatch 512 trees:", int(sol.sum()), "solved", len(m), "states", round(time.time() - t0, 2), "s")
for i in (0, 300, 511):
    arr = m.tree_arrays(i); n = arr["n"]; nb = arr["neighbors"]; stt = arr["states"]
    assert len({stt[k].tobytes() for k in range(1, n + 1)}) == n
    idx = np.argwhere(nb[1:n + 1] > 0)[:3000]
    src = idx[:, 0] + 1; actn = idx[:, 1]
    assert (orc.multi_rotate(stt[src], actn // 2, 1 - actn % 2) == stt[nb[src, actn]]).all()
print("stress ok")
