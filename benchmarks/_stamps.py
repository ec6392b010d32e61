import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from benchmarks.nets import FcSmall
from librubiks_amd import cube, _ffi
from librubiks_amd.solving.agents import AStar
net = FcSmall().cuda().eval().to(torch.bfloat16)
lib = _ffi.lib()
agent = AStar(net, 0.16, 1000, fused_first_layer="folded")
np.random.seed(12345)
agent.search(cube.scramble(14, True)[0], time_limit=None, max_states=40000)
for g in range(3):
	np.random.seed(g)
	agent.search(cube.scramble(14, True)[0], time_limit=None, max_states=150000)
	torch.cuda.synchronize()
	buf = (C.c_ulonglong * 64)()
	lib.rk_debug_stamps(buf)
	a = np.array(buf[:], dtype=np.int64).reshape(4, 16)
	for k, name, n in ((0, "insert", 9),):
		row = a[k]
		print(name, [float(round((row[i] - row[0]) * 0.01, 2)) for i in range(n)])
