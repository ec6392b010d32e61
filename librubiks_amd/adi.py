"""
Autodidactic-Iteration data generation on the device (reference: `Train.ADI_traindata`, librubiks/train.py:256-339).

The reference scrambles on the host, builds 12 x (games*depth) action tuples, fans out with NumPy and one-hot encodes
2.7 M children on the host before the net sees them.  Here the random walks, the 12-child fan-out with its fused goal
test and the one-hot encoding are HIP kernels on device-resident buffers; only the RNG draws (NumPy legacy generator,
the reference's order, so seeds reproduce the same scrambles) and the final small target tensors touch the host.

    oh_states, policy_targets, value_targets, loss_weights = adi_traindata(net, games, depth, alpha, reward_method)

Same return values as the reference (one-hot states on `gpu`, the three target tensors on the CPU).
"""
from __future__ import annotations

import numpy as np
import torch

from librubiks_amd import gpu, no_grad, _ffi, cube


@no_grad
def adi_traindata(net, rollout_games: int, rollout_depth: int, alpha: float, reward_method: str = "lapanfix",
                  ff_batches: int = 1, fused_first_layer=False):
	assert reward_method in ("paper", "lapanfix", "schultzfix", "reward0")
	_ffi.require_gpu()
	net.eval()
	with_solved = reward_method == "lapanfix"
	# scrambling: draws as cube.sequence_scrambler (cube.py:226-227), walks on the device           train.py:277
	faces = np.random.randint(0, 6, (rollout_depth, rollout_games))
	dirs = np.random.randint(0, 2, (rollout_depth, rollout_games))
	acts = torch.from_numpy((2 * faces + (1 - dirs)).astype(np.uint8)).to(gpu)
	# walks, goal test of the scrambled states, fan-out and its goal test: ONE launch              train.py:277, :281, :285, :292
	states, solved_scrambled, substates, solved_sub = cube.device.rollout_fanout(acts, with_solved)  # (games*depth, 20), ..., (12*games*depth, 20), ...
	n = len(states)
	oh_states = cube.device.as_oh(states)
	solved_scrambled, solved_sub = solved_scrambled.bool(), solved_sub.bool()
	rewards = torch.where(solved_sub, torch.tensor(0.0 if reward_method == "reward0" else 1.0, device=gpu),
	                      torch.tensor(-1.0, device=gpu))                                         # train.py:294-296
	# value of every child, in slices so that the one-hot batch stays bounded                      train.py:301-303
	values = torch.empty(12 * n, dtype=torch.float32, device=gpu)
	step = -(-12 * n // max(1, ff_batches))
	if fused_first_layer:
		# the net's first Linear reads the 2.7 M children's 20-byte states directly: the (12 n, 480) one-hot never exists
		from librubiks_amd.oh_linear import fused_net
		from_states = fused_net(net, fused_first_layer)
		for lo in range(0, 12 * n, step):
			hi = min(lo + step, 12 * n)
			values[lo:hi] = from_states(substates[lo:hi], policy=False, value=True).reshape(-1).float()
	else:
		from librubiks_amd.solving.agents import _oh_dtype
		oh_dtype = _oh_dtype(net)                  # a bf16 net gets bf16 rows straight from the kernel (0 / 1 are exact)
		buf = torch.empty((min(step, 12 * n), 480), dtype=oh_dtype, device=gpu)
		for lo in range(0, 12 * n, step):
			hi = min(lo + step, 12 * n)
			cube.device.as_oh(substates[lo:hi], buf[:hi - lo], oh_dtype)
			values[lo:hi] = net(buf[:hi - lo], policy=False, value=True).reshape(-1).float()
	values = (values + rewards).reshape(-1, 12)                                                   # train.py:313-314
	policy_targets = torch.argmax(values, dim=1)
	value_targets = values[torch.arange(n, device=gpu), policy_targets]
	if reward_method == "lapanfix":
		value_targets[solved_scrambled] = 0                                                       # train.py:319
	elif reward_method == "schultzfix":
		value_targets[torch.arange(0, n, rollout_depth, device=gpu)] = 0                          # train.py:322-324
	weighted = np.tile(1 / np.arange(1, rollout_depth + 1), rollout_games)                        # train.py:329-332
	unweighted = np.ones_like(weighted)
	ws, us = weighted.sum(), len(unweighted)
	loss_weights = ((1 - alpha) * weighted / ws + alpha * unweighted / us) * (ws + us)
	return oh_states, policy_targets.cpu(), value_targets.cpu(), torch.from_numpy(loss_weights).float()
