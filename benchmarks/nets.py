"""
Random-init value/policy net with the architecture of the reference's default `fc_small` configuration
(librubiks/model.py:17: 480 -> 4096 -> 2048 shared, then 512 -> 12 policy and 512 -> 1 value heads; ELU + BatchNorm1d,
Xavier-uniform weights; model.py:131-161).  Benchmarks only: the net is out of this project's scope (it stays a
PyTorch module in the reference too) and there is no trained checkpoint offline, so weights are random (seed 0).
"""
import torch
import torch.nn as nn


def _stack(sizes, final):
	layers = []
	for i in range(len(sizes) - 1):
		lin = nn.Linear(sizes[i], sizes[i + 1])
		nn.init.xavier_uniform_(lin.weight)
		layers.append(lin)
		if not (final and i == len(sizes) - 2):
			layers += [nn.ELU(), nn.BatchNorm1d(sizes[i + 1])]
	return nn.Sequential(*layers)


class FcSmall(nn.Module):
	def __init__(self, seed: int = 0):
		super().__init__()
		torch.manual_seed(seed)
		self.shared_net = _stack([480, 4096, 2048], False)
		self.policy_net = _stack([2048, 512, 12], True)
		self.value_net = _stack([2048, 512, 1], True)

	def forward(self, x, policy=True, value=True):
		assert policy or value
		x = self.shared_net(x)
		out = []
		if policy:
			out.append(self.policy_net(x))
		if value:
			out.append(self.value_net(x))
		return out if len(out) > 1 else out[0]


class FastStub:
	"""
	The exact stub heuristic of the search traces (value = -(number of cubies off their solved code), policy logits zero) as
	ONE kernel: v = -20 + x @ onehot(solved) with torch.addmv.  Sums of at most twenty ones are exact in float32, so the
	numbers equal oracle.search_oracle.StubNet's bit for bit; that one is written for clarity (five small kernels), this
	one so that engine benchmarks with a "free" net measure the engine and not the stub.
	"""
	def __init__(self):
		from librubiks_amd import cube
		self.sol = cube.as_oh(cube.get_solved()).reshape(-1).float()
		self.bias = {}

	def eval(self):
		return self

	def __call__(self, x, policy=True, value=True):
		n = len(x)
		if n not in self.bias:
			self.bias[n] = torch.full((n,), -20.0, device=x.device)
		v = torch.addmv(self.bias[n], x.float() if x.dtype != torch.float32 else x, self.sol).unsqueeze(1)
		out = ([torch.zeros(n, 12, device=x.device)] if policy else []) + ([v] if value else [])
		return out if len(out) > 1 else out[0]
