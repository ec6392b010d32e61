"""
Host logic added in round 3, checked on the CPU (no GPU, no HIP call): the net signature that decides when a fused first
layer is re-copied (DeepAgent), the sliced value forward, the sharded driver's first net piece.
"""
import numpy as np
import torch

from librubiks_amd.solving.agents import DeepAgent, NET_SLICE_ROWS, _net_signature, _sliced_value_forward
from librubiks_amd.solving.sharded import net_rows


class TinyNet(torch.nn.Module):
	def __init__(self):
		super().__init__()
		self.lin = torch.nn.Linear(6, 3)
		self.bn = torch.nn.BatchNorm1d(3)

	def forward(self, x, policy=True, value=True):
		return self.bn(self.lin(x))[:, :1]


def test_net_signature_follows_every_way_a_net_changes():
	"""reference: train.py:134, :214 train the net in place and reassign agent.net; ADVICE r2 (stale fused weights)."""
	net = TinyNet().eval()
	sig = _net_signature(net)
	assert _net_signature(net) == sig                                     # looking at it changes nothing
	with torch.no_grad():
		net.lin.weight.mul_(0.5)                                          # an optimizer step: in place
	sig2 = _net_signature(net)
	assert sig2 != sig
	with torch.no_grad():
		net.bn.running_mean.add_(1.0)                                     # BatchNorm statistics are buffers, not parameters
	sig3 = _net_signature(net)
	assert sig3 != sig2
	net.train()
	assert _net_signature(net) != sig3                                    # folding is only valid in eval mode
	net.eval()
	assert _net_signature(net) == sig3
	net.load_state_dict(TinyNet().state_dict())                           # copies in place: versions move
	assert _net_signature(net) != sig3
	assert _net_signature(TinyNet().eval()) != _net_signature(net)        # another module
	# an object that is no torch module at all (the stub nets of the tests): identity only
	class Stub:
		def eval(self): return self
	a, b = Stub(), Stub()
	assert _net_signature(a) == _net_signature(a) != _net_signature(b)


def test_agent_without_fusion_never_copies_and_net_is_assignable():
	a = DeepAgent(TinyNet())
	assert a._from_states is None
	other = TinyNet()
	a.net = other                                                         # train.py:214
	assert a.net is other and a._from_states is None
	try:
		DeepAgent(TinyNet(), fused_first_layer="everything")
		assert False
	except ValueError:
		pass


def test_sliced_value_forward_equals_one_forward():
	calls = []

	def forward(rows, policy=True, value=True):
		calls.append(len(rows))
		assert policy is False and value is True
		return (rows.double() ** 2).sum(dim=1, keepdim=True)

	rows = torch.arange(1000 * 4, dtype=torch.float32).reshape(1000, 4)
	whole = forward(rows, policy=False)
	calls.clear()
	assert torch.equal(_sliced_value_forward(forward, rows, 1000).reshape(-1), whole.reshape(-1)) and calls == [1000]
	calls.clear()
	assert torch.equal(_sliced_value_forward(forward, rows, 300).reshape(-1), whole.reshape(-1)) and calls == [300, 300, 300, 100]
	calls.clear()
	_sliced_value_forward(forward, rows)                                  # default slice: one forward below NET_SLICE_ROWS
	assert calls == [1000] and NET_SLICE_ROWS >= 4096
	# a net that returns [policy, value] lists (the reference's Model with both heads): the value is the last entry
	both = lambda r, policy=True, value=True: [r[:, :2], r[:, :1] + 1]
	assert torch.equal(_sliced_value_forward(both, rows, 256).reshape(-1), rows[:, 0] + 1)


def test_fixed_rows_of_the_sharded_net_batch():
	"""The net of a sharded search runs on a FIXED number of rows per iteration (no count travels to the host): a rank's expected
	share of the 12 N children plus six standard deviations plus 64, in multiples of 64, never more than 12 N."""
	for N in (1, 10, 27, 100, 700, 1000, 5600):
		K = 12 * N
		assert net_rows(K, 1) == K                                        # one rank: the whole batch
		for world in (2, 3, 8, 64):
			f = net_rows(K, world)
			mu = K / world
			assert min(K, mu + 6 * (mu * (1 - 1 / world)) ** 0.5 + 64) <= f <= K and (f % 64 == 0 or f == K)
			assert f < mu + 6 * mu ** 0.5 + 64 + 64 or f == K
	assert net_rows(8400, 8) == 1344 and net_rows(8400, 2) == 4544 and net_rows(67200, 8) == 9024
