"""
Fused one-hot -> first Linear layer (reference: `as_oh`, librubiks/cube/cube.py:265-277, feeding the
first `nn.Linear(480, H)` of `Model`, librubiks/model.py:127 / :150).

    first = OhLinear(net.shared_net[0])            # copies the layer's weight and bias into the engine's layouts
    y = first(states)                              # states: (n, 20) int8 on the GPU -> (n, H); no (n, 480) one-hot in HBM

`route="gather"` is the exact float32 path (a fixed-order sum of the 20 selected weight rows + bias); `route="mfma"` runs on
the matrix cores in bf16 with the one-hot operand synthesised in registers.  `fuse_first_linear(net)` wraps a net of the
reference's shape (`shared_net` = Sequential starting with a Linear, `policy_net`, `value_net`) into a callable that takes
states instead of one-hot rows: `AStar(..., fused_first_layer=True)` and `adi_traindata(..., fused_first_layer=True)` use it.
"""
from __future__ import annotations

import ctypes as C

import torch

from librubiks_amd import _ffi

_ROUTES = {"gather": _ffi.OHL_GATHER, "mfma": _ffi.OHL_MFMA}
_CODES = {torch.float32: _ffi.OH_F32, torch.bfloat16: _ffi.OH_BF16}


class OhLinear:
	def __init__(self, linear: torch.nn.Linear, route: str = None):
		if linear.in_features != 480:
			raise ValueError("the fused layer replaces nn.Linear(480, H) behind the 20-byte representation's one-hot")
		w = linear.weight.detach()
		if w.dtype not in _CODES or not w.is_cuda:
			raise ValueError("weights must be float32 or bfloat16 on the GPU")
		self.out_features, self.dtype = linear.out_features, w.dtype
		# the exact route for float32 nets, the matrix cores for bf16 nets, unless told otherwise
		self.route = route or ("gather" if w.dtype == torch.float32 else "mfma")
		b = linear.bias.detach().to(w.dtype).contiguous() if linear.bias is not None else None
		h = C.c_void_p()
		_ffi.check(_ffi.lib().rk_ohl_create(C.byref(h), w.contiguous().data_ptr(), _CODES[w.dtype], b.data_ptr() if b is not None else None,
		                                    self.out_features, _ffi.stream_ptr()))
		self._h = h

	def __del__(self):
		try:
			if getattr(self, "_h", None) is not None:
				_ffi.lib().rk_ohl_destroy(self._h)
				self._h = None
		except Exception:
			pass

	def __call__(self, states: torch.Tensor, out: torch.Tensor = None, route: str = None) -> torch.Tensor:
		route = route or self.route
		if states.dtype != torch.int8 or not states.is_cuda or not states.is_contiguous() or states.shape[-1] != 20:
			raise ValueError("states must be a contiguous (n, 20) int8 tensor on the GPU")
		n = states.numel() // 20
		dtype = torch.bfloat16 if route == "mfma" else self.dtype
		if out is None:
			out = torch.empty((n, self.out_features), dtype=dtype, device=states.device)
		_ffi.check(_ffi.lib().rk_ohl_forward(self._h, states.data_ptr(), out.data_ptr(), _CODES[out.dtype], n, _ROUTES[route], _ffi.stream_ptr()))
		return out


def fuse_first_linear(net, route: str = None):
	"""
	-> callable(states, policy=True, value=True) with the semantics of `net(as_oh(states), policy, value)` for a net of
	the reference's structure (model.py:117-141): shared_net[0] is the Linear(480, H) that gets fused, the rest runs as is.
	"""
	shared = getattr(net, "shared_net", None)
	if not isinstance(shared, torch.nn.Sequential) or not isinstance(shared[0], torch.nn.Linear):
		raise TypeError("fuse_first_linear needs a net whose shared_net starts with nn.Linear(480, H) (the reference's Model)")
	first, rest = OhLinear(shared[0], route), shared[1:]

	def forward(states: torch.Tensor, policy: bool = True, value: bool = True):
		x = rest(first(states).to(shared[0].weight.dtype))
		out = ([net.policy_net(x)] if policy else []) + ([net.value_net(x)] if value else [])
		return out if len(out) > 1 else out[0]

	forward.first = first
	return forward
