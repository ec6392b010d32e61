"""
Fused one-hot -> first Linear layer (reference: `as_oh`, librubiks/cube/cube.py:265-277, feeding the
first `nn.Linear(480, H)` of `Model`, librubiks/model.py:127 / :150).

    first = OhLinear(net.shared_net[0])            # copies the layer's weight and bias into the engine's layouts
    y = first(states)                              # states: (n, 20) int8 on the GPU -> (n, H); no (n, 480) one-hot in HBM

`route="gather"` is the exact float32 path (a fixed-order sum of the 20 selected weight rows + bias); `route="mfma"` runs on
the matrix cores in bf16 with the one-hot operand synthesised in registers (`"mfma_direct"` / `"mfma_tiled"` force one of its two
forms -- few rows / many rows, same bits -- which `"mfma"` picks by the batch size).  `fuse_first_linear(net)` wraps a net of the
reference's shape (`shared_net` = Sequential starting with a Linear, `policy_net`, `value_net`) into a callable that takes
states instead of one-hot rows: `AStar(..., fused_first_layer=True)` and `adi_traindata(..., fused_first_layer=True)` use it.

`TailLinear` is the same idea at the net's other end: a head's last, narrow Linear (12 logits, 1 value, or both heads' 13 columns
side by side) with the activation in front of it as ONE launch (`rk_tail_linear`) instead of an elementwise kernel over (n, K) and a
GEMM with a 13-column output; `fuse_first_linear(..., fold_batchnorm=True)` uses it for bfloat16 nets (`FUSE_TAIL`).

`fuse_first_linear(net, epilogue=True)` also moves the activation and the eval-mode BatchNorm1d that follow the layer
(model.py:157-159) into the kernel's epilogue, `fold_batchnorm=True` folds the remaining eval-mode BatchNorm1d layers into
the Linear layers behind them: the same function in exact arithmetic, different float rounding (so not the default).
"""
from __future__ import annotations

import ctypes as C

import torch

from librubiks_amd import _ffi

FUSE_TAIL = True       # "folded" nets: the heads' last Linear with its activation as one launch (benchmarks switch it off for A/B)
MFMA_FORM = None       # None: route "mfma" picks its form by the batch size; "mfma_direct" / "mfma_tiled": benchmarks force one for A/B

_ROUTES = {"gather": _ffi.OHL_GATHER, "mfma": _ffi.OHL_MFMA, "mfma_direct": _ffi.OHL_MFMA_DIRECT, "mfma_tiled": _ffi.OHL_MFMA_TILED}
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

	def set_epilogue(self, activation: torch.nn.Module = None, batchnorm: torch.nn.BatchNorm1d = None):
		"""
		Later calls return  scale * act(x W^T + b) + shift: `activation` nn.ELU / nn.ReLU / None, `batchnorm` an EVAL-mode
		BatchNorm1d (its running statistics and affine parameters are copied now, like the weights were) or None.
		"""
		act, alpha = _ffi.OHL_ACT_NONE, 1.0
		if isinstance(activation, torch.nn.ELU):
			act, alpha = _ffi.OHL_ACT_ELU, float(activation.alpha)
		elif isinstance(activation, torch.nn.ReLU):
			act = _ffi.OHL_ACT_RELU
		elif activation is not None:
			raise TypeError(f"no fused epilogue for {type(activation).__name__}: ELU and ReLU are built")
		scale = shift = None
		if batchnorm is not None:
			scale, shift = batchnorm_affine(batchnorm)
			if scale.numel() != self.out_features:
				raise ValueError("BatchNorm1d width differs from the layer's")
		_ffi.check(_ffi.lib().rk_ohl_set_epilogue(self._h, act, alpha, scale.data_ptr() if scale is not None else None,
		                                          shift.data_ptr() if shift is not None else None, _ffi.stream_ptr()))
		return self

	def __call__(self, states: torch.Tensor, out: torch.Tensor = None, route: str = None) -> torch.Tensor:
		if states.dtype != torch.int8 or not states.is_cuda or not states.is_contiguous() or states.shape[-1] != 20:
			raise ValueError("states must be a contiguous (n, 20) int8 tensor on the GPU")
		return self.from_pointer(states.data_ptr(), states.numel() // 20, out, route, states.device)

	def from_pointer(self, d_states: int, n: int, out: torch.Tensor = None, route: str = None, device=None) -> torch.Tensor:
		"""The layer on n 20-byte states at a device address (an engine's own buffer: rk_mcts_children), no tensor around them."""
		route = route or self.route
		if route == "mfma" and MFMA_FORM is not None:
			route = MFMA_FORM
		dtype = torch.bfloat16 if route.startswith("mfma") else self.dtype
		if out is None:
			out = torch.empty((n, self.out_features), dtype=dtype, device=device or "cuda")
		_ffi.check(_ffi.lib().rk_ohl_forward(self._h, d_states, out.data_ptr(), _CODES[out.dtype], n, _ROUTES[route], _ffi.stream_ptr()))
		return out


def _act_code(activation):
	if activation is None:
		return _ffi.OHL_ACT_NONE, 1.0
	if isinstance(activation, torch.nn.ELU):
		return _ffi.OHL_ACT_ELU, float(activation.alpha)
	if isinstance(activation, torch.nn.ReLU):
		return _ffi.OHL_ACT_RELU, 1.0
	return None


class TailLinear(torch.nn.Module):
	"""
	activation -> Linear(K, M) for the narrow last layer of a head (reference: librubiks/model.py:124-125,128-129; M = 12 logits, 1 value,
	or 13 for both heads side by side) as one launch, `rk_tail_linear`: bfloat16 in and out, the activation in float32 rounded to
	bfloat16 as torch's own activation kernel stores it, float32 accumulation.  Same function as the two modules it replaces, another
	summation order (so it belongs to the "folded" mode, which says so).  Snapshots the layer's parameters like the other fused pieces.
	"""
	IN_FEATURES = (512, 1024, 2048)

	@staticmethod
	def fits(activation, linear) -> bool:
		return isinstance(linear, torch.nn.Linear) and linear.weight.is_cuda and linear.weight.dtype == torch.bfloat16 \
		       and linear.in_features in TailLinear.IN_FEATURES and 1 <= linear.out_features <= 16 and _act_code(activation) is not None

	def __init__(self, activation, linear: torch.nn.Linear):
		super().__init__()
		if not TailLinear.fits(activation, linear):
			raise ValueError("TailLinear replaces [ELU | ReLU | nothing] -> nn.Linear(512 | 1024 | 2048, <= 16) in bfloat16 on the GPU")
		self.act, self.alpha = _act_code(activation)
		self.in_features, self.out_features = linear.in_features, linear.out_features
		self.weight = linear.weight.detach().clone().contiguous()
		self.bias = linear.bias.detach().clone().contiguous() if linear.bias is not None else None

	def forward(self, x: torch.Tensor) -> torch.Tensor:
		if x.dtype != torch.bfloat16 or not x.is_cuda or x.dim() != 2 or x.shape[1] != self.in_features or x.stride(1) != 1 \
		   or x.stride(0) % 8 or x.data_ptr() % 16:
			raise ValueError(f"TailLinear takes (n, {self.in_features}) bfloat16 rows on the GPU, 16-byte aligned")
		out = torch.empty((len(x), self.out_features), dtype=torch.bfloat16, device=x.device)
		_ffi.check(_ffi.lib().rk_tail_linear(x.data_ptr(), len(x), self.in_features, x.stride(0), self.weight.data_ptr(),
		                                     self.bias.data_ptr() if self.bias is not None else None, self.out_features, self.act, self.alpha,
		                                     out.data_ptr(), _ffi.stream_ptr()))
		return out


def _fuse_tail(mods: list) -> list:
	"""mods with [activation, Linear(K, <= 16)] at the end replaced by a TailLinear where that is built (else mods as they are)"""
	if len(mods) >= 2 and isinstance(mods[-2], (torch.nn.ELU, torch.nn.ReLU)) and TailLinear.fits(mods[-2], mods[-1]):
		return list(mods[:-2]) + [TailLinear(mods[-2], mods[-1])]
	return list(mods)


def batchnorm_affine(bn: torch.nn.BatchNorm1d):
	"""eval-mode BatchNorm1d as y = scale * x + shift (float32, computed in float64): scale = gamma / sqrt(var + eps), shift = beta - mean * scale"""
	if bn.training or bn.running_mean is None:
		raise ValueError("only an eval-mode BatchNorm1d with running statistics is an affine map (call net.eval() first)")
	var, mean = bn.running_var.detach().double(), bn.running_mean.detach().double()
	gamma = bn.weight.detach().double() if bn.weight is not None else torch.ones_like(var)
	beta = bn.bias.detach().double() if bn.bias is not None else torch.zeros_like(var)
	scale = gamma / torch.sqrt(var + bn.eps)
	return scale.float().contiguous(), (beta - mean * scale).float().contiguous()


class _Affine(torch.nn.Module):
	def __init__(self, scale, shift, dtype):
		super().__init__()
		self.scale, self.shift = scale.to(dtype), shift.to(dtype)

	def forward(self, x):
		return x * self.scale + self.shift


def _fold(modules, pending):
	"""
	modules with every eval-mode BatchNorm1d folded into the Linear behind it; `pending` = (scale, shift) float64 of an
	affine map still to be applied to the input of `modules`.  -> (list of modules, affine map left over at the end)
	"""
	out = []
	for m in modules:
		if isinstance(m, torch.nn.BatchNorm1d) and not m.training and m.running_mean is not None:
			s, t = (x.double() for x in batchnorm_affine(m))
			pending = (s, t) if pending is None else (pending[0] * s, pending[1] * s + t)
			continue
		if pending is not None:
			if isinstance(m, torch.nn.Linear):
				s, t = pending
				w = m.weight.detach().double()
				lin = torch.nn.Linear(m.in_features, m.out_features, bias=True, device=m.weight.device, dtype=m.weight.dtype)
				with torch.no_grad():
					lin.weight.copy_((w * s[None, :]).to(m.weight.dtype))
					lin.bias.copy_(((m.bias.detach().double() if m.bias is not None else 0.0) + w @ t).to(m.weight.dtype))
				m, pending = lin, None
			else:                                  # something else wants the normalised values: apply the map as it stands
				ref = next(iter(m.parameters()), None)
				out.append(_Affine(pending[0], pending[1], ref.dtype if ref is not None else torch.float32))
				pending = None
		out.append(m)
	return out, pending


def _merge_heads(pol, val):
	"""
	The policy and the value head (lists of modules of equal structure: Linear, activation, ..., Linear; model.py:128-129)
	as ONE stack: the first Linear layers side by side (both read the shared trunk), the later ones block-diagonal, so
	both heads cost one GEMM per layer and one activation kernel.  -> (Sequential giving (n, 12 + 1) rows, 12) or None if
	the heads do not match.  Zeros off the diagonal add exact zeros; the accumulation order inside the GEMM may differ.
	"""
	if len(pol) != len(val):
		return None
	out, first, wp, wv = [], True, 0, 0
	for a, b in zip(pol, val):
		if isinstance(a, torch.nn.Linear) and isinstance(b, torch.nn.Linear) and a.weight.dtype == b.weight.dtype:
			if first and a.in_features != b.in_features:
				return None
			dev, dt = a.weight.device, a.weight.dtype
			if first:
				w = torch.cat([a.weight.detach(), b.weight.detach()], 0)
			else:
				if a.in_features != wp or b.in_features != wv:
					return None
				w = torch.zeros((a.out_features + b.out_features, wp + wv), device=dev, dtype=dt)
				w[:a.out_features, :wp] = a.weight.detach()
				w[a.out_features:, wp:] = b.weight.detach()
			bias = torch.cat([a.bias.detach() if a.bias is not None else torch.zeros(a.out_features, device=dev, dtype=dt),
			                  b.bias.detach() if b.bias is not None else torch.zeros(b.out_features, device=dev, dtype=dt)])
			lin = torch.nn.Linear(w.shape[1], w.shape[0], bias=True, device=dev, dtype=dt)
			with torch.no_grad():
				lin.weight.copy_(w)
				lin.bias.copy_(bias)
			out.append(lin)
			first, wp, wv = False, a.out_features, b.out_features
		elif isinstance(a, torch.nn.ELU) and isinstance(b, torch.nn.ELU) and a.alpha == b.alpha:
			out.append(torch.nn.ELU(alpha=a.alpha))
		elif isinstance(a, torch.nn.ReLU) and isinstance(b, torch.nn.ReLU):
			out.append(torch.nn.ReLU())
		elif isinstance(a, _Affine) and isinstance(b, _Affine):
			m = _Affine(torch.cat([a.scale, b.scale]), torch.cat([a.shift, b.shift]), a.scale.dtype)
			out.append(m)
		else:
			return None
	return (torch.nn.Sequential(*out), wp) if not first else None


def fused_net(net, mode):
	"""`fused_first_layer` option of the agents and of adi_traindata: True -> the layer alone (exact for float32 nets),
	"epilogue" -> with its activation and BatchNorm in the kernel, "folded" -> and every other eval-mode BatchNorm folded"""
	if mode not in (True, "epilogue", "folded"):
		raise ValueError('fused_first_layer is False, True, "epilogue" or "folded"')
	net.eval()
	return fuse_first_linear(net, epilogue=mode in ("epilogue", "folded"), fold_batchnorm=mode == "folded")


def fuse_first_linear(net, route: str = None, epilogue: bool = False, fold_batchnorm: bool = False):
	"""
	-> callable(states, policy=True, value=True) with the semantics of `net(as_oh(states), policy, value)` for a net of
	the reference's structure (model.py:117-141): shared_net[0] is the Linear(480, H) that gets fused, the rest runs as is.
	epilogue: shared_net[1] (ELU / ReLU) and, in eval mode, shared_net[2] (BatchNorm1d) run inside the layer's kernel.
	fold_batchnorm: the other eval-mode BatchNorm1d layers are folded into the Linear layers that follow them.
	Both snapshot the net as it is now (like the weights of the first layer) and change float rounding, not the function.
	"""
	shared = getattr(net, "shared_net", None)
	if not isinstance(shared, torch.nn.Sequential) or not isinstance(shared[0], torch.nn.Linear):
		raise TypeError("fuse_first_linear needs a net whose shared_net starts with nn.Linear(480, H) (the reference's Model)")
	first, taken = OhLinear(shared[0], route), 1
	dtype = shared[0].weight.dtype
	if epilogue and len(shared) > 1 and isinstance(shared[1], (torch.nn.ELU, torch.nn.ReLU)):
		bn = shared[2] if len(shared) > 2 and isinstance(shared[2], torch.nn.BatchNorm1d) and not shared[2].training \
		     and shared[2].running_mean is not None else None
		first.set_epilogue(shared[1], bn)
		taken = 3 if bn is not None else 2
	rest, policy_net, value_net = shared[taken:], net.policy_net, net.value_net
	if fold_batchnorm:
		mods, pending = _fold(list(rest), None)
		rest = torch.nn.Sequential(*mods)
		pol, left_p = _fold(list(policy_net), pending)
		val, left_v = _fold(list(value_net), pending)
		if left_p is not None: pol.append(_Affine(left_p[0], left_p[1], dtype))
		if left_v is not None: val.append(_Affine(left_v[0], left_v[1], dtype))
		merged = _merge_heads(pol, val)                # both heads as one stack, used when both are asked for (MCTS)
		if FUSE_TAIL:                                  # the last, narrow Linear of every stack with its activation: one launch instead of two
			pol, val = _fuse_tail(pol), _fuse_tail(val)
			if merged is not None:
				merged = (torch.nn.Sequential(*_fuse_tail(list(merged[0]))), merged[1])
		policy_net, value_net = torch.nn.Sequential(*pol), torch.nn.Sequential(*val)
	else:
		merged = None

	def tail(x: torch.Tensor, policy: bool = True, value: bool = True):
		x = rest(x.to(dtype))
		if policy and value and merged is not None:
			both = merged[0](x)                        # (n, 12 + 1): the heads are views of one tensor
			return [both[:, :merged[1]], both[:, merged[1]:]]
		out = ([policy_net(x)] if policy else []) + ([value_net(x)] if value else [])
		return out if len(out) > 1 else out[0]

	def forward(states: torch.Tensor, policy: bool = True, value: bool = True):
		return tail(first(states), policy, value)

	forward.first, forward.tail = first, tail          # engines that own the states call first.from_pointer(...) and tail(...)
	forward.modules = (rest, policy_net, value_net)
	forward.merged_heads = merged[0] if merged is not None else None
	return forward
