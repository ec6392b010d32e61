"""
Drop-in for `librubiks.cube` (reference: librubiks/cube/cube.py) whose arithmetic runs on the MI355X.

Same names, argument meaning, dtypes, shapes and out-of-place behaviour as the reference module; every
function that touches cube state launches a HIP kernel of librubiks_hip.so through ctypes.  NumPy in ->
NumPy out (host arrays are copied to the device and back); torch CUDA tensors in -> torch CUDA tensors out
(no copies, launches are ordered on torch's current stream) -- the second form is what the search engines
and bench.py use.

Random draws (scramble, sequence_scrambler) stay on the host's legacy NumPy generator in the reference's
draw order, so seeds reproduce the reference's scrambles exactly; the moves are applied on the device.
"""
from __future__ import annotations

import functools

import ctypes as C

import numpy as np
import torch

from librubiks_amd import gpu, _ffi
from librubiks_amd._ffi import REPR_2024, REPR_686, INT64_MAX
from librubiks_amd.cube.maps import SimpleState, get_corner_pos, get_side_pos, get_tensor_map, get_633maps, neighbors_686  # noqa: F401  (cube.py:22)

####################
# Action constants #   (cube.py:29-35)
####################
F, B, T, D, L, R = 0, 1, 2, 3, 4, 5
action_names = ('F', 'B', 'T', 'D', 'L', 'R')
action_space = [(a // 2, 1 - a % 2) for a in range(12)]
action_dim = len(action_space)
dtype = np.int8

_ITER_ACTIONS_1 = np.array([[a // 2 for a in range(12)], [1 - a % 2 for a in range(12)]], dtype=np.uint8)

##################
# Representation #   (cube.py:96-128)
##################
_is2024 = True
_stored_repr: bool = True


def set_is2024(is2024: bool):
	global _is2024
	assert type(is2024) is bool
	_is2024 = is2024


def get_is2024():
	return _is2024


def store_repr():
	global _stored_repr
	_stored_repr = _is2024


def restore_repr():
	global _is2024
	_is2024 = _stored_repr


def with_used_repr(fun):
	"""Method decorator: run with the representation in self.is2024, restore afterwards (cube.py:115-124)."""
	@functools.wraps(fun)
	def wrapper(self, *args, **kwargs):
		store_repr()
		set_is2024(self.is2024)
		try:
			return fun(self, *args, **kwargs)
		finally:
			restore_repr()
	return wrapper


def _repr_id() -> int:
	return REPR_2024 if _is2024 else REPR_686


def shape():
	return (20,) if _is2024 else (6, 8, 6)


def _row_bytes() -> int:
	return 20 if _is2024 else 288


def get_oh_shape() -> int:
	return 480 if _is2024 else 288


##########
# Solved #   (cube.py:58-89)
##########
def _load_solved(repr_id: int, shp) -> np.ndarray:
	out = np.empty(shp, dtype=np.int8)
	_ffi.check(_ffi.lib().rk_solved(repr_id, out.ctypes.data))
	return out


_solved2024 = _load_solved(REPR_2024, (20,))
_solved686 = _load_solved(REPR_686, (6, 8, 6))


def get_solved_instance() -> np.ndarray:
	"""The shared instance -- read only by convention (cube.py:77-80)."""
	return _solved2024 if _is2024 else _solved686


def get_solved() -> np.ndarray:
	return get_solved_instance().copy()


################
# Device plumbing
################
def _is_dev(x) -> bool:
	return isinstance(x, torch.Tensor) and x.is_cuda


def _to_dev_states(states) -> torch.Tensor:
	"""(n, *shape) int8 on the GPU, contiguous."""
	if isinstance(states, torch.Tensor):
		t = states if states.dtype == torch.int8 else states.to(torch.int8)
		return t.to(gpu, non_blocking=True).contiguous()
	arr = np.ascontiguousarray(states, dtype=np.int8)
	return torch.from_numpy(arr).to(gpu)


def _to_dev_u8(x) -> torch.Tensor:
	if isinstance(x, torch.Tensor):
		return x.to(device=gpu, dtype=torch.uint8).contiguous()
	return torch.from_numpy(np.ascontiguousarray(x, dtype=np.uint8)).to(gpu)


def _actions_from(faces, dirs, n: int) -> torch.Tensor:
	"""Validated uint8 action indices 2*face + (1-dir) on the device (cube.py:33-34)."""
	if _is_dev(faces) or _is_dev(dirs):
		f = torch.as_tensor(faces, device=gpu).to(torch.int64)
		d = torch.as_tensor(dirs, device=gpu).to(torch.int64)
		a = 2 * f + (1 - d)
		return a.to(torch.uint8).contiguous()
	f = np.asarray(faces).astype(np.int64).ravel()
	d = np.asarray(dirs).astype(np.int64).ravel()
	if len(f) != n or len(d) != n:
		raise IndexError(f"need {n} faces and directions, got {len(f)} and {len(d)}")
	if n and (f.min() < 0 or f.max() > 5 or d.min() < 0 or d.max() > 1):
		raise IndexError("face must be in 0..5 and direction in 0..1")
	return torch.from_numpy((2 * f + (1 - d)).astype(np.uint8)).to(gpu)


_PINNED_FROM = 1 << 20      # device -> host results from 1 MiB on are staged through page-locked memory
#: Results up to this size are RETURNED as views of page-locked memory (torch's caching host allocator: a block goes back to its
#: pool when the array dies, never to the OS); larger ones -- and everything once `pinned_result_budget` bytes of such results are
#: alive -- are copied through one reusable page-locked staging buffer into an ordinary pageable array, so that a caller who
#: keeps many results (a training loop keeping children arrays) does not pin its RAM.  Set either to 0 to switch the path off.
pinned_result_max = 512 << 20
pinned_result_budget = 2 << 30
_pinned_alive = [0]
_staging = [None]


def _release_pinned(nbytes: int):
	_pinned_alive[0] -= nbytes


def _to_host(t: torch.Tensor) -> np.ndarray:
	"""
	Device tensor -> NumPy array.  Large results (the 240 MB of children of a 1 M-state fan-out) land in page-locked memory:
	a fresh pageable array costs its first-touch page faults on top of a staged copy -- 36-45 ms per 1 M parents against
	8-9 ms this way (benchmarks/kernels.py, host_path).  The page-locked path is bounded (see `pinned_result_max`,
	`pinned_result_budget`); a refused page-locked allocation falls back to the pageable copy.
	"""
	nbytes = t.numel() * t.element_size()
	if nbytes >= _PINNED_FROM and nbytes <= pinned_result_max and _pinned_alive[0] + nbytes <= pinned_result_budget:
		try:
			host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
		except RuntimeError:
			host = None
		if host is not None:
			host.copy_(t, non_blocking=True)
			torch.cuda.current_stream().synchronize()
			out = host.numpy()
			_pinned_alive[0] += nbytes
			import weakref
			weakref.finalize(host, _release_pinned, nbytes)              # the array keeps `host` alive; the budget follows its lifetime
			return out
	if nbytes >= _PINNED_FROM and pinned_result_max > 0:
		# over the budget: one reusable page-locked staging buffer (64 MiB), copied piecewise into a pageable array
		try:
			if _staging[0] is None:
				_staging[0] = torch.empty(64 << 20, dtype=torch.uint8, pin_memory=True)
			stage = _staging[0]
		except RuntimeError:
			stage = None
		if stage is not None:
			flat = t.contiguous().view(torch.uint8).reshape(-1)
			out = np.empty(nbytes, np.uint8)
			for at in range(0, nbytes, len(stage)):
				k = min(len(stage), nbytes - at)
				stage[:k].copy_(flat[at:at + k], non_blocking=True)
				torch.cuda.current_stream().synchronize()
				out[at:at + k] = stage[:k].numpy()
			return out.view(_NP_OF[t.dtype]).reshape(tuple(t.shape))
	return t.cpu().numpy()


_NP_OF = {torch.int8: np.int8, torch.uint8: np.uint8, torch.float32: np.float32, torch.float16: np.float16, torch.int64: np.int64,
          torch.int32: np.int32, torch.float64: np.float64, torch.bool: np.bool_}


def _new_stats() -> torch.Tensor:
	return torch.tensor([0, INT64_MAX], dtype=torch.int64, device=gpu)


def _check_dev(t: torch.Tensor, dtype: torch.dtype, what: str):
	"""The C ABI takes raw pointers: refuse anything that is not a dense device tensor of the expected element type."""
	if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()):
		raise ValueError(f"{what} must be a contiguous CUDA tensor of dtype {dtype}, got "
		                 f"{type(t).__name__}" + (f" {t.dtype} on {t.device}, contiguous={t.is_contiguous()}" if isinstance(t, torch.Tensor) else ""))


class device:
	"""
	Device-resident forms of the hot path: torch CUDA tensors in and out, no host copies, no synchronisation.
	`states` are int8 (n, 20) [or (n, 6, 8, 6)], contiguous.  These are thin wrappers over the C ABI.
	"""

	#: Device-pointer entries take action codes as they are: the kernels treat a code >= 12 as action 0 so that they
	#: never index past the move table, but the result is then meaningless (the host paths raise IndexError, as the
	#: reference's table indexing would).  Such a code leaves a sticky mark on the device: `device.bad_actions_seen()`
	#: reads and clears it (one synchronisation, whenever the caller likes).  Set `check_actions` to True to pay one
	#: reduction + sync per call and get the IndexError at once.
	check_actions = False

	@staticmethod
	def bad_actions_seen() -> bool:
		"""True if a device-pointer call since the last check was given an action code outside 0..11 (clears the mark)."""
		_ffi.require_gpu()
		seen = C.c_int(0)
		_ffi.check(_ffi.lib().rk_bad_actions_seen(C.byref(seen), _ffi.stream_ptr()))
		return bool(seen.value)

	@staticmethod
	def _validate(actions: torch.Tensor):
		if device.check_actions and actions.numel() and int(actions.max()) >= 12:
			raise IndexError(f"action code {int(actions.max())} outside 0..11")

	@staticmethod
	def multi_rotate(states: torch.Tensor, actions: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
		"""out[i] = move actions[i] (uint8 action index, 0..11) applied to states[i]   (cube.py:256-263)."""
		_ffi.require_gpu()
		_check_dev(states, torch.int8, "states")
		_check_dev(actions, torch.uint8, "actions")
		device._validate(actions)
		n = len(states)
		if len(actions) != n:
			raise ValueError(f"{n} states but {len(actions)} actions")
		if out is None:
			out = torch.empty_like(states)
		else:
			_check_dev(out, torch.int8, "out")
		_ffi.check(_ffi.lib().rk_multi_rotate(_repr_id(), states.data_ptr(), actions.data_ptr(), out.data_ptr(), n, _ffi.stream_ptr()))
		return out

	@staticmethod
	def multi_rotate_solved(states: torch.Tensor, actions: torch.Tensor, out: torch.Tensor = None, flags: torch.Tensor = None,
	                        stats: torch.Tensor = None):
		"""
		`multi_rotate` and `multi_is_solved` of the moved states in ONE launch (the pair of agents.py:157-159, :696-703 and
		train.py:277-281): returns (moved states, uint8 flags).  `stats` (int64[2] = [count, first index], initialise to
		[0, INT64_MAX]) is updated if given.  20-byte states: the moved states are not read back for the test.
		"""
		_ffi.require_gpu()
		_check_dev(states, torch.int8, "states")
		_check_dev(actions, torch.uint8, "actions")
		device._validate(actions)
		n = len(states)
		if len(actions) != n:
			raise ValueError(f"{n} states but {len(actions)} actions")
		if out is None:
			out = torch.empty_like(states)
		else:
			_check_dev(out, torch.int8, "out")
		if flags is None:
			flags = torch.empty(n, dtype=torch.uint8, device=states.device)
		else:
			_check_dev(flags, torch.uint8, "flags")
		if stats is not None:
			_check_dev(stats, torch.int64, "stats")
		_ffi.check(_ffi.lib().rk_multi_rotate_solved(_repr_id(), states.data_ptr(), actions.data_ptr(), out.data_ptr(), flags.data_ptr(),
		                                             stats.data_ptr() if stats is not None else None, n, _ffi.stream_ptr()))
		return out, flags

	@staticmethod
	def expand12(parents: torch.Tensor, children: torch.Tensor = None, solved: torch.Tensor = None,
	             stats: torch.Tensor = None, want_flags: bool = True):
		"""
		All 12 children of every parent, parent-major / action-minor, with the goal test fused
		(agents.py:277-281 + cube.py:88-89).  Returns (children int8 (12n, ...), solved uint8 (12n,) or None).
		`stats` (int64[2] = [count, first index], initialise to [0, INT64_MAX]) is updated if given.
		"""
		_ffi.require_gpu()
		_check_dev(parents, torch.int8, "parents")
		n = len(parents)
		if children is None:
			children = torch.empty((12 * n, *parents.shape[1:]), dtype=torch.int8, device=parents.device)
		else:
			_check_dev(children, torch.int8, "children")
			if children.numel() != 12 * parents.numel():
				raise ValueError("children must hold 12 states per parent")
		if solved is None and want_flags:
			solved = torch.empty(12 * n, dtype=torch.uint8, device=parents.device)
		elif solved is not None:
			_check_dev(solved, torch.uint8, "solved")
			if solved.numel() != 12 * n:
				raise ValueError("solved must hold 12 flags per parent")
		if stats is not None:
			_check_dev(stats, torch.int64, "stats")
		_ffi.check(_ffi.lib().rk_expand12(
			_repr_id(), parents.data_ptr(), children.data_ptr(),
			solved.data_ptr() if solved is not None else None,
			stats.data_ptr() if stats is not None else None, n, _ffi.stream_ptr()))
		return children, solved

	@staticmethod
	def to_soa(states: torch.Tensor) -> torch.Tensor:
		"""(n, 20) int8 rows -> int32 (5, n) planes (plane j = bytes 4j..4j+3 of every state)."""
		_ffi.require_gpu()
		n = len(states)
		planes = torch.empty((5, n), dtype=torch.int32, device=states.device)
		_ffi.check(_ffi.lib().rk_states_to_soa(states.data_ptr(), planes.data_ptr(), n, _ffi.stream_ptr()))
		return planes

	@staticmethod
	def from_soa(planes: torch.Tensor) -> torch.Tensor:
		"""int32 (..., 5, n) planes -> (n, 20) int8 rows (leading dimensions, e.g. the 12 actions, become rows blocks)."""
		_ffi.require_gpu()
		lead = planes.shape[:-2]
		n = planes.shape[-1]
		flat = planes.reshape(-1, 5, n)
		out = torch.empty((flat.shape[0], n, 20), dtype=torch.int8, device=planes.device)
		for i in range(flat.shape[0]):
			_ffi.check(_ffi.lib().rk_states_from_soa(flat[i].data_ptr(), out[i].data_ptr(), n, _ffi.stream_ptr()))
		return out.reshape(*lead, n, 20)

	@staticmethod
	def expand12_soa(parents: torch.Tensor, children: torch.Tensor = None, solved: torch.Tensor = None,
	                 stats: torch.Tensor = None, want_flags: bool = True):
		"""
		Structure-of-arrays fan-out: parents int32 (5, n) -> children int32 (12, 5, n) [action-major] and solved
		uint8 (12, n).  Same arithmetic as expand12; every access of a wavefront is one contiguous run.
		"""
		_ffi.require_gpu()
		n = parents.shape[1]
		if children is None:
			children = torch.empty((12, 5, n), dtype=torch.int32, device=parents.device)
		if solved is None and want_flags:
			solved = torch.empty((12, n), dtype=torch.uint8, device=parents.device)
		_ffi.check(_ffi.lib().rk_expand12_soa(
			parents.data_ptr(), children.data_ptr(), solved.data_ptr() if solved is not None else None,
			stats.data_ptr() if stats is not None else None, n, _ffi.stream_ptr()))
		return children, solved

	@staticmethod
	def multi_is_solved(states: torch.Tensor, flags: torch.Tensor = None, stats: torch.Tensor = None) -> torch.Tensor:
		"""uint8 (n,) flags, 1 where the state is solved (cube.py:88-89)."""
		_ffi.require_gpu()
		_check_dev(states, torch.int8, "states")
		n = len(states)
		if flags is None:
			flags = torch.empty(n, dtype=torch.uint8, device=states.device)
		else:
			_check_dev(flags, torch.uint8, "flags")
		if stats is not None:
			_check_dev(stats, torch.int64, "stats")
		_ffi.check(_ffi.lib().rk_multi_is_solved(
			_repr_id(), states.data_ptr(), flags.data_ptr(), stats.data_ptr() if stats is not None else None,
			n, _ffi.stream_ptr()))
		return flags

	@staticmethod
	def apply_sequences(actions: torch.Tensor, with_solved: bool, only_last: bool) -> torch.Tensor:
		"""actions uint8 (depth, games) -> states of every game along its move sequence (cube.py:218-232)."""
		_ffi.require_gpu()
		_check_dev(actions, torch.uint8, "actions")
		device._validate(actions)
		depth, games = actions.shape
		moves = depth - int(with_solved)
		rows = 1 if only_last else moves + int(with_solved)
		out = torch.empty((games * max(rows, 0), 20), dtype=torch.int8, device=gpu)
		_ffi.check(_ffi.lib().rk_apply_sequences(
			REPR_2024, actions.data_ptr(), depth, games, int(with_solved), int(only_last), out.data_ptr(), _ffi.stream_ptr()))
		return out

	@staticmethod
	def rollout_fanout(actions: torch.Tensor, with_solved: bool):
		"""actions uint8 (depth, games) -> (states (games*depth, 20), states_solved uint8, children (12*games*depth, 20), children_solved uint8):
		the walks of `apply_sequences`, the goal test of every state, their fan-out and its goal test in ONE launch (train.py:277-292)."""
		_ffi.require_gpu()
		_check_dev(actions, torch.uint8, "actions")
		device._validate(actions)
		depth, games = actions.shape
		n = games * depth
		states = torch.empty((n, 20), dtype=torch.int8, device=gpu)
		state_flags = torch.empty(n, dtype=torch.uint8, device=gpu)
		children = torch.empty((12 * n, 20), dtype=torch.int8, device=gpu)
		child_flags = torch.empty(12 * n, dtype=torch.uint8, device=gpu)
		_ffi.check(_ffi.lib().rk_rollout_fanout(REPR_2024, actions.data_ptr(), depth, games, int(with_solved), states.data_ptr(), state_flags.data_ptr(),
		                                        children.data_ptr(), child_flags.data_ptr(), None, _ffi.stream_ptr()))
		return states, state_flags, children, child_flags

	@staticmethod
	def as_oh(states: torch.Tensor, out: torch.Tensor = None, dtype: torch.dtype = torch.float32) -> torch.Tensor:
		"""One-hot (n, 480 | 288) of `dtype` (float32, float16 or bfloat16)   (cube.py:265-277, 363-369)."""
		_ffi.require_gpu()
		_check_dev(states, torch.int8, "states")
		n = len(states)
		code = {torch.float32: _ffi.OH_F32, torch.float16: _ffi.OH_F16, torch.bfloat16: _ffi.OH_BF16}[dtype]
		if out is None:
			out = torch.empty((n, get_oh_shape()), dtype=dtype, device=states.device)
		else:
			_check_dev(out, dtype, "out")
			if out.numel() < n * get_oh_shape():
				raise ValueError("one-hot output too small")
		_ffi.check(_ffi.lib().rk_as_oh(_repr_id(), states.data_ptr(), out.data_ptr(), code, n, _ffi.stream_ptr()))
		return out


################
# Rotate logic #   (cube.py:41-52)
################
def rotate(state: np.ndarray, face: int, direction: int) -> np.ndarray:
	"""One move on one state: face 0..5, direction 0 (negative) or 1 (positive).  Out of place."""
	face, direction = int(face), int(direction)
	if _is_dev(state):
		acts = torch.tensor([2 * face + (1 - direction)], dtype=torch.uint8, device=gpu)
		return device.multi_rotate(state.reshape(1, *shape()).contiguous(), acts)[0]
	if isinstance(state, torch.Tensor):
		return multi_rotate(np.asarray(state)[None], [face], [direction])[0]
	# one state from the host: straight to the library's host entry (what reference code that loops over `rotate` pays per call)
	if not (0 <= face <= 5 and 0 <= direction <= 1):
		raise IndexError("face must be in 0..5 and direction in 0..1")
	_ffi.require_gpu()
	src = np.ascontiguousarray(state, dtype=np.int8)
	if src.size != _row_bytes():
		raise ValueError(f"one state has shape {shape()}, got {src.shape}")
	out = np.empty_like(src)
	act = np.array([2 * face + (1 - direction)], dtype=np.uint8)
	_ffi.check(_ffi.lib().rk_multi_rotate_host(_repr_id(), src.ctypes.data, act.ctypes.data, out.ctypes.data, 1, _ffi.stream_ptr()))
	return out


#: Host arrays whose call moves up to this many bytes (inputs + outputs) go straight through the library's *_host entries, which pass
#: them zero-copy through a page-locked, device-mapped buffer (rk_api.hip, ZERO_COPY_MAX: one launch, one wait, no torch hop).
_ZERO_COPY_BYTES = 1 << 20


def _small(n_in: int, n_out: int, extra: int = 0) -> bool:
	"""n_in input states and n_out output states (plus `extra` bytes of actions / flags) fit the zero-copy path"""
	return n_in > 0 and (n_in + n_out) * _row_bytes() + extra + 1024 <= _ZERO_COPY_BYTES      # (1 KiB: the buffer's parts are aligned)


def _host_rows(states) -> np.ndarray:
	"""Host states as a C-contiguous int8 array of whole rows (n, *shape()): what the *_host entries read n * row bytes from.  A
	single state of shape (20,) has len 20 -- it must not be taken for 20 rows."""
	src = np.ascontiguousarray(states, dtype=np.int8)
	if src.ndim != 1 + len(shape()) or src.shape[1:] != shape():
		raise ValueError(f"states must have shape (n, {', '.join(map(str, shape()))}), got {src.shape}")
	return src


def _host_actions(faces, dirs, n: int) -> np.ndarray:
	f, d = np.asarray(faces).ravel(), np.asarray(dirs).ravel()
	if len(f) != n or len(d) != n:
		raise IndexError(f"need {n} faces and directions, got {len(f)} and {len(d)}")
	if f.dtype == np.uint8 and d.dtype == np.uint8:
		# what iter_actions() hands out: no negative values to look for, and the arithmetic stays in bytes (d <= 1 keeps 1 - d in range)
		if n and (f.max() > 5 or d.max() > 1):
			raise IndexError("face must be in 0..5 and direction in 0..1")
		return 2 * f + (1 - d)
	f, d = f.astype(np.int64, copy=False), d.astype(np.int64, copy=False)
	if n and (f.min() < 0 or f.max() > 5 or d.min() < 0 or d.max() > 1):
		raise IndexError("face must be in 0..5 and direction in 0..1")
	return 2 * f.astype(np.uint8) + (1 - d.astype(np.uint8))          # (validated: the arithmetic can stay in bytes)


def multi_rotate(states: np.ndarray, faces: np.ndarray, directions: np.ndarray) -> np.ndarray:
	"""Performs action (faces[i], directions[i]) on states[i]; returns a new array."""
	_ffi.require_gpu()
	n = len(states)
	if n == 0:
		return states.clone() if isinstance(states, torch.Tensor) else np.array(states, dtype=np.int8, copy=True)
	if _small(n, n, n) and not isinstance(states, torch.Tensor) and not _is_dev(faces) and not _is_dev(directions):
		src = _host_rows(states)
		acts = _host_actions(faces, directions, n)
		out = np.empty_like(src)
		_ffi.check(_ffi.lib().rk_multi_rotate_host(_repr_id(), src.ctypes.data, acts.ctypes.data, out.ctypes.data, n, _ffi.stream_ptr()))
		return out
	acts = _actions_from(faces, directions, n)
	out = device.multi_rotate(_to_dev_states(states), acts)
	return out if _is_dev(states) else _to_host(out)


def expand(states: np.ndarray, return_solved: bool = False):
	"""
	The fan-out idiom `multi_rotate(np.repeat(states, 12, 0), *iter_actions(len(states)))` as one call
	(agents.py:277-281, train.py:285): (12 n, ...) children, parent-major.  Not in the reference's surface; the
	search engines and the trainer's data generation call this instead of building 12 n action tuples.
	"""
	_ffi.require_gpu()
	dev_in = _is_dev(states)
	if not isinstance(states, torch.Tensor) and len(states) and _small(len(states), 12 * len(states), 12 * len(states)):
		# a few parents from the host (BFS, one-step agents): the library's host entry, no torch hop
		src = _host_rows(states)
		children = np.empty((12 * len(src), *src.shape[1:]), dtype=np.int8)
		flags = np.empty(12 * len(src), dtype=np.uint8) if return_solved else None
		_ffi.check(_ffi.lib().rk_expand12_host(_repr_id(), src.ctypes.data, children.ctypes.data, flags.ctypes.data if return_solved else None, None,
		                                       len(src), _ffi.stream_ptr()))
		return (children, flags.astype(bool)) if return_solved else children
	children, solved = device.expand12(_to_dev_states(states), want_flags=return_solved)
	if not dev_in:
		children = _to_host(children)
		solved = solved.cpu().numpy().astype(bool) if return_solved else None
	elif return_solved:
		solved = solved.bool()
	return (children, solved) if return_solved else children


#################
# Solving logic #   (cube.py:85-89)
#################
def is_solved(state: np.ndarray) -> bool:
	return bool(multi_is_solved(state.reshape(1, *shape()))[0])


def multi_is_solved(states: np.ndarray) -> np.ndarray:
	_ffi.require_gpu()
	if len(states) == 0:
		return np.zeros(0, dtype=bool)
	if _small(len(states), 0, len(states)) and not isinstance(states, torch.Tensor):
		src = _host_rows(states)
		flags = np.empty(len(src), dtype=np.uint8)
		_ffi.check(_ffi.lib().rk_multi_is_solved_host(_repr_id(), src.ctypes.data, flags.ctypes.data, None, len(src), _ffi.stream_ptr()))
		return flags.astype(bool)
	flags = device.multi_is_solved(_to_dev_states(states))
	return flags.bool() if _is_dev(states) else flags.cpu().numpy().astype(bool)


########################
# Representation logic #   (cube.py:130-173)
########################
def as_oh(states: np.ndarray) -> torch.Tensor:
	"""n states -> (n, 480) [or (n, 288)] float32 one-hot tensor on `librubiks_amd.gpu`; one state -> (1, ...)."""
	_ffi.require_gpu()
	if not isinstance(states, torch.Tensor):
		src = np.ascontiguousarray(states, dtype=np.int8)
		n = src.size // _row_bytes()
		if src.size == n * _row_bytes() and _small(n, 0):
			# states from the host: the library reads them zero-copy and writes the one-hot where the net will read it (no torch hop)
			out = torch.empty((n, get_oh_shape()), dtype=torch.float32, device=gpu)
			_ffi.check(_ffi.lib().rk_as_oh_host(_repr_id(), src.ctypes.data, out.data_ptr(), _ffi.OH_F32, n, _ffi.stream_ptr()))
			return out
	t = _to_dev_states(states)
	if t.dim() == len(shape()):
		t = t.unsqueeze(0)
	return device.as_oh(t)


def as_correct(t: torch.Tensor) -> torch.Tensor:
	"""6x8x6 only: (n, 6, 8) float tensor, +1 where a sticker shows its face's colour, -1 elsewhere (cube.py:371-380)."""
	assert not get_is2024(), "Correctness representation is only implemented for 20x24 representation"
	_ffi.require_gpu()
	n = len(t)
	s = t.reshape(n, 288).to(device=gpu, dtype=torch.int8).contiguous()
	out = torch.empty((n, 6, 8), dtype=torch.float32, device=gpu)
	_ffi.check(_ffi.lib().rk_as_correct686(s.data_ptr(), out.data_ptr(), n, _ffi.stream_ptr()))
	return out


def repeat_state(state: np.ndarray, n: int = action_dim) -> np.ndarray:
	"""n copies of `state` as an (n, *shape) array (cube.py:142-147)."""
	if isinstance(state, torch.Tensor):
		return state.unsqueeze(0).repeat(n, *[1] * state.dim())
	state = np.asarray(state)
	return np.tile(state, [n, *[1] * state.ndim])


################
# Action logic #   (cube.py:179-200)
################
def iter_actions(n: int = 1) -> np.ndarray:
	"""uint8 (2, 12 n): the 12 (face, direction) pairs tiled n times, for use with multi_rotate."""
	return np.tile(_ITER_ACTIONS_1, (1, n))


def indices_to_actions(indices: np.ndarray):
	faces = indices // 2
	dirs = 1 - indices % 2
	return faces, dirs


def rev_action(action: int) -> int:
	return action + 1 if action % 2 == 0 else action - 1


def rev_actions(actions: np.ndarray) -> np.ndarray:
	actions = np.asarray(actions)
	return actions + 1 - 2 * (actions % 2)


##################
# Scramble logic #   (cube.py:206-234)
##################
def _apply(actions_dg: np.ndarray, with_solved: bool, only_last: bool) -> np.ndarray:
	acts = np.ascontiguousarray(actions_dg, dtype=np.uint8)
	depth, games = acts.shape                   # with_solved: the first state of a game is the solved one, depth - 1 moves follow
	rows = 1 if only_last else depth
	if games * rows > 0 and games * rows * 20 + acts.size + 1024 <= _ZERO_COPY_BYTES and depth - int(with_solved) >= 0:
		# small walks (one scramble): the library's host entry, no torch hop
		if acts.size and int(acts.max()) >= 12:
			raise IndexError(f"action code {int(acts.max())} outside 0..11")
		out = np.empty((games * rows, 20), dtype=np.int8)
		_ffi.check(_ffi.lib().rk_apply_sequences_host(REPR_2024, acts.ctypes.data, depth, games, int(with_solved), int(only_last),
		                                              out.ctypes.data, _ffi.stream_ptr()))
		return out
	return _to_host(device.apply_sequences(torch.from_numpy(acts).to(gpu), with_solved, only_last))


def scramble(depth: int, force_not_solved=False):
	"""Random walk of `depth` moves from solved: returns (state, faces, dirs).  Same draws as the reference."""
	_ffi.require_gpu()
	while True:
		faces = np.random.randint(6, size=(depth,))
		dirs = np.random.randint(2, size=(depth,))
		if _is2024:
			state = _apply((2 * faces + (1 - dirs)).reshape(depth, 1), False, True)[0] if depth else get_solved()
		else:
			state = get_solved()
			for f, d in zip(faces, dirs):
				state = rotate(state, f, d)
		if not (force_not_solved and depth != 0 and is_solved(state)):
			return state, faces, dirs


def sequence_scrambler(games: int, depth: int, with_solved: bool):
	"""
	Out-of-place scrambler for ADI: the states along `games` random walks, game-major, and their one-hot
	encoding -- (games*depth, *shape) int8 array and (games*depth, oh) float tensor on `gpu`.
	"""
	_ffi.require_gpu()
	faces = np.random.randint(0, 6, (depth, games))
	dirs = np.random.randint(0, 2, (depth, games))
	if _is2024:
		acts = torch.from_numpy((2 * faces + (1 - dirs)).astype(np.uint8)).to(gpu)
		dev_states = device.apply_sequences(acts, bool(with_solved), False)
		return dev_states.cpu().numpy(), device.as_oh(dev_states)
	cur = repeat_state(get_solved_instance(), games)
	seq = [cur] if with_solved else []
	for d in range(depth - int(with_solved)):
		cur = multi_rotate(cur, faces[d], dirs[d])
		seq.append(cur)
	states = np.stack(seq, axis=1).reshape(games * len(seq), *shape())
	return states, as_oh(states)


#############
# Rendering #   (cube.py:149-173, 279-307, 382-388; maps.py:26-51) -- host-side formatting, no cube arithmetic
#############
_CORNER_POS, _EDGE_POS = get_633maps(F, B, T, D, L, R)
_RING_TO_33 = np.array([0, 3, 6, 7, 8, 5, 2, 1])
_RING_SHIFT = np.array([0, 6, 6, 4, 2, 4])


def as633(state: np.ndarray) -> np.ndarray:
	"""(6, 3, 3) colour picture of a state, faces in the order F, B, T, D, L, R."""
	if isinstance(state, torch.Tensor):
		state = state.cpu().numpy()
	pic = np.repeat(np.arange(6), 9).reshape(6, 3, 3)
	if _is2024:
		for i in range(8):
			slot, ori = divmod(int(state[i]), 3)
			if slot in (0, 2, 5, 7):
				ori = -ori
			for where, col in zip(_CORNER_POS[slot], np.roll([s[0] for s in _CORNER_POS[i]], ori)):
				pic[where] = col
		for i in range(12):
			slot, ori = divmod(int(state[8 + i]), 2)
			for where, col in zip(_EDGE_POS[slot], np.roll([s[0] for s in _EDGE_POS[i]], ori)):
				pic[where] = col
		return pic
	colours = np.argmax(state, axis=2)
	flat = pic.reshape(6, 9)
	for face in range(6):
		flat[face, _RING_TO_33] = np.roll(colours[face], -_RING_SHIFT[face])
	return flat.reshape(6, 3, 3)


def as69(state: np.ndarray) -> np.ndarray:
	return as633(state).reshape((6, 9))


def stringify(state: np.ndarray) -> str:
	pic = as633(state)
	grid = np.full((9, 12), " ", dtype="<U1")
	for face, (r, c) in {T: (0, 1), L: (1, 0), F: (1, 1), R: (1, 2), B: (1, 3), D: (2, 1)}.items():
		grid[3 * r:3 * r + 3, 3 * c:3 * c + 3] = pic[face].astype(str)
	return "\n".join(" ".join(row) for row in grid)
