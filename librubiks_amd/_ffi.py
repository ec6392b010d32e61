"""
ctypes binding of librubiks_hip.so (C ABI: include/rubiks_hip.h).

The library is the product; this file only declares its entry points.  If the shared object is missing
the import of any compute module fails loudly -- there is deliberately no Python/NumPy fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (imported first so that librubiks_hip.so binds to the HIP runtime torch already loaded)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librubiks_hip.so")

REPR_2024, REPR_686 = 0, 1
OH_F32, OH_F16, OH_BF16, OH_STATES = 0, 1, 2, 3
OHL_GATHER, OHL_MFMA, OHL_MFMA_DIRECT, OHL_MFMA_TILED = 0, 1, 2, 3
OHL_ACT_NONE, OHL_ACT_ELU, OHL_ACT_RELU = 0, 1, 2
INT64_MAX = (1 << 63) - 1

_vp, _sz, _i = C.c_void_p, C.c_size_t, C.c_int

#: name -> (restype, argtypes); kept in the order of include/rubiks_hip.h.  tests/test_abi.py checks that the
#: header, this table and the built library agree symbol for symbol.
SIGNATURES = {
	"rk_version": (_i, []),
	"rk_last_error": (C.c_char_p, []),
	"rk_init": (_i, [_i]),
	"rk_set_pacing": (_i, [_i]),
	"rk_calibrate_pacing": (_i, [_i]),
	"rk_get_pacing": (_i, [C.POINTER(C.c_uint), C.POINTER(C.c_int), C.POINTER(C.c_float)]),
	"rk_pace_slot_of_device": (_i, [_i]),
	"rk_stream_register": (_i, [_vp]),
	"rk_stream_forget": (_i, [_vp]),
	"rk_tables": (_i, [_i, _vp]),
	"rk_face_definitions": (_i, [_vp]),
	"rk_solved": (_i, [_i, _vp]),
	"rk_malloc": (_i, [C.POINTER(_vp), _sz]),
	"rk_free": (_i, [_vp]),
	"rk_memcpy_h2d": (_i, [_vp, _vp, _sz, _vp]),
	"rk_memcpy_d2h": (_i, [_vp, _vp, _sz, _vp]),
	"rk_memset": (_i, [_vp, _i, _sz, _vp]),
	"rk_stream_synchronize": (_i, [_vp]),
	"rk_multi_rotate": (_i, [_i, _vp, _vp, _vp, _sz, _vp]),
	"rk_multi_rotate_solved": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
	"rk_bad_actions_seen": (_i, [C.POINTER(C.c_int), _vp]),
	"rk_multi_rotate_fd": (_i, [_i, _vp, _vp, _vp, _vp, _sz, _vp]),
	"rk_expand12": (_i, [_i, _vp, _vp, _vp, _vp, _sz, _vp]),
	"rk_expand12_soa": (_i, [_vp, _vp, _vp, _vp, _sz, _vp]),
	"rk_states_to_soa": (_i, [_vp, _vp, _sz, _vp]),
	"rk_states_from_soa": (_i, [_vp, _vp, _sz, _vp]),
	"rk_multi_is_solved": (_i, [_i, _vp, _vp, _vp, _sz, _vp]),
	"rk_apply_sequences": (_i, [_i, _vp, _i, _i, _i, _i, _vp, _vp]),
	"rk_rollout_fanout": (_i, [_i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
	"rk_as_oh": (_i, [_i, _vp, _vp, _i, _sz, _vp]),
	"rk_ohl_create": (_i, [C.POINTER(_vp), _vp, _i, _vp, _i, _vp]),
	"rk_ohl_destroy": (_i, [_vp]),
	"rk_ohl_forward": (_i, [_vp, _vp, _vp, _i, _sz, _i, _vp]),
	"rk_ohl_set_epilogue": (_i, [_vp, _i, C.c_float, _vp, _vp, _vp]),
	"rk_tail_linear": (_i, [_vp, _sz, _i, _sz, _vp, _vp, _i, _i, C.c_float, _vp, _vp]),
	"rk_as_correct686": (_i, [_vp, _vp, _sz, _vp]),
	"rk_astar_create": (_i, [C.POINTER(_vp), _sz, _i]),
	"rk_astar_destroy": (_i, [_vp]),
	"rk_astar_reset": (_i, [_vp, _vp, C.c_double, _vp]),
	"rk_astar_set_budget": (_i, [_vp, C.c_longlong, _vp]),
	"rk_astar_grow": (_i, [_vp, _sz, _vp]),
	"rk_astar_step_expand": (_i, [_vp, _vp, _i, _vp]),
	"rk_astar_step_commit": (_i, [_vp, _vp, _vp]),
	"rk_astar_set_values_dtype": (_i, [_vp, _i]),
	"rk_astar_status": (_i, [_vp, _vp, _vp]),
	"rk_astar_expand": (_i, [_vp, _i, _vp, _vp]),
	"rk_astar_new_states_oh": (_i, [_vp, _vp, _i, _vp]),
	"rk_astar_commit": (_i, [_vp, _vp, _vp]),
	"rk_astar_size": (C.c_longlong, [_vp]),
	"rk_astar_open_size": (C.c_longlong, [_vp]),
	"rk_astar_export": (_i, [_vp, _sz, _sz, _vp, _vp, _vp, _vp, _vp]),
	"rk_astar_path": (C.c_longlong, [_vp, C.c_longlong, _vp, _sz, _vp]),
	"rk_astar_lookup": (C.c_longlong, [_vp, _vp, _vp]),
	"rk_astar_export_open": (C.c_longlong, [_vp, _vp, _vp, _sz, _vp]),
	"rk_astar_next_pops": (C.c_longlong, [_vp, _vp, _sz, _vp]),
	"rk_astarb_create": (_i, [C.POINTER(_vp), _i, _sz, _i]),
	"rk_astarb_destroy": (_i, [_vp]),
	"rk_astarb_reset": (_i, [_vp, _vp, _vp, C.c_double, _vp]),
	"rk_astarb_set_values_dtype": (_i, [_vp, _i, _vp]),
	"rk_astarb_step_expand": (_i, [_vp, _vp, _i, _vp]),
	"rk_astarb_step_expand_compact": (_i, [_vp, _vp, _i, _vp, _vp]),
	"rk_astarb_step_commit": (_i, [_vp, _vp, _vp]),
	"rk_astarb_status": (_i, [_vp, _vp, _vp]),
	"rk_astarb_export": (_i, [_vp, _i, _sz, _sz, _vp, _vp, _vp, _vp, _vp]),
	"rk_astarb_path": (C.c_longlong, [_vp, _i, C.c_longlong, _vp, _sz, _vp]),
	"rk_astar_create_sharded": (_i, [C.POINTER(_vp), _sz, _i, _i, _i]),
	"rk_shard_owner": (_i, [_vp, _i]),
	"rk_astar_shard_block_bytes": (C.c_longlong, [_vp]),
	"rk_astar_shard_gather_len": (C.c_longlong, [_vp]),
	"rk_astar_shard_gather_ptr": (_vp, [_vp]),
	"rk_astar_shard_bind": (_i, [_vp, _vp]),
	"rk_astar_shard_reset": (_i, [_vp, _vp, C.c_double, _vp, _vp]),
	"rk_astar_shard_select": (_i, [_vp, _vp, C.c_double, C.c_double, _vp, _vp]),
	"rk_astar_shard_decision": (_i, [_vp, _vp, _vp]),
	"rk_astar_shard_insert": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
	"rk_astar_shard_new_count": (_i, [_vp, _vp, _vp]),
	"rk_astar_shard_push": (_i, [_vp, _vp, _vp, _vp, _vp]),
	"rk_astar_shard_push_rows": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
	"rk_astar_shard_flush": (_i, [_vp, _vp, _vp]),
	"rk_astar_shard_clear_send": (_i, [_vp, _vp, _i, _i, _vp]),
	"rk_astar_shard_parent": (_i, [_vp, C.c_longlong, _vp, _vp]),
	"rk_astar_shard_export_ranks": (_i, [_vp, _sz, _sz, _vp, _vp]),
	"rk_comm_unique_id": (_i, [_vp]),
	"rk_comm_create": (_i, [C.POINTER(_vp), _vp, _i, _i]),
	"rk_comm_destroy": (_i, [_vp]),
	"rk_comm_rank": (_i, [_vp]),
	"rk_comm_world": (_i, [_vp]),
	"rk_comm_all_gather": (_i, [_vp, _vp, _vp, _sz, _vp]),
	"rk_comm_all_to_all": (_i, [_vp, _vp, _vp, _sz, _vp]),
	"rk_comm_broadcast": (_i, [_vp, _vp, _sz, _i, _vp]),
	"rk_mcts_create": (_i, [C.POINTER(_vp), _i, _sz, _sz]),
	"rk_mcts_destroy": (_i, [_vp]),
	"rk_mcts_reset": (_i, [_vp, _vp, _vp, C.c_double, C.c_double, _vp]),
	"rk_mcts_roots_oh": (_i, [_vp, _vp, _i, _vp]),
	"rk_mcts_set_root_pv": (_i, [_vp, _vp, _vp, _vp]),
	"rk_mcts_expand": (_i, [_vp, _vp]),
	"rk_mcts_set_expand_ahead": (_i, [_vp, C.c_longlong]),
	"rk_mcts_children_oh": (_i, [_vp, _vp, _i, _vp]),
	"rk_mcts_backup_select": (_i, [_vp, _vp, _vp, _vp]),
	"rk_mcts_backup_select_logits": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp]),
	"rk_mcts_backup_select_logits_range": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _i, _vp]),
	"rk_mcts_children": (_vp, [_vp]),
	"rk_mcts_status": (_i, [_vp, _vp, _vp]),
	"rk_mcts_export": (_i, [_vp, _i, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
	"rk_mcts_path": (C.c_longlong, [_vp, _i, _vp, _vp, _sz, _vp]),
	"rk_mcts_grow": (_i, [_vp, _sz, _sz, _vp, _vp]),
	"rk_mcts_search_graph": (_i, [_vp, _vp]),
	"rk_mcts_graph_path": (C.c_longlong, [_vp, _i, _vp, _sz, _vp]),
	"rk_multi_rotate_host": (_i, [_i, _vp, _vp, _vp, _sz, _vp]),
	"rk_expand12_host": (_i, [_i, _vp, _vp, _vp, _vp, _sz, _vp]),
	"rk_multi_is_solved_host": (_i, [_i, _vp, _vp, _vp, _sz, _vp]),
	"rk_apply_sequences_host": (_i, [_i, _vp, _i, _i, _i, _i, _vp, _vp]),
	"rk_as_oh_host": (_i, [_i, _vp, _vp, _i, _sz, _vp]),
}

_lib = None


class RubiksHipError(RuntimeError):
	pass


def lib():
	"""The loaded library.  Raises ImportError when it has not been built (python -m librubiks_amd.build)."""
	global _lib
	if _lib is None:
		if not os.path.exists(LIB_PATH):
			raise ImportError(
				f"{LIB_PATH} is missing: build it with `python -m librubiks_amd.build` (hipcc, gfx950). "
				"librubiks_amd has no CPU fallback.")
		handle = C.CDLL(LIB_PATH)
		for name, (res, args) in SIGNATURES.items():
			fn = getattr(handle, name)
			fn.restype, fn.argtypes = res, args
		_lib = handle
	return _lib


def check(rc: int):
	if rc != 0:
		raise RubiksHipError(f"librubiks_hip error {rc}: {lib().rk_last_error().decode()}")


_gpu_seen = False


def require_gpu():
	global _gpu_seen
	if not _gpu_seen:                      # asked once: a device does not disappear, and the question costs a microsecond per launch
		if not torch.cuda.is_available():
			raise RubiksHipError("no HIP device visible: librubiks_amd computes only on a gfx950 GPU (no CPU fallback)")
		_gpu_seen = True


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


_registered_streams = {0}          # the null stream needs no registration


def stream_ptr() -> int:
	"""hipStream_t of torch's current stream, so that kernels order with the caller's torch work.  (The raw accessor skips
	building a torch.cuda.Stream object: a launch through this shim costs the host about 8 us -- profiles/r03_kernels.json,
	"LAUNCH FLOOR" -- and a third of that was this call.)  A stream seen for the first time is registered with the library
	(rk_stream_register): torch's streams come from a pool and are never destroyed, which is what registration promises.  The owner
	of a `torch.cuda.ExternalStream` calls `forget_stream` before destroying it."""
	s = _raw_stream(torch.cuda.current_device()) if _raw_stream is not None else torch.cuda.current_stream().cuda_stream
	if s not in _registered_streams:
		lib().rk_stream_register(s)
		_registered_streams.add(s)
	return s


def forget_stream(stream) -> None:
	"""Before a stream that was current during a call into this package is destroyed (only external streams ever are)."""
	s = int(getattr(stream, "cuda_stream", stream) or 0)
	if s:
		lib().rk_stream_forget(s)
		_registered_streams.discard(s)
