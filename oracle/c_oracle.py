"""
ctypes front end of oracle/cube_oracle.c -- TEST INFRASTRUCTURE (see cube_oracle.py header).

`build()` compiles the C restatement with gcc into oracle/_build/liboracle.so (git-ignored, but it
travels to the GPU box with the gpurun snapshot).  `lib()` loads it, building on demand.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "cube_oracle.c")
_OUT = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
	if not force and os.path.exists(_OUT) and os.path.getmtime(_OUT) >= os.path.getmtime(_SRC):
		return _OUT
	os.makedirs(os.path.dirname(_OUT), exist_ok=True)
	cmd = ["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-shared", "-fPIC", _SRC, "-o", _OUT]
	subprocess.run(cmd, check=True)
	return _OUT


def lib():
	global _lib
	if _lib is None:
		_lib = C.CDLL(build())
		_lib.orc_multi_is_solved.restype = C.c_longlong
		_lib.orc_max_threads.restype = C.c_int
	return _lib


def _p(a: np.ndarray):
	return a.ctypes.data_as(C.c_void_p)


def tables():
	lut = np.empty((12, 2, 24), np.uint8)
	perm = np.empty((12, 48), np.uint8)
	solved = np.empty(20, np.int8)
	lib().orc_tables(_p(lut), _p(perm), _p(solved))
	return lut, perm, solved


def max_threads() -> int:
	return int(lib().orc_max_threads())


def multi_rotate(states: np.ndarray, actions: np.ndarray, threads: int = 1) -> np.ndarray:
	states = np.ascontiguousarray(states, np.int8)
	actions = np.ascontiguousarray(actions, np.uint8)
	out = np.empty_like(states)
	lib().orc_multi_rotate(_p(states), _p(actions), _p(out), C.c_size_t(len(states)), C.c_int(threads))
	return out


def expand12(states: np.ndarray, threads: int = 1, out=None, solved=None):
	states = np.ascontiguousarray(states, np.int8)
	n = len(states)
	if out is None:
		out = np.empty((12 * n, 20), np.int8)
	if solved is None:
		solved = np.empty(12 * n, np.uint8)
	lib().orc_expand12(_p(states), _p(out), _p(solved), C.c_size_t(n), C.c_int(threads))
	return out, solved


def multi_is_solved(states: np.ndarray):
	states = np.ascontiguousarray(states, np.int8)
	flags = np.empty(len(states), np.uint8)
	first = C.c_longlong(-1)
	cnt = lib().orc_multi_is_solved(_p(states), _p(flags), C.c_size_t(len(states)), C.byref(first))
	return flags.astype(bool), int(cnt), int(first.value)


def as_oh(states: np.ndarray) -> np.ndarray:
	states = np.ascontiguousarray(np.atleast_2d(states), np.int8)
	oh = np.empty((len(states), 480), np.float32)
	lib().orc_as_oh(_p(states), _p(oh), C.c_size_t(len(states)))
	return oh


def multi_rotate686(states: np.ndarray, actions: np.ndarray) -> np.ndarray:
	states = np.ascontiguousarray(states, np.int8)
	actions = np.ascontiguousarray(actions, np.uint8)
	out = np.empty_like(states)
	lib().orc_multi_rotate686(_p(states), _p(actions), _p(out), C.c_size_t(len(states)))
	return out


def digest(rows: np.ndarray):
	"""(order-independent sum, order-dependent chain) 64-bit FNV digests of a (n, 20) int8 array."""
	rows = np.ascontiguousarray(rows).view(np.int8).reshape(-1, 20)
	s, c = C.c_uint64(0), C.c_uint64(0)
	lib().orc_digest(_p(rows), C.c_size_t(len(rows)), C.byref(s), C.byref(c))
	return int(s.value), int(c.value)


if __name__ == "__main__":
	print(build(force=True))
