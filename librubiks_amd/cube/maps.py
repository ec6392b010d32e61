"""
Drop-in for `librubiks.cube.maps` (reference: librubiks/cube/maps.py): the readable face definitions and the tables derived
from them, under the reference's names.  Nothing here is computed a second time in Python: the face rings, the neighbour table
and the move table come out of librubiks_hip.so (`rk_face_definitions`, `rk_tables`; csrc/rk_tables.h generates them at
compile time), so what a caller reads here IS what the kernels use.

	SimpleState                      solved cubie positions / orientations (maps.py:54-65)
	ActionMap, Actions               the six face definitions (maps.py:67-98)
	get_corner_pos, get_side_pos     code of a cubie = slot * 3 + orientation / slot * 2 + orientation (maps.py:101-105)
	get_tensor_map(dtype)            (2, 6, 2, 24) delta table, [0] negative and [1] positive direction (maps.py:107-145)
	get_633maps(F, B, T, D, L, R)    sticker coordinates of the 8 corners and 12 sides in the 6x3x3 picture (maps.py:26-51)
	neighbors_686                    the four neighbours of every face in positive direction (maps.py:149-156)
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from librubiks_amd import _ffi


def get_633maps(F, B, T, D, L, R):
	"""Where the stickers of every cubie sit in the 6x3x3 picture: ((face, row, column), ...) per cubie, the tracked sticker
	first, then the others in the order of a right turn (maps.py:13-51).  Pure rendering geometry: as633 / stringify use it."""
	corner_633map = (
		((F, 0, 0), (L, 0, 2), (T, 2, 0)), ((F, 2, 0), (D, 0, 0), (L, 2, 2)),
		((F, 2, 2), (R, 2, 0), (D, 0, 2)), ((F, 0, 2), (T, 2, 2), (R, 0, 0)),
		((B, 0, 2), (T, 0, 0), (L, 0, 0)), ((B, 2, 2), (L, 2, 0), (D, 2, 0)),
		((B, 2, 0), (D, 2, 2), (R, 2, 2)), ((B, 0, 0), (R, 0, 2), (T, 0, 2)),
	)
	side_633map = (
		((F, 0, 1), (T, 2, 1)), ((F, 1, 0), (L, 1, 2)), ((F, 2, 1), (D, 0, 1)), ((F, 1, 2), (R, 1, 0)),
		((T, 1, 0), (L, 0, 1)), ((D, 1, 0), (L, 2, 1)), ((D, 1, 2), (R, 2, 1)), ((T, 1, 2), (R, 0, 1)),
		((B, 0, 1), (T, 0, 1)), ((B, 1, 2), (L, 1, 0)), ((B, 2, 1), (D, 2, 1)), ((B, 1, 0), (R, 1, 2)),
	)
	return corner_633map, side_633map


class SimpleState:
	"""The solved state in readable form: cubie i sits in slot i with orientation 0 (maps.py:54-65)."""
	corners = np.arange(8)
	corner_orientations = np.zeros(8, dtype=int)
	sides = np.arange(12)
	side_orientations = np.zeros(12, dtype=int)

	def __str__(self):
		rows = (("Corners:", self.corners), ("Corner orientations:", self.corner_orientations),
		        ("Sides:", self.sides), ("Side orientations:", self.side_orientations))
		return "\n".join(f"{name:<21}{[int(x) for x in vals]}" for name, vals in rows)


@dataclass
class ActionMap:
	corner_map: tuple      # ring of corner slots in positive revolution, closed (first slot repeated at the end)
	side_map: tuple        # ring of side slots, closed
	corner_static: int     # the corner orientation that stays; the other two swap
	side_switch: bool      # whether side orientations flip


def _face_definitions() -> np.ndarray:
	out = np.empty((6, 14), np.uint8)
	_ffi.check(_ffi.lib().rk_face_definitions(out.ctypes.data))
	return out


_FACES = _face_definitions()


class Actions:
	pass


for _name, _row in zip("FBTDLR", _FACES):
	setattr(Actions, _name, ActionMap(tuple(int(x) for x in _row[:4]) + (int(_row[0]),), tuple(int(x) for x in _row[4:8]) + (int(_row[4]),),
	                                  int(_row[8]), bool(_row[9])))
del _name, _row

neighbors_686 = _FACES[:, 10:14].astype(np.int64)


def get_corner_pos(pos: int, orientation: int):
	return pos * 3 + orientation


def get_side_pos(pos: int, orientation: int):
	return pos * 2 + orientation


def get_tensor_map(dtype):
	"""maps[dir][face][kind][v] = new code - v (dir 0 negative, 1 positive; kind 0 corners, 1 sides), from the library's absolute
	table T[a][kind][v] with a = 2 face + (1 - dir)."""
	lut = np.empty((12, 2, 24), np.uint8)
	_ffi.check(_ffi.lib().rk_tables(_ffi.REPR_2024, lut.ctypes.data))
	delta = lut.astype(np.int64) - np.arange(24)
	# action 2 face is the positive turn, 2 face + 1 the negative one
	return np.stack([delta[1::2], delta[0::2]]).astype(dtype)
