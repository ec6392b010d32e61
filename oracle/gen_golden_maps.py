"""
Golden-vector capture for `librubiks.cube.maps` -- TEST INFRASTRUCTURE.  Runs ONLY in the build container.

Imports the real reference from /root/reference and writes tests/golden/cube_maps.json: data only (names, integer tables), no
source text.  Kept apart from gen_golden.py so that the other fixtures need not be regenerated.

    PYTHONDONTWRITEBYTECODE=1 python3 oracle/gen_golden_maps.py

  public_names      the public names of `librubiks.cube` (what `from librubiks.cube import *` + attribute access give a caller;
                    cube/cube.py:22 re-exports the maps helpers), and of `librubiks.cube.maps`
  neighbors_686     maps.py:149-156
  maps633           get_633maps(0..5) (maps.py:26-51): sticker positions of the 8 corner and 12 side cubies
  simple_state      SimpleState's four arrays (maps.py:54-60) and its str()
  tensor_map_*      get_tensor_map for int8 and int64 (dtype and values); corner/side positions for every (pos, orientation)
"""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

from librubiks import cube  # noqa: E402
from librubiks.cube import maps  # noqa: E402


def public(mod):
	return sorted(n for n, v in vars(mod).items() if not n.startswith("_") and not isinstance(v, types.ModuleType))


def main():
	c, s = maps.get_633maps(0, 1, 2, 3, 4, 5)
	st = maps.SimpleState()
	rec = {
		"public_names_cube": public(cube),
		"public_names_maps": public(maps),
		"neighbors_686": maps.neighbors_686.tolist(),
		"neighbors_686_dtype": str(maps.neighbors_686.dtype),
		"maps633_corners": [[list(map(int, x)) for x in cubie] for cubie in c],
		"maps633_sides": [[list(map(int, x)) for x in cubie] for cubie in s],
		"maps633_swapped": [[[list(map(int, x)) for x in cubie] for cubie in part] for part in maps.get_633maps(5, 4, 3, 2, 1, 0)],
		"simple_state": {k: getattr(st, k).tolist() for k in ("corners", "corner_orientations", "sides", "side_orientations")},
		"simple_state_str": str(st),
		"tensor_map_int8": maps.get_tensor_map(np.int8).tolist(),
		"tensor_map_int8_dtype": str(maps.get_tensor_map(np.int8).dtype),
		"tensor_map_int64_dtype": str(maps.get_tensor_map(np.int64).dtype),
		"corner_pos": [[int(maps.get_corner_pos(p, o)) for o in range(3)] for p in range(8)],
		"side_pos": [[int(maps.get_side_pos(p, o)) for o in range(2)] for p in range(12)],
		"action_maps": {n: {"corner_map": list(getattr(maps.Actions, n).corner_map), "side_map": list(getattr(maps.Actions, n).side_map),
		                    "corner_static": int(getattr(maps.Actions, n).corner_static), "side_switch": bool(getattr(maps.Actions, n).side_switch)}
		                for n in "FBTDLR"},
	}
	with open(os.path.join(OUT, "cube_maps.json"), "w") as f:
		json.dump(rec, f, indent=1, sort_keys=True)
	print("wrote cube_maps.json:", len(rec["public_names_cube"]), "public names of librubiks.cube")


if __name__ == "__main__":
	main()
