import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
	sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
	config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
	import torch
	return torch.cuda.is_available()


def pytest_collection_modifyitems(config, items):
	if _has_gpu():
		return
	skip = pytest.mark.skip(reason="no HIP device in this container")
	for item in items:
		if "gpu" in item.keywords:
			item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
	"""All committed golden vectors (captured from the real reference by oracle/gen_golden.py)."""
	g = {}
	for name in ("cube_tables", "cube_kat", "astar_trace", "mcts_trace", "adi_trace"):
		with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
			g[name] = {k: z[k] for k in z.files}
	with open(os.path.join(GOLDEN, "cube_text.json")) as f:
		g["text"] = json.load(f)
	with open(os.path.join(GOLDEN, "cube_maps.json")) as f:      # oracle/gen_golden_maps.py
		g["maps"] = json.load(f)
	return g


@pytest.fixture(autouse=True)
def _default_repr():
	"""Every test starts and ends in the 20-byte representation."""
	from librubiks_amd import cube
	cube.set_is2024(True)
	yield
	cube.set_is2024(True)
