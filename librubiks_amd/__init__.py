"""
librubiks_amd -- MI355X-native drop-in for the cube hot path of peleiden/librubiks.

    from librubiks_amd import cube            # same surface as `from librubiks import cube`
    from librubiks_amd.solving import agents  # AStar / MCTS whose expand loops run on the GPU

All cube arithmetic runs in hand-written HIP kernels (librubiks_hip.so, C ABI in include/rubiks_hip.h).
There is no CPU fallback: without the library or without a gfx950 device the compute entry points raise.
Mirrors librubiks/__init__.py:5-22 for the `cpu` / `gpu` / `no_grad` / `reset_cuda` names.
"""
import functools

import torch

cpu = torch.device("cpu")
gpu = torch.device("cuda" if torch.cuda.is_available() else "cpu")

__version__ = "0.1.0"


def reset_cuda():
	torch.cuda.empty_cache()
	if torch.cuda.is_available():
		torch.cuda.synchronize()


def no_grad(fun):
	@functools.wraps(fun)
	def wrapper(*args, **kwargs):
		with torch.no_grad():
			return fun(*args, **kwargs)
	return wrapper
