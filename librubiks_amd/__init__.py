"""
librubiks_amd -- MI355X-native drop-in for the cube hot path of peleiden/librubiks.

    from librubiks_amd import cube            # same surface as `from librubiks import cube`
    from librubiks_amd.solving import agents  # AStar / MCTS whose expand loops run on the GPU

All cube arithmetic runs in hand-written HIP kernels (librubiks_hip.so, C ABI in include/rubiks_hip.h).
There is no CPU fallback: without the library or without a gfx950 device the compute entry points raise.

The names `cpu`, `gpu`, `no_grad` and `reset_cuda` exist because callers of the reference import them from its
package root (librubiks/__init__.py:5-22).
"""
import torch

__version__ = "0.1.0"

_HAS_DEVICE = torch.cuda.is_available()

#: where one-hot batches and the nets live; the host device when no GPU is visible (then only host helpers work)
gpu = torch.device("cuda") if _HAS_DEVICE else torch.device("cpu")
cpu = torch.device("cpu")


def no_grad(fun):
	"""Decorator: run `fun` without autograd bookkeeping."""
	return torch.no_grad()(fun)


def reset_cuda():
	"""Release cached device memory and wait for the device to go idle."""
	if _HAS_DEVICE:
		torch.cuda.empty_cache()
		torch.cuda.synchronize()
