"""ky_amd -- MI355X-native path-tracing integrator behind ky's Scene/Camera/Integrator/Film API.

The product is ky_amd/lib/libkyhip.so (hand-written HIP kernels for gfx950 + the C ABI of include/kyhip.h)
and the C++ host layer ky_amd/host/ky.hpp.  This Python package is plumbing around them (ctypes, torch
device buffers, torch.distributed for the multi-GPU film gather).
"""
from . import _abi  # noqa: F401

__all__ = ["_abi", "api", "dist"]
