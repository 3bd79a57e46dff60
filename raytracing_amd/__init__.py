"""raytracing_amd -- MI355X (gfx950) back end of the raytracer's per-pixel render hot path.

The product is libmrt.so (hand-written HIP kernels behind the C-ABI of include/mrt.h) plus
the C++ host layer under raytracing_amd/host.  This Python package is the harness-side
binding (ctypes) used by the tests and bench.py.  There is no CPU fallback: importing
`raytracing_amd.capi` fails loudly if the built libraries are missing.
"""
from . import capi, scenes  # noqa: F401
from .capi import Context, MrtError  # noqa: F401
