"""bwt_amd -- MI355X (gfx950) Burrows-Wheeler block codec.

The product is ``libbwtgpu.so`` (hand-written HIP kernels behind the C ABI of
``include/bwtgpu.h``) and the C++ host layer in ``bwt_amd/host`` that keeps the
reference's ``bwt.hpp`` / ``inverse_bwt.hpp`` / ``FileProcessor`` interface.
This Python package is a thin ctypes view of the same C ABI (``bwt_amd.api``)
plus the synthetic-input definitions (``bwt_amd.synth``).
"""
from . import synth  # noqa: F401  (pure numpy, importable without a GPU)

__all__ = ["api", "synth", "build"]


def __getattr__(name):
    if name in ("api", "build"):
        import importlib

        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
