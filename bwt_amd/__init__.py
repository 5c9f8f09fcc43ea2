"""MI355X-native SA/BWT build and FM-index search (hot paths of yichizhng/bwt).

The product is bwt_amd/libbwt_mi355x.so (HIP kernels for gfx950 behind a
C-ABI, see include/); this package is the Python plumbing used by the tests
and bench.py.  Importing it does not import torch; ``bwt_amd.gpu`` does.
"""
from . import _lib, synth  # noqa: F401

__all__ = ["_lib", "synth"]
