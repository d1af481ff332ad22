"""MI355X-native multi-frame super-resolution hot path.

Host-side mirror of the C-ABI in ``include/mfsr.h`` (HIP kernels for gfx950 in
``csrc/``).  Importing the package does not load the HIP library; the first
call into :mod:`multi_frame_super_resolution_amd.capi` does, and fails loudly
when it is missing (no CPU fallback).
"""
__version__ = "0.1.0"
