"""fast-match_amd -- MI355X-native descriptor-matching hot path of arnfred/Fast-Match.

Python surface mirrors the reference modules (``fastmatch``, ``cache``, ``matchutil``);
the arithmetic runs in hand-written HIP kernels for gfx950 behind the C-ABI declared in
``include/fastmatch_hip.h`` (``libfastmatch_hip.so``, loaded with ctypes by ``_ffi``).

The directory name contains a hyphen, so import it through the root-level alias::

    import fastmatch_amd                      # == this package
    from fastmatch_amd import fastmatch, cache, matchutil
"""
from . import _ffi                                    # noqa: F401
from ._ffi import FastMatchHipError, Context, default_context   # noqa: F401

__all__ = ["_ffi", "FastMatchHipError", "Context", "default_context"]
