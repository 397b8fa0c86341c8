"""mecat_amd — MI355X-native mecat2pw hot path (index -> seed/DDF filter -> dw extension).

The product is libmecat_hip.so (HIP kernels behind the C ABI of include/mecat_hip.h) plus the C++ host driver
mecat_amd/bin/mecat2pw.  This Python package is a thin ctypes mirror of that ABI used by bench.py, the tests and
the multi-GPU launcher; it contains no compute and has NO CPU fallback: importing `mecat_amd.hip` raises if the
shared library is missing, and every call raises MhipError if no gfx950 device is usable.
"""
from .hip import (  # noqa: F401
    Candidate, AlnJob, AlnResult, Context, Index, MhipError, Params, Volume, lib, lib_path,
)

__all__ = ["Candidate", "AlnJob", "AlnResult", "Context", "Index", "MhipError", "Params", "Volume", "lib", "lib_path"]
