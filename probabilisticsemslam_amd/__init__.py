"""MI355X-native k-best assignment engine for the probabilistic data-association
hot path of EladMichael/probabilisticSemSlam (shortestPathCPP.cpp / assignment.cpp).

The product is the C-ABI library ``libkbest_amd.so`` (HIP kernels for gfx950 +
``include/kbest_c.h``).  This package is the thin Python driver used by the
tests and by ``bench.py``: ctypes bindings that mirror the reference's call
surface (``kBest2D``, ``kBest2DCutoff``, ``assignmentProb``) plus a device-
pointer entry for buffers that already live in HBM (torch is only the
allocator / stream / process-group plumbing).
"""
from .engine import (KBestEngine, KBestError, KBestMulti, assignmentProb, kBest2D, kBest2DCutoff, lib_path, load_library)  # noqa: F401
