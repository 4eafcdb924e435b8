"""humid_amd -- MI355X-native neighbour-search-and-cluster hot path of HUMID.

The compute lives in libhumid_hip.so (hand-written HIP for gfx950, C ABI in
include/humid_hip.h).  This package is the Python host side: the ctypes binding, the mirror of
the reference's surface for this path, the multi-GPU orchestration and the synthetic inputs.
"""
from .api import (DIRECTIONAL, MAXIMUM, ClusterGraph, Context, Dedup, HumidError,  # noqa: F401
                  at_least_double)

__version__ = "0.1.0"
