"""MI355X-native TSDF volumetric fusion behind the semantic-slam `tsdf.hpp` API.

The product is the C-ABI HIP library `libtsdf_hip.so` (include/tsdf_hip.h, sources under
csrc/) plus the C++ drop-in classes in include/tsdf.hpp / include/TSDFfusion.hpp.  This
Python package is the thin host-side mirror used by tests and bench.py:

    capi     ctypes binding of the C ABI (no fallback: raises if the library is missing)
    ingest   keyframe poses / associations saved by the SLAM front-end
    sharded  z-slab sharding of one grid over ranks / devices
    synth    synthetic depth + pose workloads (numpy)
"""
from . import capi, synth  # noqa: F401
from .capi import TsdfError, Volume, make_config  # noqa: F401
