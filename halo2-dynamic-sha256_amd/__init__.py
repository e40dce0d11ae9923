"""MI355X-native SHA-256 witness engine for the halo2 dynamic-SHA256 gadget.

Product code: the C ABI (include/hsw.h -> libhsw.so, hand-written gfx950 HIP)
plus thin host plumbing.  Nothing here imports oracle/.

The directory name carries a hyphen (it mirrors the reference repo's name), so
import it with importlib:

    import importlib
    hsw = importlib.import_module("halo2-dynamic-sha256_amd")
"""
from . import _native
from ._native import (HSW_OK, HSW_REPR_CANONICAL, HSW_REPR_COMPACT64, HSW_REPR_MONTGOMERY, HSW_SKIP_CHIP,
                      HSW_SKIP_GATE, HswError, Shape, build, shape_query)
from ._native import digest_prepare
from .engine import WitnessEngine
from .gadget import AssignedHashResult, Sha256DynamicConfig

__all__ = ["WitnessEngine", "Sha256DynamicConfig", "AssignedHashResult", "digest_prepare", "HswError", "Shape", "shape_query", "build", "_native",
           "HSW_OK", "HSW_REPR_CANONICAL", "HSW_REPR_MONTGOMERY", "HSW_REPR_COMPACT64", "HSW_SKIP_GATE", "HSW_SKIP_CHIP"]
