"""tmac_amd — MI355X-native implementation of T-MAC's LUT mpGEMM hot path.

The product is ``lib/libtmac_hip.so`` (HIP kernels + C++ dispatch behind the C-ABI of
``include/tmac_hip.h``).  This package is the thin Python host side: a ctypes binding
(:mod:`.binding`), the offline weight transform (:mod:`.weights`, mirrors
``python/t_mac/weights.py`` of the reference) and a ``TMACGeMMWrapper`` with the reference's
method names (:mod:`.wrapper`, mirrors ``include/t-mac/tmac_gemm_wrapper.h``).

The package directory is ``tmac_amd/``; ``t-mac_amd`` at the repo root is a symlink to it (the project's name with its hyphen).
"""
from .binding import (TMACHipError, KCfg, lib, lib_path, F32, F16, load_library, build_library)  # noqa: F401
from .weights import preprocess_weights  # noqa: F401
from .wrapper import TMACGeMMWrapper, Weights, Workspace, DecodeChain, Comm  # noqa: F401
from . import convert, weights, sharding, binding  # noqa: F401

__version__ = "0.1.0"
