"""carnd_mpc_project_amd -- MI355X-native batched MPC solver (host-side Python glue).

The product is the C-ABI library ``lib/libmpc_amd.so`` (HIP kernels for gfx950,
``include/mpc_amd.h``).  This package only binds it with ctypes, moves batches
between torch/numpy and the ABI's struct-of-arrays layout, and generates the
synthetic batches BASELINE.json names.  There is no CPU compute path here: if
the library is missing or no gfx950 device is present, solving raises.

The directory is called ``carnd-mpc-project_amd`` (not importable as written);
``__graft_entry__.load_package()`` imports it under the name
``carnd_mpc_project_amd``.
"""
from ._abi import (MpcParams, MpcBatchStats, MpcWireTelemetry, MpcError, library, library_path, build_library,
                   params_default, params_from_json, inflight_advice, STATUS_NAMES, PRECISION_F32, PRECISION_F64)
from .solver import BatchedMPC
from . import scenarios
from . import sharding

__all__ = ["MpcParams", "MpcBatchStats", "MpcWireTelemetry", "MpcError", "library", "library_path", "build_library",
           "params_default", "params_from_json", "inflight_advice", "BatchedMPC", "scenarios", "sharding", "STATUS_NAMES",
           "PRECISION_F32", "PRECISION_F64"]
