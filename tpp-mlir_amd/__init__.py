"""tpp-mlir_amd - MI355X-native runtime behind tpp-mlir's xsmm dispatch/invoke C-ABI.

The product is csrc/ (hand-written gfx950 HIP kernels + the C-ABI, built into
libtpp_xsmm_runner_utils.so); this package is the thin host-side mirror of the
reference's operator interface used by tests, bench.py and the multi-GPU MLP.
The directory name carries a hyphen (it is the reference's name + _amd), so import
it with importlib.import_module("tpp-mlir_amd").
"""
from .runtime import (BinaryFlags, BinaryKind, DataType, GemmFlags, REFERENCE_SYMBOLS, UnaryFlags,  # noqa: F401
                      UnaryKind, XsmmRuntime, get_runtime, library_path, load_library)
from .mlp import (ColumnShardedMlp, MlpSpec, ShardedMlp, all_gather_rows, gathered_to_rows,  # noqa: F401
                  layer_dispatch_args, row_partition)
from .peer import PeerGather  # noqa: F401
from .build import build, build_tools  # noqa: F401
