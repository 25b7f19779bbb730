"""Host-side mirror of the xsmm dialect's dispatch/invoke op pairs over the C-ABI of
libtpp_xsmm_runner_utils.so (include/tpp_xsmm_abi.h).

Every method is a 1:1 binding of one exported symbol, with the reference's argument
order (lib/TPP/Conversion/ConvertXsmmToFunc/ConvertXsmmToFunc.cpp:298-352 and the
FileCheck'd call sites in test/Conversion/XsmmToFunc/xsmm-to-func.mlir): a memref
operand is (buffer, element offset); a dispatch returns an opaque i64; an invoke
takes the dtype and that i64 first. Nothing here computes: if the HIP library is
missing or no MI355X is visible, calls fail loudly (there is no CPU fallback).

Operands may be torch tensors (their device or host storage is used in place /
mirrored by the runtime), numpy arrays (host pointers), or raw integer addresses.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_NAME = "libtpp_xsmm_runner_utils.so"

I64 = ctypes.c_int64
VP = ctypes.c_void_p


class DataType:  # XsmmEnum.td:13-20
    F32 = 1
    BF16 = 2


class UnaryKind:  # XsmmEnum.td:34-45
    NONE, IDENTITY, ZERO, RELU, VNNI2, TRANSPOSE = 0, 1, 2, 5, 28, 29


class UnaryFlags:  # XsmmEnum.td:47-56
    NONE, BCAST_ROW, BCAST_COL, BCAST_SCALAR = 0, 2, 4, 8


class BinaryKind:  # XsmmEnum.td:22-32
    NONE, ADD, MUL, SUB, DIV = 0, 1, 2, 3, 4


class BinaryFlags:  # XsmmEnum.td:58-70
    NONE = 0
    BCAST_ROW_IN_0, BCAST_ROW_IN_1 = 1, 2
    BCAST_COL_IN_0, BCAST_COL_IN_1 = 4, 8
    BCAST_SCALAR_IN_0, BCAST_SCALAR_IN_1 = 16, 32


class GemmFlags:
    """Values AS THEY TRAVEL ON THE WIRE: the dialect's vnni_b (4096) reaches the
    runtime as 2048 and vnni_a (2048) as 4096 (ConvertXsmmToFunc.cpp:251-265)."""
    NONE, BETA_0 = 0, 4
    NO_RESET_TILECONFIG, NO_SETUP_TILECONFIG = 64, 128
    VNNI_B = 2048
    VNNI_A = 4096
    VNNI_C = 8192


_SIGNATURES = {
    # name: (restype, argtypes)
    "xsmm_gemm_dispatch": (I64, [I64] * 8),
    "xsmm_brgemm_dispatch": (I64, [I64] * 10),
    "xsmm_fused_brgemm_dispatch": (I64, [I64] * 14),
    "xsmm_unary_dispatch": (I64, [I64] * 7),
    "xsmm_binary_dispatch": (I64, [I64] * 8),
    "xsmm_intel_amx_tile_config_dispatch": (I64, [I64] * 10),
    "xsmm_gemm_invoke": (None, [I64, I64, VP, I64, VP, I64, VP, I64]),
    "xsmm_brgemm_invoke": (None, [I64, I64, VP, I64, VP, I64, VP, I64, I64]),
    "xsmm_fused_brgemm_invoke": (None, [I64, I64, VP, I64, VP, I64, VP, I64, VP, I64, I64]),
    "xsmm_unary_invoke": (None, [I64, I64, VP, I64, VP, I64]),
    "xsmm_unary_scalar_invoke": (None, [I64, I64, ctypes.c_float, VP, I64]),
    "xsmm_binary_invoke": (None, [I64, I64, VP, I64, VP, I64, VP, I64]),
    "xsmm_intel_amx_tile_config_invoke": (None, [I64, I64, VP, I64]),
    "perf_start_timer": (I64, []),
    "perf_stop_timer": (ctypes.c_double, [I64]),
    # extensions (not in the reference ABI)
    "xsmm_hip_set_async": (ctypes.c_int, [ctypes.c_int]),
    "xsmm_hip_set_stream": (None, [VP]),
    "xsmm_hip_set_tile_queue": (ctypes.c_int, [ctypes.c_int]),
    "xsmm_hip_flush": (None, []),
    "xsmm_hip_fused_brgemm_chain_invoke": (ctypes.c_int, [I64, I64, ctypes.POINTER(I64), ctypes.POINTER(VP), ctypes.POINTER(I64),
                                                          ctypes.POINTER(VP), ctypes.POINTER(I64), ctypes.POINTER(VP),
                                                          ctypes.POINTER(I64), ctypes.POINTER(VP), ctypes.POINTER(I64),
                                                          ctypes.POINTER(I64)]),
    "xsmm_hip_peer_alloc": (VP, [I64]),
    "xsmm_hip_peer_free": (None, [VP]),
    "xsmm_hip_ipc_export": (ctypes.c_int, [VP, VP]),
    "xsmm_hip_ipc_open": (VP, [VP]),
    "xsmm_hip_ipc_close": (ctypes.c_int, [VP]),
    "xsmm_hip_peer_gather": (None, [VP, I64, I64, I64, I64, ctypes.POINTER(VP), ctypes.POINTER(VP), ctypes.POINTER(VP), VP, VP, VP, VP, I64]),
    "xsmm_hip_peer_overlap": (ctypes.c_int, [ctypes.c_int]),
    "xsmm_hip_peer_wait_stream": (VP, []),
    "xsmm_hip_peer_drain": (None, []),
    "xsmm_hip_chain_status": (ctypes.c_int64, []),
    "xsmm_hip_set_strict": (ctypes.c_int, [ctypes.c_int]),
    "xsmm_hip_get_strict": (ctypes.c_int, []),
    "xsmm_hip_set_launch_thread": (ctypes.c_int, [ctypes.c_int]),
    "xsmm_hip_launch_thread_stats": (None, [ctypes.POINTER(ctypes.c_int64)]),
    "xsmm_hip_tile_queue_stats": (None, [ctypes.POINTER(ctypes.c_int64)]),
    "xsmm_hip_get_stream": (VP, []),
    "xsmm_hip_synchronize": (None, []),
    "xsmm_hip_host_resident": (ctypes.c_int, [VP, I64]),
    "xsmm_hip_host_update": (ctypes.c_int, [VP]),
    "xsmm_hip_host_release": (ctypes.c_int, [VP]),
    "xsmm_hip_set_host_cache": (ctypes.c_int, [ctypes.c_int]),
    "xsmm_hip_host_cache_stats": (None, [ctypes.POINTER(ctypes.c_int64)]),
    "xsmm_hip_device_count": (ctypes.c_int, []),
    "xsmm_hip_kernel_name": (ctypes.c_char_p, [I64]),
    "xsmm_hip_last_grouped_kernel": (ctypes.c_char_p, []),
    "xsmm_hip_last_refined_kernel": (ctypes.c_char_p, []),
    "xsmm_hip_force_variant": (None, [ctypes.c_int]),
    "xsmm_hip_force_split": (ctypes.c_int, [ctypes.c_int]),
    "xsmm_hip_set_fold_transpose": (ctypes.c_int, [ctypes.c_int]),
    "xsmm_hip_fold_transpose_stats": (None, [ctypes.POINTER(ctypes.c_int64)]),
    "xsmm_hip_set_vnni_factor": (ctypes.c_int, [ctypes.c_int]),
    "xsmm_hip_get_vnni_factor": (ctypes.c_int, []),
    "xsmm_hip_version": (ctypes.c_char_p, []),
}

REFERENCE_SYMBOLS = [n for n in _SIGNATURES if not n.startswith("xsmm_hip_")]


def library_path():
    return os.environ.get("TPP_XSMM_LIBRARY", os.path.join(_HERE, SO_NAME))


def load_library(path=None):
    """dlopen the runtime. torch (if installed) is imported first so that ONE HIP
    runtime (the one torch already mapped) serves both."""
    path = path or library_path()
    if not os.path.exists(path):
        raise FileNotFoundError(
            "%s is not built: run `python -c \"import __graft_entry__ as g; g.build()\"` (hipcc, gfx950)" % path)
    try:
        import torch  # noqa: F401  (maps libamdhip64 before our library resolves it)
    except Exception:  # torch is plumbing, not a requirement
        pass
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = missing export
        fn.restype = res
        fn.argtypes = args
    return lib


def _addr(x):
    """raw address of an operand buffer"""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):  # torch.Tensor
        return x.data_ptr()
    if hasattr(x, "ctypes"):  # numpy.ndarray
        return x.ctypes.data
    if isinstance(x, ctypes.c_void_p):
        return x.value
    raise TypeError("unsupported operand type %r" % type(x))


class XsmmRuntime:
    """The 13 xsmm_* + 2 perf_* entry points, named after the dialect ops
    (xsmm.brgemm.dispatch -> brgemm_dispatch, xsmm.brgemm -> brgemm)."""

    def __init__(self, path=None):
        self.lib = load_library(path)

    # ---- dispatch ----------------------------------------------------------------
    def gemm_dispatch(self, dtype, m, n, k, lda, ldb, ldc, flags=0):
        return self.lib.xsmm_gemm_dispatch(dtype, m, n, k, lda, ldb, ldc, flags)

    def brgemm_dispatch(self, dtype, m, n, k, lda, ldb, ldc, stride_a, stride_b, flags=0):
        return self.lib.xsmm_brgemm_dispatch(dtype, m, n, k, lda, ldb, ldc, stride_a, stride_b, flags)

    def fused_brgemm_dispatch(self, dtype, m, n, k, lda, ldb, ldc, stride_a, stride_b, gemm_flags=0,
                              unary_flags=0, unary_kind=0, binary_flags=0, binary_kind=0):
        return self.lib.xsmm_fused_brgemm_dispatch(dtype, m, n, k, lda, ldb, ldc, stride_a, stride_b, gemm_flags,
                                                   unary_flags, unary_kind, binary_flags, binary_kind)

    def unary_dispatch(self, kind, dtype, m, n, ldi, ldo, flags=0):
        return self.lib.xsmm_unary_dispatch(kind, dtype, m, n, ldi, ldo, flags)

    def binary_dispatch(self, kind, dtype, m, n, ldi_lhs, ldi_rhs, ldo, flags=0):
        return self.lib.xsmm_binary_dispatch(kind, dtype, m, n, ldi_lhs, ldi_rhs, ldo, flags)

    def intel_amx_tile_config_dispatch(self, dtype, m, n, k, lda, ldb, ldc, stride_a, stride_b, flags=0):
        return self.lib.xsmm_intel_amx_tile_config_dispatch(dtype, m, n, k, lda, ldb, ldc, stride_a, stride_b, flags)

    # ---- invoke ------------------------------------------------------------------
    def gemm(self, dtype, handle, a, off_a, b, off_b, c, off_c):
        self.lib.xsmm_gemm_invoke(dtype, handle, _addr(a), off_a, _addr(b), off_b, _addr(c), off_c)

    def brgemm(self, dtype, handle, a, off_a, b, off_b, c, off_c, num_batches):
        self.lib.xsmm_brgemm_invoke(dtype, handle, _addr(a), off_a, _addr(b), off_b, _addr(c), off_c, num_batches)

    def fused_brgemm(self, dtype, handle, a, off_a, b, off_b, c, off_c, d, off_d, num_batches):
        self.lib.xsmm_fused_brgemm_invoke(dtype, handle, _addr(a), off_a, _addr(b), off_b, _addr(c), off_c,
                                          _addr(d), off_d, num_batches)

    def unary(self, dtype, handle, inp, off_in, out, off_out):
        self.lib.xsmm_unary_invoke(dtype, handle, _addr(inp), off_in, _addr(out), off_out)

    def unary_scalar(self, dtype, handle, scalar, out, off_out):
        self.lib.xsmm_unary_scalar_invoke(dtype, handle, float(scalar), _addr(out), off_out)

    def binary(self, dtype, handle, lhs, off_lhs, rhs, off_rhs, out, off_out):
        self.lib.xsmm_binary_invoke(dtype, handle, _addr(lhs), off_lhs, _addr(rhs), off_rhs, _addr(out), off_out)

    def intel_amx_tile_config(self, dtype, handle, tile_state, off=0):
        self.lib.xsmm_intel_amx_tile_config_invoke(dtype, handle, _addr(tile_state), off)

    # ---- timers (runtime/PerfRunnerUtils.cpp) ---------------------------------------
    def perf_start_timer(self):
        return self.lib.perf_start_timer()

    def perf_stop_timer(self, start):
        return self.lib.perf_stop_timer(start)

    # ---- extensions ----------------------------------------------------------------
    def set_async(self, enable):
        return bool(self.lib.xsmm_hip_set_async(1 if enable else 0))

    def set_stream(self, stream):
        """stream: raw hipStream_t value, or a torch.cuda.Stream"""
        self.lib.xsmm_hip_set_stream(getattr(stream, "cuda_stream", stream) or None)

    def set_tile_queue(self, enable):
        return int(self.lib.xsmm_hip_set_tile_queue(int(enable)))  # 0 off, 1 on, 2 on + scheduler thread for several callers; returns the previous mode

    def flush(self):
        self.lib.xsmm_hip_flush()

    def fused_brgemm_chain(self, dtype, calls):
        """calls: [(handle, a, off_a, b, off_b, c, off_c, d, off_d, num_batches)] - the effect of fused_brgemm on each in
        order; returns True if the chain ran as ONE launch (see the header), False if it ran call by call"""
        return bool(self.lib.xsmm_hip_fused_brgemm_chain_invoke(dtype, len(calls), *self.pack_chain(calls)))

    @staticmethod
    def pack_chain(calls):
        """the argument arrays of xsmm_hip_fused_brgemm_chain_invoke for `calls` (reusable: a timing loop packs once)"""
        n = len(calls)
        cols = list(zip(*calls))
        i64 = lambda v: (I64 * n)(*[int(x) for x in v])  # noqa: E731
        ptr = lambda v: (VP * n)(*[_addr(x) for x in v])  # noqa: E731
        return (i64(cols[0]), ptr(cols[1]), i64(cols[2]), ptr(cols[3]), i64(cols[4]), ptr(cols[5]), i64(cols[6]),
                ptr(cols[7]), i64(cols[8]), i64(cols[9]))

    def tile_queue_stats(self):
        """(grouped launches, invokes with full bookkeeping, invokes replayed, groups ended by a known terminator, replays abandoned)"""
        out = (ctypes.c_int64 * 5)()
        self.lib.xsmm_hip_tile_queue_stats(out)
        return tuple(out)

    def set_strict(self, on):
        """strict mode: kernel choice by descriptor + batch count only (include/tpp_xsmm_abi.h); -1 if refused"""
        return self.lib.xsmm_hip_set_strict(1 if on else 0)

    def get_strict(self):
        return self.lib.xsmm_hip_get_strict()

    def set_launch_thread(self, on):
        """complete replayed groups launched by the runtime's launch thread (default) or by the caller; previous setting"""
        return self.lib.xsmm_hip_set_launch_thread(1 if on else 0)

    def launch_thread_stats(self):
        """(launches handed to the launch thread since process start, 1 if the thread exists right now)"""
        out = (ctypes.c_int64 * 2)()
        self.lib.xsmm_hip_launch_thread_stats(out)
        return tuple(out)

    def chain_status(self):
        """starved chain launches found and re-run call by call since process start (0: never)"""
        return int(self.lib.xsmm_hip_chain_status())

    def synchronize(self):
        self.lib.xsmm_hip_synchronize()

    def host_resident(self, buf, nbytes=None):
        """declare a long-lived host buffer (numpy array): uploaded once, used from its device copy afterwards"""
        return self.lib.xsmm_hip_host_resident(_addr(buf), int(nbytes if nbytes is not None else buf.nbytes))

    def host_update(self, buf):
        return self.lib.xsmm_hip_host_update(_addr(buf))

    def host_release(self, buf):
        return self.lib.xsmm_hip_host_release(_addr(buf))

    def set_host_cache(self, on):
        """host operands keep a device mirror between invokes; only pages the host wrote are uploaded again (include/tpp_xsmm_abi.h).
        Returns the previous setting, -1 if the kernel lacks userfaultfd WP_ASYNC / PAGEMAP_SCAN"""
        return self.lib.xsmm_hip_set_host_cache(1 if on else 0)

    def host_cache_stats(self):
        """dict of the ten counters of xsmm_hip_host_cache_stats"""
        out = (ctypes.c_int64 * 10)()
        self.lib.xsmm_hip_host_cache_stats(out)
        names = ("extents", "mirror_bytes", "uploaded_bytes", "scans", "written_back_bytes", "pages_not_written_back", "grows", "fast_invokes",
                 "slow_invokes", "extents_dropped")
        return dict(zip(names, out))

    def device_count(self):
        return self.lib.xsmm_hip_device_count()

    def kernel_name(self, handle):
        return self.lib.xsmm_hip_kernel_name(handle).decode()

    def last_grouped_kernel(self):
        return self.lib.xsmm_hip_last_grouped_kernel().decode()

    def last_refined_kernel(self):
        """the kernel an invoke-time refinement chose for the most recent non-queued GEMM invoke, "" = the handle's own kernel"""
        return self.lib.xsmm_hip_last_refined_kernel().decode()

    def force_variant(self, v):
        self.lib.xsmm_hip_force_variant(v)

    def force_split(self, workgroups_per_tile):
        """-1 the model, 0 / 1 never, n > 1: n workgroups share the batch-reduce range of one f32 output tile; returns the previous setting"""
        return self.lib.xsmm_hip_force_split(workgroups_per_tile)

    def set_fold_transpose(self, enable):
        """transposes that feed a gemm's B operand folded into the gemm (default on); returns the previous setting"""
        return self.lib.xsmm_hip_set_fold_transpose(1 if enable else 0)

    def fold_transpose_stats(self):
        """(gemm invokes served from a transpose's source, remembered transposes dropped as dead, remembered transposes launched)"""
        out = (ctypes.c_int64 * 3)()
        self.lib.xsmm_hip_fold_transpose_stats(out)
        return tuple(int(v) for v in out)

    def set_vnni_factor(self, v):
        """VNNI blocking factor (2 / 4) of bf16 B operands dispatched from now on; returns the previous one"""
        old = self.lib.xsmm_hip_set_vnni_factor(int(v))
        if old < 0:
            raise ValueError("the VNNI factor is 2 or 4")
        return old

    def version(self):
        return self.lib.xsmm_hip_version().decode()


_RT = None


def get_runtime():
    global _RT
    if _RT is None:
        _RT = XsmmRuntime()
    return _RT
