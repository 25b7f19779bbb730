"""ctypes front-end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg. The product package (tpp-mlir_amd) never imports it.
Function-by-function reference citations live in oracle/xsmm_oracle.c and
oracle/tensor_init.cpp.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

F32, BF16 = 1, 2
I64 = ctypes.c_int64
VP = ctypes.c_void_p


def usable_cpus():
    """CPUs this process may really keep busy: the affinity mask AND the cgroup CPU quota (cgroup v2 cpu.max / v1 cfs_quota_us).
    Returns (cpus, detail dict). A container with nproc = 256 and a quota of 16 CPUs can SUSTAIN 16 threads; larger teams are
    throttled by the scheduler and run far slower than 16 threads do."""
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    cpus = aff if quota is None else max(1, min(aff, int(quota)))
    return cpus, {"affinity_cpus": aff, "cgroup_cpu_quota": quota}


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("xsmm_oracle.c", "tensor_init.cpp", "Makefile")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    cb = os.path.join(_HERE, "libcpu_baseline.so")
    src = os.path.join(_HERE, "cpu_baseline.c")
    if force or not os.path.exists(cb) or os.path.getmtime(src) > os.path.getmtime(cb):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libcpu_baseline.so"])
    return so


class CpuBaseline:
    """oracle/cpu_baseline.c: the reference's packed 32x32x32-tile batch-reduce call structure under OpenMP.
    native=True compiles it for THIS host (-O3 -march=native) first and falls back to the portable build."""

    def __init__(self, native=True):
        self.flags = "-O3 -march=x86-64-v3 -fopenmp (portable build)"
        path = os.path.join(_HERE, "libcpu_baseline.so")
        if native:
            nat = os.path.join(_HERE, "_cpu_native.so")
            try:
                subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-fopenmp", "-shared",
                                       os.path.join(_HERE, "cpu_baseline.c"), "-o", nat],
                                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                path, self.flags = nat, "-O3 -march=native -fopenmp (compiled on this host)"
            except Exception:
                pass
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        for f in ("cpu_pack_blocks", "cpu_unpack_blocks", "cpu_pack_b"):
            getattr(L, f).argtypes = [VP, I64, I64, I64, VP]
            getattr(L, f).restype = None
        L.cpu_brgemm_tiled_f32.argtypes = [I64, I64, I64, VP, VP, VP, ctypes.c_int, I64]
        L.cpu_brgemm_tiled_f32.restype = ctypes.c_int
        L.cpu_baseline_threads.restype = ctypes.c_int
        self.L = L

    def threads(self):
        return self.L.cpu_baseline_threads()

    def set_threads(self, n):
        self.L.cpu_baseline_set_threads.argtypes = [ctypes.c_int]
        self.L.cpu_baseline_set_threads(int(n))
        return self.threads()

    def pack(self, A, B, C, m, n, k):
        """row-major A [m][k], B [k][n], C [m][n] -> the packed block buffers the tile loop runs on"""
        Ap, Bp, Cp = np.empty(m * k, np.float32), np.empty(k * n, np.float32), np.empty(m * n, np.float32)
        self.L.cpu_pack_blocks(_p(A), m, k, k, _p(Ap))
        self.L.cpu_pack_b(_p(B), k, n, n, _p(Bp))
        self.L.cpu_pack_blocks(_p(C), m, n, n, _p(Cp))
        return Ap, Bp, Cp

    def unpack_c(self, Cp, m, n):
        C = np.empty(m * n, np.float32)
        self.L.cpu_unpack_blocks(_p(Cp), m, n, n, _p(C))
        return C

    def run(self, m, n, k, Ap, Bp, Cp, beta0, reps=1):
        rc = self.L.cpu_brgemm_tiled_f32(m, n, k, _p(Ap), _p(Bp), _p(Cp), 1 if beta0 else 0, reps)
        assert rc == 0


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        try:
            build()  # (a no-op unless a source is newer than the library)
        except Exception:
            if not os.path.exists(so):
                raise
        try:
            L = ctypes.CDLL(so)
        except OSError:
            build(force=True)
            L = ctypes.CDLL(so)
        L.oracle_fused_brgemm.restype = ctypes.c_int
        L.oracle_fused_brgemm.argtypes = [I64] * 14 + [VP, VP, VP, VP, I64]
        L.oracle_brgemm.restype = ctypes.c_int
        L.oracle_brgemm.argtypes = [I64] * 10 + [VP, VP, VP, I64]
        L.oracle_gemm.restype = ctypes.c_int
        L.oracle_gemm.argtypes = [I64] * 8 + [VP, VP, VP]
        L.oracle_unary.restype = ctypes.c_int
        L.oracle_unary.argtypes = [I64] * 7 + [VP, VP, VP]
        L.oracle_binary.restype = ctypes.c_int
        L.oracle_binary.argtypes = [I64] * 8 + [VP, VP, VP]
        L.oracle_fused_brgemm_omp.restype = ctypes.c_int
        L.oracle_fused_brgemm_omp.argtypes = [I64] * 12 + [VP, VP, VP, VP, I64, I64]
        L.oracle_num_threads.restype = ctypes.c_int
        L.oracle_f32_to_bf16.restype = ctypes.c_uint16
        L.oracle_f32_to_bf16.argtypes = [ctypes.c_float]
        L.oracle_bf16_to_f32.restype = ctypes.c_float
        L.oracle_bf16_to_f32.argtypes = [ctypes.c_uint16]
        L.tinit_create.restype = VP
        L.tinit_create.argtypes = [ctypes.c_int, ctypes.c_int]
        L.tinit_destroy.argtypes = [VP]
        L.tinit_fill.argtypes = [VP, VP, I64]
        _LIB = L
    return _LIB


# ---- bf16 helpers on numpy (uint16 storage) -------------------------------------
def f32_to_bf16(x):
    """Round-to-nearest-even f32 -> bf16 bits (uint16), vectorised; NaN -> qNaN."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r[nan] = ((u[nan] >> 16) | 0x40).astype(np.uint16)
    return r


def bf16_to_f32(h):
    h = np.ascontiguousarray(h, dtype=np.uint16)
    return (h.astype(np.uint32) << 16).view(np.float32)


def np_dtype(dt):
    return np.float32 if dt == F32 else np.uint16


def _p(arr, off=0):
    """pointer to element `off` of a 1-D contiguous numpy buffer"""
    if arr is None:
        return None
    return ctypes.c_void_p(arr.ctypes.data + off * arr.itemsize)


# ---- ops (buffers are flat numpy arrays; offsets in elements, as on the wire) ----
def brgemm(dt, m, n, k, lda, ldb, ldc, sa, sb, flags, A, offA, B, offB, C, offC, br):
    rc = lib().oracle_brgemm(dt, m, n, k, lda, ldb, ldc, sa, sb, flags, _p(A, offA), _p(B, offB), _p(C, offC), br)
    assert rc == 0, "oracle_brgemm: unsupported arguments"


def gemm(dt, m, n, k, lda, ldb, ldc, flags, A, offA, B, offB, C, offC):
    rc = lib().oracle_gemm(dt, m, n, k, lda, ldb, ldc, flags, _p(A, offA), _p(B, offB), _p(C, offC))
    assert rc == 0


def fused_brgemm(dt, m, n, k, lda, ldb, ldc, sa, sb, gflags, uflags, ukind, bflags, bkind,
                 A, offA, B, offB, C, offC, D, offD, br):
    rc = lib().oracle_fused_brgemm(dt, m, n, k, lda, ldb, ldc, sa, sb, gflags, uflags, ukind, bflags, bkind,
                                   _p(A, offA), _p(B, offB), _p(C, offC), _p(D, offD), br)
    assert rc == 0, "oracle_fused_brgemm: unsupported arguments"


def set_vnni_factor(v):
    """the oracle's stand-in for libxsmm_cpuid_dot_pack_factor(BF16) (VNNIUtils.cpp:25-45): 2 (default) or 4; returns the old one"""
    L = lib()
    L.oracle_set_vnni_factor.argtypes = [ctypes.c_int]
    old = L.oracle_get_vnni_factor()
    assert L.oracle_set_vnni_factor(int(v)) == 0, "VNNI factor must be 2 or 4"
    return old


def pack_vnni(w, k, n, v):
    """row-major bf16 bits [k][n] -> VNNI-v [k/v][n][v] (MLIRGen.cpp:657-664 / VNNIUtils.cpp:75-77); a pure index move"""
    w = np.asarray(w).reshape(k, n)
    return np.ascontiguousarray(w.reshape(k // v, v, n).transpose(0, 2, 1)).reshape(-1)


def unary(kind, dt, m, n, ldi, ldo, flags, inp, offIn, out, offOut):
    rc = lib().oracle_unary(kind, dt, m, n, ldi, ldo, flags, _p(inp, offIn), _p(out, offOut), None)
    assert rc == 0, "oracle_unary: unsupported arguments"


def unary_scalar(kind, dt, m, n, ldi, ldo, flags, scalar, out, offOut):
    s = ctypes.c_float(scalar)
    rc = lib().oracle_unary(kind, dt, m, n, ldi, ldo, flags, None, _p(out, offOut), ctypes.byref(s))
    assert rc == 0


def binary(kind, dt, m, n, ldl, ldr, ldo, flags, lhs, offL, rhs, offR, out, offO):
    rc = lib().oracle_binary(kind, dt, m, n, ldl, ldr, ldo, flags, _p(lhs, offL), _p(rhs, offR), _p(out, offO))
    assert rc == 0, "oracle_binary: unsupported arguments"


def fused_brgemm_omp(dt, m, n, k, lda, ldb, ldc, sa, sb, gflags, ukind, bkind, A, B, C, D, br, row_block=32):
    rc = lib().oracle_fused_brgemm_omp(dt, m, n, k, lda, ldb, ldc, sa, sb, gflags, ukind, bkind,
                                       _p(A), _p(B), _p(C), _p(D), br, row_block)
    assert rc == 0


def num_threads():
    return lib().oracle_num_threads()


class TensorInit:
    """One cached generator of tpp-run (per init type / dtype / seed); fill the
    kernel arguments of one dtype in argument order from a single instance."""
    KINDS = {"const": 0, "simple": 1, "cont": 2, "random": 3, "normal": 4}

    def __init__(self, kind="normal", seed=123):
        self._h = lib().tinit_create(self.KINDS[kind], seed)

    def fill(self, n, dt=F32):
        out = np.empty(n, dtype=np.float32)
        lib().tinit_fill(self._h, out.ctypes.data, n)
        return out if dt == F32 else f32_to_bf16(out)

    def __del__(self):
        try:
            lib().tinit_destroy(self._h)
        except Exception:
            pass


# ---------------------------------------------------------------- mlir-gen's inputs for the MLP benchmark (test infrastructure)
def glibc_rand_sequence(seed, n):
    """the first n values of glibc's rand() after srand(seed) (TYPE_3 additive feedback generator r[i] = r[i-3] + r[i-31],
    310 values discarded, top 31 bits returned) - what mlir-gen's getRand() draws its per-tensor seeds from
    (tools/mlir-gen/MLIRGen.cpp:131-137: srand(seed); :810-819: `temp = seed; seed = rand(); return temp`)"""
    seed &= 0xffffffff
    r = [seed if seed else 1]
    for i in range(1, 31):
        prev = r[i - 1] if r[i - 1] < 2 ** 31 else r[i - 1] - 2 ** 32
        hi, lo = divmod(prev, 127773)
        v = 16807 * lo - 2836 * hi
        r.append(v + 2147483647 if v < 0 else v)
    r += r[0:3]
    out = []
    for i in range(34, 344 + n):
        r.append((r[i - 31] + r[i - 3]) & 0xffffffff)
        if i >= 344:
            out.append(r[i] >> 1)
    return out


def mlir_gen_seed_chain(seed, n):
    """the seeds mlir-gen hands to its n dense constants, in creation order: seed, rand(), rand(), ... (MLIRGen.cpp:810-819)"""
    return ([seed] + glibc_rand_sequence(seed, n))[:n]


def mlir_gen_mlp_tensors(layers, seed=123, dt=BF16, bias=True):
    """weights ([K][N] flat) and biases of `mlir-gen --kernel=const --seed=S --layers=...` in creation order W0, b0, W1, b1, ...
    (MLIRGen.cpp:243-262: createDenseTensor(initType, type, getRand()) per tensor): EVERY tensor has its own `normal`
    generator (TensorInit.cpp:75-96: one cached generator per (type, dtype, seed)) seeded from the chain. The kernel's input
    is a function argument: tpp-run fills it from ITS --seed (MLIRBench.cpp:207-246) - TensorInit('normal', seed).fill()."""
    n_t = (len(layers) - 1) * (2 if bias else 1)
    seeds = mlir_gen_seed_chain(seed, n_t)
    W, Bs, i = [], [], 0
    for k, n in zip(layers[:-1], layers[1:]):
        W.append(TensorInit("normal", seeds[i]).fill(k * n, dt))
        i += 1
        if bias:
            Bs.append(TensorInit("normal", seeds[i]).fill(n, dt))
            i += 1
    return W, Bs, seeds

