"""Backend for tests/fixture_runner.py that drives the PRODUCT library through its
C-ABI. mode="host": numpy buffers are passed as host pointers (exercises the
runtime's mirror path, the way tpp-run's JIT'd code calls the reference).
mode="device": buffers are uploaded to torch device tensors first and the device
pointers are passed (zero-copy path), then downloaded for checking."""
import importlib

import numpy as np

pkg = importlib.import_module("tpp-mlir_amd")


class AbiBackend:
    def __init__(self, mode="host"):
        self.rt = pkg.get_runtime()
        self.mode = mode
        self.name = "abi-" + mode
        self._dev = {}  # id(numpy array) -> (array, torch tensor)

    def _buf(self, arr):
        if self.mode == "host":
            return arr
        import torch
        key = id(arr)
        if key not in self._dev:
            src = arr.view(np.int16) if arr.dtype == np.uint16 else arr
            self._dev[key] = (arr, torch.from_numpy(src.copy()).cuda())
        return self._dev[key][1]

    def gemm(self, d, A, oa, B, ob, C, oc):
        h = self.rt.gemm_dispatch(d["dtype"], d["m"], d["n"], d["k"], d["lda"], d["ldb"], d["ldc"], d["flags"])
        self.rt.gemm(d["dtype"], h, self._buf(A), oa, self._buf(B), ob, self._buf(C), oc)

    def brgemm(self, d, A, oa, B, ob, C, oc, br):
        h = self.rt.brgemm_dispatch(d["dtype"], d["m"], d["n"], d["k"], d["lda"], d["ldb"], d["ldc"],
                                    d["stride_a"], d["stride_b"], d["flags"])
        self.rt.brgemm(d["dtype"], h, self._buf(A), oa, self._buf(B), ob, self._buf(C), oc, br)

    def fused_brgemm(self, d, A, oa, B, ob, C, oc, D, od, br):
        h = self.rt.fused_brgemm_dispatch(d["dtype"], d["m"], d["n"], d["k"], d["lda"], d["ldb"], d["ldc"],
                                          d["stride_a"], d["stride_b"], d["flags"], d["unary_flags"],
                                          d["unary_kind"], d["binary_flags"], d["binary_kind"])
        self.rt.fused_brgemm(d["dtype"], h, self._buf(A), oa, self._buf(B), ob, self._buf(C), oc, self._buf(D), od, br)

    def unary(self, d, X, ox, O, oo):
        h = self.rt.unary_dispatch(d["kind"], d["dtype"], d["m"], d["n"], d["ldi"], d["ldo"], d["flags"])
        self.rt.unary(d["dtype"], h, self._buf(X), ox, self._buf(O), oo)

    def binary(self, d, L, ol, R, or_, O, oo):
        h = self.rt.binary_dispatch(d["kind"], d["dtype"], d["m"], d["n"], d["ldi_lhs"], d["ldi_rhs"], d["ldo"],
                                    d["flags"])
        self.rt.binary(d["dtype"], h, self._buf(L), ol, self._buf(R), or_, self._buf(O), oo)

    def finish(self):
        if self.mode == "device":
            self.rt.synchronize()
            for arr, t in self._dev.values():
                host = t.cpu().numpy()
                arr[...] = host.view(np.uint16) if arr.dtype == np.uint16 else host
            self._dev.clear()
