#!/usr/bin/env python3
"""Harvest golden vectors for the xsmm hot path from the reference's own lit tests.

Run in the build container (needs /root/reference, which never travels to the GPU
box):      python tests/golden/harvest.py
It writes one JSON fixture per reference test into tests/golden/. A fixture is
DATA ONLY: input buffers, the dispatch/invoke call list (the argument tuples the
reference's FileCheck lines pin), and the expected numbers the reference's
CHECK / check.expect_almost_eq lines assert. No reference source text is copied:
literal matrices and CHECK numbers are parsed out of the .mlir files as numbers.

Where a reference test exercises the compiler (linalg IR lowered by tpp-opt) rather
than hand-written xsmm ops, the call list replays the dispatch tuple pinned by the
test's `IR:` FileCheck lines over the loop nest implied by its indexing maps; the
expected values are the test's CHECK lines either way.

Seeded inputs (`-seed=123`) come from oracle/tensor_init.cpp (restatement of
TensorInit*.cpp); their values are stored literally so the fixture also pins the
generator.
"""
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from oracle import pyoracle as orc  # noqa: E402

F32, BF16 = 1, 2
NUM = r"[-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+)"


def read(rel):
    with open(os.path.join(REF, rel)) as f:
        return f.read()


def dense_literals(text):
    """all `dense<[ ... ]>` literal tensors in file order, flattened row-major"""
    out = []
    for m in re.finditer(r"dense<\s*(\[.*?\])\s*>", text, re.S):
        out.append([float(x) for x in re.findall(NUM, m.group(1))])
    return out


def check_numbers(text, prefix="CHECK", after=None, count=None):
    """numbers on `// <prefix>[-SAME]: ( ... )` lines, in file order"""
    vals = []
    started = after is None
    for line in text.splitlines():
        if not started:
            started = after in line
            continue
        m = re.match(r"\s*//\s*%s(?:-SAME)?:\s*(.*)$" % re.escape(prefix), line)
        if m and "(" in m.group(1):
            vals += [float(x) for x in re.findall(NUM, m.group(1))]
            if count is not None and len(vals) >= count:
                return vals[:count]
    return vals


def check_fill(text, prefix="CHECK"):
    """the single value a test asserts for every element: all numbers on its `// <prefix>[-COUNT-n | -SAME]: ( ... )`
    lines (they must all agree), or - tests that compare against a constant buffer - the `%outVal` constant of
    check.expect_almost_eq"""
    vals = []
    for line in text.splitlines():
        m = re.match(r"\s*//\s*%s(?:-COUNT-\d+|-SAME)?:\s*(\(.*)$" % re.escape(prefix), line)
        if m:
            vals += [float(x) for x in re.findall(NUM, m.group(1))]
    if not vals and "check.expect_almost_eq" in text:
        m = re.search(r"%outVal\s*=\s*arith\.constant\s+(" + NUM + r")\s*:", text)
        vals = [float(m.group(1))]
    assert vals and all(v == vals[0] for v in vals), (prefix, vals[:8])
    return vals[0]


def buf(dt, data=None, size=None, const=None):
    if data is not None:
        return {"dtype": dt, "data": [float(x) for x in data]}
    return {"dtype": dt, "size": int(size), "const": float(const)}


def dump(name, fx):
    fx["name"] = name
    path = os.path.join(HERE, name + ".json")
    with open(path, "w") as f:
        json.dump(fx, f, indent=None, separators=(",", ":"))
        f.write("\n")
    print("wrote", os.path.relpath(path, ROOT), "(%d calls)" % len(fx["calls"]))


def brgemm_call(disp, a, b, c, batch):
    keys = ["dtype", "m", "n", "k", "lda", "ldb", "ldc", "stride_a", "stride_b", "flags"]
    return {"op": "brgemm", "dispatch": dict(zip(keys, disp)), "a": a, "b": b, "c": c, "batch": batch}


def gemm_call(disp, a, b, c):
    keys = ["dtype", "m", "n", "k", "lda", "ldb", "ldc", "flags"]
    return {"op": "gemm", "dispatch": dict(zip(keys, disp)), "a": a, "b": b, "c": c}


def fused_call(disp, a, b, c, d, batch):
    keys = ["dtype", "m", "n", "k", "lda", "ldb", "ldc", "stride_a", "stride_b", "flags",
            "unary_flags", "unary_kind", "binary_flags", "binary_kind"]
    return {"op": "fused_brgemm", "dispatch": dict(zip(keys, disp)), "a": a, "b": b, "c": c, "d": d, "batch": batch}


def unary_call(disp, i, o):
    keys = ["kind", "dtype", "m", "n", "ldi", "ldo", "flags"]
    return {"op": "unary", "dispatch": dict(zip(keys, disp)), "in": i, "out": o}


def binary_call(disp, l, r, o):
    keys = ["kind", "dtype", "m", "n", "ldi_lhs", "ldi_rhs", "ldo", "flags"]
    return {"op": "binary", "dispatch": dict(zip(keys, disp)), "lhs": l, "rhs": r, "out": o}


def expect(buffer, rows, cols, ld, values=None, fill=None, tol="printed", offset=0):
    e = {"buffer": buffer, "offset": offset, "rows": rows, "cols": cols, "ld": ld, "tol": tol}
    if values is not None:
        assert len(values) == rows * cols, (len(values), rows, cols)
        e["values"] = [float(v) for v in values]
    else:
        e["fill"] = float(fill)
    return e


def main():
    T = "test/Integration/"
    TB = "test/BF16/Integration/"

    # ---- hand-written xsmm ops, all-ones inputs (tpp-run default init: const 1.0,
    # TensorInit.cpp:84-90 with seed 0) ------------------------------------------
    dump("xsmm_brgemm", {
        "source": [T + "xsmm-brgemm.mlir:5-20"],
        "buffers": {"A": buf(F32, size=2 * 32 * 16, const=1), "B": buf(F32, size=2 * 16 * 64, const=1),
                    "C": buf(F32, size=64 * 32, const=1)},
        "calls": [brgemm_call([F32, 32, 64, 16, 16, 64, 64, 512, 1024, 0], ["A", 0], ["B", 0], ["C", 0], 2)],
        "expect": [expect("C", 32, 64, 64, fill=check_fill(read(T + "xsmm-brgemm.mlir")), tol="exact")]})
    dump("xsmm_ternary", {
        "source": [T + "xsmm-ternary.mlir:5-16"],
        "buffers": {"A": buf(F32, size=24, const=1), "B": buf(F32, size=24, const=1), "C": buf(F32, size=9, const=1)},
        "calls": [brgemm_call([F32, 3, 3, 4, 4, 3, 3, 12, 12, 0], ["A", 0], ["B", 0], ["C", 0], 2)],
        "expect": [expect("C", 3, 3, 3, fill=check_fill(read(T + "xsmm-ternary.mlir")), tol="exact")]})
    # fused: wire tuple (gemm_flags, unary_flags, unary_kind, binary_flags, binary_kind) = (0,0,5,4,1)
    dump("xsmm_quarternary", {
        "source": [T + "xsmm-quarternary.mlir:4-15"],
        "buffers": {"A": buf(F32, size=64 * 16, const=1), "B": buf(F32, size=64 * 16, const=1),
                    "C": buf(F32, size=16, const=1), "D": buf(F32, size=4, const=1)},
        "calls": [fused_call([F32, 4, 4, 4, 4, 4, 4, 8, 8, 0, 0, 5, 4, 1], ["A", 0], ["B", 0], ["C", 0], ["D", 0], 16)],
        "expect": [expect("C", 4, 4, 4, fill=check_fill(read(T + "xsmm-quarternary.mlir")), tol="exact")]})
    dump("xsmm_unary_relu", {
        "source": [T + "xsmm-unary.mlir:5-13"],
        "buffers": {"X": buf(F32, size=9, const=1)},
        "calls": [unary_call([5, F32, 3, 3, 3, 3, 0], ["X", 0], ["X", 0])],
        "expect": [expect("X", 3, 3, 3, fill=check_fill(read(T + "xsmm-unary.mlir")), tol="exact")]})
    dump("xsmm_zero", {
        "source": [T + "xsmm-zero.mlir:5-16"],
        "buffers": {"X": buf(F32, size=9, const=5)},
        "calls": [unary_call([2, F32, 3, 3, 3, 3, 0], ["X", 0], ["X", 0])],
        "expect": [expect("X", 3, 3, 3, fill=check_fill(read(T + "xsmm-zero.mlir")), tol="exact")]})
    dump("xsmm_binary_add", {
        "source": [T + "xsmm-binary.mlir:5-17"],
        "buffers": {"L": buf(F32, size=9, const=1), "R": buf(F32, size=9, const=1), "O": buf(F32, size=9, const=1)},
        "calls": [binary_call([1, F32, 3, 3, 3, 3, 3, 0], ["L", 0], ["R", 0], ["O", 0])],
        "expect": [expect("O", 3, 3, 3, fill=check_fill(read(T + "xsmm-binary.mlir")), tol="exact")]})

    # ---- literal matrices -------------------------------------------------------
    t = read(T + "xsmm-mul.mlir")
    a = dense_literals(t)[0]
    dump("xsmm_mul", {
        "source": [T + "xsmm-mul.mlir:5-35"],
        "buffers": {"X": buf(F32, data=a), "O": buf(F32, size=32, const=0)},
        "calls": [binary_call([2, F32, 4, 8, 8, 8, 8, 0], ["X", 0], ["X", 0], ["O", 0])],
        "expect": [expect("O", 4, 8, 8, values=check_numbers(t, count=32))]})
    t = read(T + "xsmm-sub.mlir")
    a = dense_literals(t)[0]
    dump("xsmm_sub", {
        "source": [T + "xsmm-sub.mlir:5-35"],
        "buffers": {"X": buf(F32, data=a), "O": buf(F32, size=32, const=7)},
        "calls": [binary_call([3, F32, 4, 8, 8, 8, 8, 0], ["X", 0], ["X", 0], ["O", 0])],
        "expect": [expect("O", 4, 8, 8, values=check_numbers(t, count=32), tol="exact")]})
    t = read(T + "xsmm-div.mlir")
    lits = dense_literals(t)
    want = check_numbers(t, count=32)
    dump("xsmm_div", {
        "source": [T + "xsmm-div.mlir:5-31", T + "xsmm-div.mlir:79-140"],
        "buffers": {"L": buf(F32, data=lits[0]), "R": buf(F32, data=lits[1]), "Rcol": buf(F32, data=lits[2]),
                    "Rrow": buf(F32, data=lits[3]), "Rsc": buf(F32, data=lits[4]),
                    "O0": buf(F32, size=32, const=0), "O1": buf(F32, size=32, const=0),
                    "O2": buf(F32, size=32, const=0), "O3": buf(F32, size=32, const=0)},
        "calls": [binary_call([4, F32, 4, 8, 8, 8, 8, 0], ["L", 0], ["R", 0], ["O0", 0]),
                  binary_call([4, F32, 4, 8, 8, 8, 8, 8], ["L", 0], ["Rcol", 0], ["O1", 0]),   # bcast_col_in1
                  binary_call([4, F32, 4, 8, 8, 1, 8, 2], ["L", 0], ["Rrow", 0], ["O2", 0]),   # bcast_row_in1
                  binary_call([4, F32, 4, 8, 8, 1, 8, 32], ["L", 0], ["Rsc", 0], ["O3", 0])],  # bcast_scalar_in1
        "expect": [expect(o, 4, 8, 8, values=want, tol="exact") for o in ("O0", "O1", "O2", "O3")]})
    t = read(T + "xsmm-transpose.mlir")
    a = dense_literals(t)[0]
    dump("xsmm_transpose", {
        "source": [T + "xsmm-transpose.mlir:5-38"],
        "buffers": {"X": buf(F32, data=a), "O": buf(F32, size=32, const=0)},
        "calls": [unary_call([29, F32, 4, 8, 8, 4, 0], ["X", 0], ["O", 0])],
        "expect": [expect("O", 8, 4, 4, values=check_numbers(t, count=32))]})
    # bit-exact statement of the same test: the move must not change a single bit
    xt = np.array(a, dtype=np.float32).reshape(4, 8).T.reshape(-1)
    dump("xsmm_transpose_bits", {
        "source": [T + "xsmm-transpose.mlir:5-38 (expected = exact f32 images of the literal inputs)"],
        "buffers": {"X": buf(F32, data=a), "O": buf(F32, size=32, const=0)},
        "calls": [unary_call([29, F32, 4, 8, 8, 4, 0], ["X", 0], ["O", 0])],
        "expect": [expect("O", 8, 4, 4, values=[float(v) for v in xt], tol="exact")]})

    # ---- seeded: xsmm-fusion (A 2x4x8 then bias 1x4 from ONE normal(seed 123) stream) --
    t = read(T + "xsmm-fusion.mlir")
    gen = orc.TensorInit("normal", 123)
    A = gen.fill(64)
    bias = gen.fill(4)
    dump("xsmm_fusion_seed123", {
        "source": [T + "xsmm-fusion.mlir:9-57 (dispatch tuple :51, invoke :52, RESULT :54-57)",
                   "lib/TPP/Transforms/Utils/TensorInitFloat.cpp:85-95 (input stream)"],
        "buffers": {"A": buf(F32, data=A), "B": buf(F32, size=64, const=2), "C": buf(F32, size=16, const=0),
                    "bias": buf(F32, data=bias)},
        "calls": [fused_call([F32, 4, 4, 8, 8, 4, 4, 32, 32, 4, 0, 5, 4, 1],
                             ["A", 0], ["B", 0], ["C", 0], ["bias", 0], 2)],
        "expect": [expect("C", 4, 4, 4, values=check_numbers(t, prefix="RESULT", count=16))]})

    # ---- seeded: transpose-fp32 (arg0 3x5 then arg1 5x3 from one stream) ----------
    t = read(T + "transpose-fp32.mlir")
    gen = orc.TensorInit("normal", 123)
    x = gen.fill(15)
    y = gen.fill(15)
    dump("transpose_fp32_seed123", {
        "source": [T + "transpose-fp32.mlir:1-25"],
        "buffers": {"X": buf(F32, data=x), "O": buf(F32, data=y)},
        "calls": [unary_call([29, F32, 3, 5, 5, 3, 0], ["X", 0], ["O", 0])],
        "expect": [expect("O", 5, 3, 3, values=check_numbers(t, count=15))]})
    # ---- seeded bf16 VNNI-2 pack: transpose-bf16 (arg0 4x4 bf16, out [2][4][2]) ----
    t = read(T + "transpose-bf16.mlir")
    gen = orc.TensorInit("normal", 123)
    x = orc.bf16_to_f32(gen.fill(16, BF16))
    y = orc.bf16_to_f32(gen.fill(16, BF16))
    dump("vnni_pack_bf16_seed123", {
        "source": [T + "transpose-bf16.mlir:1-34 (values printed as exact bf16)"],
        "buffers": {"X": buf(BF16, data=x), "O": buf(BF16, data=y)},
        "calls": [unary_call([28, BF16, 4, 4, 4, 4, 0], ["X", 0], ["O", 0])],
        "expect": [expect("O", 8, 2, 2, values=check_numbers(t, count=16))]})

    # ---- compiler-level tests replayed with their pinned dispatch tuples ----------
    # xsmm-strided-brgemm: dims (i,ii,k,kk,j,jj); brgemm per (i, j) into a zero 2x2
    # tile (ldc = 2), then tile += bias tile of D (binary add with ldo 16).
    t = read(T + "xsmm-strided-brgemm.mlir")
    lits = dense_literals(t)  # D, A, B in file order (C is a splat)
    calls = []
    for i in range(2):
        for j in range(8):
            tile = "T%d_%d" % (i, j)
            calls.append(brgemm_call([F32, 2, 2, 4, 8, 16, 2, 4, 64, 0], ["A", i * 16], ["B", j * 2], [tile, 0], 2))
            calls.append(binary_call([1, F32, 2, 2, 2, 16, 16, 0], [tile, 0], ["D", i * 32 + j * 2], ["D", i * 32 + j * 2]))
    bufs = {"A": buf(F32, data=lits[1]), "B": buf(F32, data=lits[2]), "D": buf(F32, data=lits[0])}
    for i in range(2):
        for j in range(8):
            bufs["T%d_%d" % (i, j)] = buf(F32, size=4, const=0)
    dump("xsmm_strided_brgemm", {
        "source": [T + "xsmm-strided-brgemm.mlir:18-77 (dispatch tuple :34, CHECK :74-77)"],
        "buffers": bufs, "calls": calls,
        "expect": [expect("D", 4, 16, 16, values=check_numbers(t, count=64))]})

    # xsmm-strided-brgemm1: dims (b,i,h,k,j); A[(b,h),(i,k)] 16x8, B[(b,k),(h,j)] 8x16,
    # C[(b,i),(h,j)] 4x16; gemm [2,2,4, 4,16,16] flags beta_0 (=4) per (b, h).
    t = read(T + "xsmm-strided-brgemm1.mlir")
    lits = dense_literals(t)
    calls = []
    for b in range(2):
        for h in range(8):
            calls.append(gemm_call([F32, 2, 2, 4, 4, 16, 16, 4], ["A", (b * 8 + h) * 8], ["B", b * 64 + h * 2],
                                   ["C", b * 32 + h * 2]))
    dump("xsmm_strided_brgemm1", {
        "source": [T + "xsmm-strided-brgemm1.mlir:7-56 (dispatch tuple :33, CHECK :52-55)"],
        "buffers": {"A": buf(F32, data=lits[0]), "B": buf(F32, data=lits[1]), "C": buf(F32, size=64, const=-5)},
        "calls": calls,
        "expect": [expect("C", 4, 16, 16, values=check_numbers(t, count=64))]})

    # xsmm-strided-brgemm2: A[(b,i),(h,k)] 4x8, B[(b,k),(h,j)] 8x16 as [2,4,2,8],
    # C[(b,h),(i,j)] 4x16 as [2,2,2,8]; gemm [2,8,4, 8,16,8] beta_0 per (b, h).
    t = read(T + "xsmm-strided-brgemm2.mlir")
    lits = dense_literals(t)
    calls = []
    for b in range(2):
        for h in range(2):
            calls.append(gemm_call([F32, 2, 8, 4, 8, 16, 8, 4], ["A", b * 16 + h * 4], ["B", b * 64 + h * 8],
                                   ["C", (b * 2 + h) * 16]))
    dump("xsmm_strided_brgemm2", {
        "source": [T + "xsmm-strided-brgemm2.mlir:7-57 (dispatch tuple :34, CHECK :53-56)"],
        "buffers": {"A": buf(F32, data=lits[0]), "B": buf(F32, data=lits[1]), "C": buf(F32, size=64, const=-5)},
        "calls": calls,
        "expect": [expect("C", 4, 16, 16, values=check_numbers(t, count=64))]})

    # tpp-brgemm (batch 1 -> gemm) and non-unit batch (brgemm), C starts at 0, beta=1
    t = read(T + "tpp-brgemm.mlir")
    lits = dense_literals(t)
    dump("tpp_brgemm", {
        "source": [T + "tpp-brgemm.mlir:11-52"],
        "buffers": {"A": buf(F32, data=lits[0]), "B": buf(F32, data=lits[1]), "C": buf(F32, size=16, const=0)},
        "calls": [gemm_call([F32, 4, 4, 8, 8, 4, 4, 0], ["A", 0], ["B", 0], ["C", 0])],
        "expect": [expect("C", 4, 4, 4, values=check_numbers(t, count=16))]})
    t = read(T + "tpp-brgemm-non-unit-batch.mlir")
    lits = dense_literals(t)
    dump("tpp_brgemm_non_unit_batch", {
        "source": [T + "tpp-brgemm-non-unit-batch.mlir:11-70"],
        "buffers": {"A": buf(F32, data=lits[0]), "B": buf(F32, data=lits[1]), "C": buf(F32, size=16, const=0)},
        "calls": [brgemm_call([F32, 4, 4, 8, 8, 4, 4, 32, 32, 0], ["A", 0], ["B", 0], ["C", 0], 2)],
        "expect": [expect("C", 4, 4, 4, values=check_numbers(t, count=16))]})

    # mlir-gen (10x10x10 on ones: matmul with C init 1 -> 11; fc adds bias 1 -> 12) and
    # mlp-fp32-1layer-512 (128x256x512 + bias + relu on ones -> 257); whole-layer calls.
    dump("mlir_gen_matmul_fc", {
        "source": [T + "mlir-gen.mlir:14-30"],
        "buffers": {"A": buf(F32, size=100, const=1), "W": buf(F32, size=100, const=1),
                    "C": buf(F32, size=100, const=1), "C2": buf(F32, size=100, const=1),
                    "bias": buf(F32, size=10, const=1)},
        "calls": [gemm_call([F32, 10, 10, 10, 10, 10, 10, 0], ["A", 0], ["W", 0], ["C", 0]),
                  fused_call([F32, 10, 10, 10, 10, 10, 10, 0, 0, 0, 0, 0, 4, 1], ["A", 0], ["W", 0], ["C2", 0], ["bias", 0], 1)],
        "expect": [expect("C", 10, 10, 10, fill=check_fill(read(T + "mlir-gen.mlir"), "GEN-MATMUL"), tol="exact"),
                   expect("C2", 10, 10, 10, fill=check_fill(read(T + "mlir-gen.mlir"), "GEN-FC"), tol="exact")]})
    dump("mlp_fp32_1layer_512", {
        "source": [T + "mlp-fp32-1layer-512.mlir (CHECK 257)"],
        "buffers": {"A": buf(F32, size=128 * 256, const=1), "W": buf(F32, size=256 * 512, const=1),
                    "C": buf(F32, size=128 * 512, const=0), "bias": buf(F32, size=512, const=1)},
        "calls": [fused_call([F32, 128, 512, 32, 256, 512, 512, 32, 32 * 512, 4, 0, 5, 4, 1],
                             ["A", 0], ["W", 0], ["C", 0], ["bias", 0], 8)],
        "expect": [expect("C", 128, 512, 512, fill=check_fill(read(T + "mlp-fp32-1layer-512.mlir")), tol="exact")]})

    # ---- BASELINE config 1 ("plumbing"): the call script the default pipeline produces for
    # `mlir-gen --kernel=args --float-type=f32 --batch=256 --layers=256,256`:
    # A / W / C relayout into 32x32 blocks ([8][8][32][32], pack-matmul default tiles,
    # ToBlockLayoutAndBack.cpp:460-471; blocks moved by 2-D copies, LowerPacksAndUnpacks.cpp:45-49),
    # ONE brgemm dispatch [32,32,32,32,32,32,1024,1024], 8x8 invokes with batch 8, un-pack of C.
    # Inputs const 1.0 (seed 0), C argument initialised 1.0, beta = 1 -> every element 256 + 1 = 257
    # (same closed form as test/Integration/mlir-gen.mlir:17,28 where 10*1 + 1 = 11).
    calls = []
    ident = [1, F32, 32, 32, 256, 32, 0]      # xsmm.unary identity: 32x32 block, ldi 256 -> ldo 32
    for bi in range(8):
        for bj in range(8):
            blk = (bi * 8 + bj) * 1024
            calls.append(unary_call(ident, ["A", bi * 32 * 256 + bj * 32], ["Ap", blk]))     # A  [M/32][K/32][32][32]
            calls.append(unary_call(ident, ["W", bj * 32 * 256 + bi * 32], ["Wp", blk]))     # W  [N/32][K/32][32 k][32 n]
            calls.append(unary_call(ident, ["C", bi * 32 * 256 + bj * 32], ["Cp", blk]))     # C  [M/32][N/32][32][32]
    for bi in range(8):
        for bj in range(8):
            calls.append(brgemm_call([F32, 32, 32, 32, 32, 32, 32, 1024, 1024, 0],
                                     ["Ap", bi * 8 * 1024], ["Wp", bj * 8 * 1024], ["Cp", (bi * 8 + bj) * 1024], 8))
    unpack = [1, F32, 32, 32, 32, 256, 0]
    for bi in range(8):
        for bj in range(8):
            calls.append(unary_call(unpack, ["Cp", (bi * 8 + bj) * 1024], ["C", bi * 32 * 256 + bj * 32]))
    dump("c1_mlir_gen_matmul_256", {
        "source": ["BASELINE.json configs[0]; SURVEY.md section 8d (C1)", T + "mlir-gen.mlir:14-30 (closed form)",
                   "lib/TPP/Transforms/ToBlockLayoutAndBack.cpp:460-471 (32x32x32 blocks)"],
        "buffers": {"A": buf(F32, size=65536, const=1), "W": buf(F32, size=65536, const=1),
                    "C": buf(F32, size=65536, const=1), "Ap": buf(F32, size=65536, const=0),
                    "Wp": buf(F32, size=65536, const=0), "Cp": buf(F32, size=65536, const=0)},
        "calls": calls,
        "expect": [expect("C", 256, 256, 256, fill=257, tol="exact")]})

    # ---- bf16 -------------------------------------------------------------------
    VB = 2048  # wire value of dialect vnni_b (ConvertXsmmToFunc.cpp:251-265)
    dump("xsmm_brgemm_bf16", {
        "source": [TB + "xsmm-brgemm-bf16.mlir:5-20"],
        "buffers": {"A": buf(BF16, size=72, const=1), "B": buf(BF16, size=72, const=3), "C": buf(BF16, size=36, const=1)},
        "calls": [brgemm_call([BF16, 6, 6, 6, 6, 6, 6, 36, 36, VB], ["A", 0], ["B", 0], ["C", 0], 2)],
        "expect": [expect("C", 6, 6, 6, fill=check_fill(read(TB + "xsmm-brgemm-bf16.mlir")), tol="exact")]})
    dump("xsmm_gemm_bf16", {
        "source": [TB + "xsmm-gemm-bf16.mlir:5-17"],
        "buffers": {"A": buf(BF16, size=36, const=1), "B": buf(BF16, size=36, const=3), "C": buf(BF16, size=36, const=1)},
        "calls": [gemm_call([BF16, 6, 6, 6, 6, 6, 6, VB], ["A", 0], ["B", 0], ["C", 0])],
        "expect": [expect("C", 6, 6, 6, fill=check_fill(read(TB + "xsmm-gemm-bf16.mlir")), tol="exact")]})
    dump("xsmm_quarternary_bf16", {
        "source": [TB + "xsmm-quarternary-bf16.mlir:4-14"],
        "buffers": {"A": buf(BF16, size=1024, const=1), "B": buf(BF16, size=1024, const=1),
                    "C": buf(BF16, size=16, const=1), "D": buf(BF16, size=4, const=1)},
        "calls": [fused_call([BF16, 4, 4, 4, 4, 4, 4, 8, 8, VB, 0, 5, 4, 1], ["A", 0], ["B", 0], ["C", 0], ["D", 0], 16)],
        "expect": [expect("C", 4, 4, 4, fill=check_fill(read(TB + "xsmm-quarternary-bf16.mlir")), tol="exact")]})
    # 64*4 + 1 = 257 is a bf16 tie between 256 and 258 -> pins round-to-nearest-EVEN
    dump("xsmm_ternary_bf16", {
        "source": [TB + "xsmm-ternary-bf16.mlir:5-18"],
        "buffers": {"A": buf(BF16, size=1024, const=1), "B": buf(BF16, size=1024, const=1), "C": buf(BF16, size=16, const=1)},
        "calls": [brgemm_call([BF16, 4, 4, 4, 4, 4, 4, 8, 8, VB], ["A", 0], ["B", 0], ["C", 0], 64)],
        "expect": [expect("C", 4, 4, 4, fill=check_fill(read(TB + "xsmm-ternary-bf16.mlir")), tol="exact")]})
    dump("xsmm_unary_relu_bf16", {
        "source": [TB + "xsmm-unary-bf16.mlir:5-14"],
        "buffers": {"X": buf(BF16, size=9, const=1)},
        "calls": [unary_call([5, BF16, 3, 3, 3, 3, 0], ["X", 0], ["X", 0])],
        "expect": [expect("X", 3, 3, 3, fill=check_fill(read(TB + "xsmm-unary-bf16.mlir")), tol="exact")]})
    dump("xsmm_binary_add_bf16", {
        "source": [TB + "xsmm-binary-bf16.mlir:5-18"],
        "buffers": {"L": buf(BF16, size=9, const=1), "R": buf(BF16, size=9, const=1), "O": buf(BF16, size=9, const=1)},
        "calls": [binary_call([1, BF16, 3, 3, 3, 3, 3, 0], ["L", 0], ["R", 0], ["O", 0])],
        "expect": [expect("O", 3, 3, 3, fill=check_fill(read(TB + "xsmm-binary-bf16.mlir")), tol="exact")]})
    dump("xsmm_zero_bf16", {
        "source": [TB + "xsmm-zero-bf16.mlir:5-17"],
        "buffers": {"X": buf(BF16, size=9, const=5)},
        "calls": [unary_call([2, BF16, 3, 3, 3, 3, 0], ["X", 0], ["X", 0])],
        "expect": [expect("X", 3, 3, 3, fill=check_fill(read(TB + "xsmm-zero-bf16.mlir")), tol="exact")]})
    # vnni-packing: 16x16 -> [8][16][2]; CHECK pins the first pairs (1,17),(2,18),(3,19);
    # the full expected image is the definition out[i/2][j][i%2] = in[i][j] applied to
    # the literal input (VNNIUtils.cpp:75-77), all values exactly representable in bf16.
    t = read(TB + "vnni-packing.mlir")
    a = dense_literals(t)[0]
    src = np.array(a, dtype=np.float32).reshape(16, 16)
    packed = src.reshape(8, 2, 16).transpose(0, 2, 1).reshape(-1)
    head = check_numbers(t, count=6)
    assert [float(v) for v in packed[:6]] == head, (packed[:6], head)
    dump("vnni_packing", {
        "source": [TB + "vnni-packing.mlir:5-38"],
        "buffers": {"X": buf(BF16, data=a), "O": buf(BF16, size=256, const=0)},
        "calls": [unary_call([28, BF16, 16, 16, 16, 16, 0], ["X", 0], ["O", 0])],
        "expect": [expect("O", 8, 32, 32, values=[float(v) for v in packed], tol="exact")]})
    # mlir-gen-bf16: 16x16x16 on ones -> 17 (matmul, C init 1) / 18 (fc)
    dump("mlir_gen_bf16", {
        "source": [TB + "mlir-gen-bf16.mlir:24-26"],
        "buffers": {"A": buf(BF16, size=256, const=1), "W": buf(BF16, size=256, const=1),
                    "C": buf(BF16, size=256, const=1), "C2": buf(BF16, size=256, const=1),
                    "bias": buf(BF16, size=16, const=1)},
        "calls": [gemm_call([BF16, 16, 16, 16, 16, 16, 16, VB], ["A", 0], ["W", 0], ["C", 0]),
                  fused_call([BF16, 16, 16, 16, 16, 16, 16, 0, 0, VB, 0, 0, 4, 1], ["A", 0], ["W", 0], ["C2", 0], ["bias", 0], 1)],
        "expect": [expect("C", 16, 16, 16, fill=check_fill(read(TB + "mlir-gen-bf16.mlir"), "GEN-MATMUL-BF16"), tol="exact"),
                   expect("C2", 16, 16, 16, fill=check_fill(read(TB + "mlir-gen-bf16.mlir"), "GEN-FC-BF16"), tol="exact")]})

    # ---- round 2: the remaining replayable tests ------------------------------------------------
    # xsmm-strided-brgemm3: dims (b,i,h,k,j); A[(b,i),(h,k)] 4x8 as [2,2,2,4], B[(b,j),(h,k)] 16x8 as [2,8,2,4],
    # C[(b,h),(j,i)] 4x16 as [2,2,8,2]. Per (b, h): ONE transpose dispatch (IR-COUNT-1 xsmm_unary_dispatch) turns
    # A[b, :, h, :] (2x4, ldi 8) into a 4x2 tile (the gemm's B operand), then the pinned gemm dispatch
    # (1, 8, 2, 4, 8, 2, 2, 4) = f32 m=8 n=2 k=4 lda=8 ldb=2 ldc=2 beta_0 takes B[b, :, h, :] (8x4, lda 8) as ITS A.
    t = read(T + "xsmm-strided-brgemm3.mlir")
    lits = dense_literals(t)  # A (4x8), B (16x8); C is a splat
    assert len(lits[0]) == 32 and len(lits[1]) == 128, [len(x) for x in lits]
    calls, bufs = [], {"A": buf(F32, data=lits[0]), "B": buf(F32, data=lits[1]), "C": buf(F32, size=64, const=0)}
    for b in range(2):
        for h in range(2):
            tname = "T%d_%d" % (b, h)
            bufs[tname] = buf(F32, size=8, const=0)
            calls.append(unary_call([29, F32, 2, 4, 8, 2, 0], ["A", b * 16 + h * 4], [tname, 0]))
            calls.append(gemm_call([F32, 8, 2, 4, 8, 2, 2, 4], ["B", b * 64 + h * 4], [tname, 0], ["C", (b * 2 + h) * 16]))
    dump("xsmm_strided_brgemm3", {
        "source": [T + "xsmm-strided-brgemm3.mlir:7-57 (gemm dispatch tuple :35, transpose dispatch :33-34, CHECK :54-57)"],
        "buffers": bufs, "calls": calls,
        "expect": [expect("C", 4, 16, 16, values=check_numbers(t, count=64))]})

    # broadcast-transpose (--vector-to-XSMM, seed 123): arg0 (8), arg1 (4x8), arg2 (8x4) from ONE normal stream;
    # arg1 = broadcast of arg0 along rows (identity, bcast_col), arg2 = transpose(arg1); printed: arg2, 8 rows of 4.
    t = read(T + "broadcast-transpose.mlir")
    gen = orc.TensorInit("normal", 123)
    a0, a1, a2 = gen.fill(8), gen.fill(32), gen.fill(32)
    dump("broadcast_transpose_seed123", {
        "source": [T + "broadcast-transpose.mlir:1-40 (BROADCASTTRANSPOSE lines)",
                   "lib/TPP/Conversion/ConvertVectorToXsmm/ConvertVectorToXsmm.cpp:46-61 (identity bcast / transpose unary calls)"],
        "buffers": {"V": buf(F32, data=a0), "M": buf(F32, data=a1), "O": buf(F32, data=a2)},
        "calls": [unary_call([1, F32, 4, 8, 8, 8, 4], ["V", 0], ["M", 0]),
                  unary_call([29, F32, 4, 8, 8, 4, 0], ["M", 0], ["O", 0])],
        "expect": [expect("O", 8, 4, 4, values=check_numbers(t, prefix="BROADCASTTRANSPOSE", count=32))]})

    # tpp-run-xsmm-path: linalg add with outs == second input: binary add, out ALIASES rhs (2x2; 1 + 2 = 3)
    t = read(T + "tpp-run-xsmm-path.mlir")
    dump("tpp_run_xsmm_path", {
        "source": [T + "tpp-run-xsmm-path.mlir:7-40 (IR: xsmm_binary_dispatch / invoke; outs(%arg1) aliases the rhs)"],
        "buffers": {"L": buf(F32, size=4, const=1), "R": buf(F32, size=4, const=2)},
        "calls": [binary_call([1, F32, 2, 2, 2, 2, 2, 0], ["L", 0], ["R", 0], ["R", 0])],
        "expect": [expect("R", 2, 2, 2, fill=check_fill(t), tol="exact")]})

    # vnni-packing-chain: tensor.pack (outer_dims_perm [1,0], 16x16 tiles) as per-block identity copies
    # (LowerPacksAndUnpacks.cpp:45-49) into [2][2][16][16], then the VNNI-2 pack of every block (IR: xsmm_unary_invoke)
    # into [2][2][8][16][2]; expected image = the test's %G literal, threshold 0.
    t = read(TB + "vnni-packing-chain.mlir")
    lits = dense_literals(t)
    assert len(lits[0]) == 1024 and len(lits[1]) == 1024
    calls = []
    for cb in range(2):       # outer_dims_perm = [1, 0]: block (cb, rb) <- rows 16 rb .., columns 16 cb ..
        for rb in range(2):
            blk = (cb * 2 + rb) * 256
            calls.append(unary_call([1, BF16, 16, 16, 32, 16, 0], ["X", rb * 16 * 32 + cb * 16], ["P", blk]))
            calls.append(unary_call([28, BF16, 16, 16, 16, 16, 0], ["P", blk], ["G", blk]))
    dump("vnni_packing_chain", {
        "source": [TB + "vnni-packing-chain.mlir:7-60 (check.expect_almost_eq against %G, threshold 0.0)"],
        "buffers": {"X": buf(BF16, data=lits[0]), "P": buf(BF16, size=1024, const=0), "G": buf(BF16, size=1024, const=0)},
        "calls": calls,
        "expect": [expect("G", 32, 32, 32, values=lits[1], tol="exact")]})

    # conv-to-matmul: conv_2d_nhwc_hwcf rewritten to one matmul per output row and filter tap
    # (RewriteConvsToMatmulOrBrgemm.cpp; IR: linalg.matmul x3): A = image rows [Q x C] with row stride
    # conv_stride * C, B = filter tap [C x K], C = output row [Q x K], accumulating (beta = 1). Inputs are the
    # test's generate_1D_source broadcasts: image value = channel index, filter value = output channel index,
    # output initialised with the output channel index.
    t = read(T + "conv-to-matmul.mlir")
    outs = re.split(r"vector\.print", t)
    conv_cases = [("conv_unit_no_stride", 4, 4, 3, 8, 1, 1, 1, 0), ("conv_3x3_no_stride", 5, 5, 3, 8, 3, 3, 1, 1),
                  ("conv_3x3_stride2", 5, 5, 3, 8, 3, 3, 2, 2)]
    for name, H, W, Ci, K, R, S, st, idx in conv_cases:
        P, Q = (H - R) // st + 1, (W - S) // st + 1
        img = [float(c) for _ in range(H * W) for c in range(Ci)]
        flt = [float(f) for _ in range(R * S * Ci) for f in range(K)]
        out0 = [float(f) for _ in range(P * Q) for f in range(K)]
        want = [float(x) for x in re.findall(NUM, "".join(l.split(":", 1)[1] for l in outs[idx].splitlines()
                                                           if re.match(r"\s*//\s*CHECK(-SAME)?:", l)))]
        assert len(want) == P * Q * K, (name, len(want))
        calls = []
        for p_ in range(P):
            for r in range(R):
                for s_ in range(S):
                    calls.append(gemm_call([F32, Q, K, Ci, st * Ci, K, K, 0], ["I", ((p_ * st + r) * W + s_) * Ci],
                                           ["F", (r * S + s_) * Ci * K], ["O", p_ * Q * K]))
        dump(name, {
            "source": [T + "conv-to-matmul.mlir (function @%s, CHECK block %d)" % (name if idx else "conv_unit_no_stride", idx + 1),
                       "lib/TPP/Transforms/RewriteConvsToMatmulOrBrgemm.cpp (matmul per output row and filter tap)"],
            "buffers": {"I": buf(F32, data=img), "F": buf(F32, data=flt), "O": buf(F32, data=out0)},
            "calls": calls,
            "expect": [expect("O", P * Q, K, K, values=want, tol="exact")]})

    # ---- ABI wire order: the argument tuples FileCheck pins in test/Conversion/XsmmToFunc/xsmm-to-func.mlir ----
    # (symbol, positional i64 arguments the compiler emits). Asserted against tpp-mlir_amd/runtime.py's argtypes
    # and the parameter order of include/tpp_xsmm_abi.h by tests/test_abi_symbols.py; the dispatch tuples are
    # dispatched through the ABI on the GPU box (every one of them must be accepted).
    t = read("test/Conversion/XsmmToFunc/xsmm-to-func.mlir")
    wire = []
    for block in t.split("// -----"):
        consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"%\[\[(\w+):\.\+\]\] = arith\.constant (-?\d+) : i64", block)}
        # the dialect-level op the tuple was lowered from (kind, dims, flag names, data type)
        om = re.search(r"xsmm\.(\w+)\.dispatch\s+(\w+\s+)?\[([\d,\s]+)\]\s*(\[[\w,\s]+\])?(.*?)data_type\s*=\s*(\w+)", block, re.S)
        for m in re.finditer(r"call @(xsmm_\w+_dispatch)\((.*?)\)\s*$", block, re.M):
            names = re.findall(r"%\[\[(\w+)\]\]", m.group(2))
            if names and all(nm in consts for nm in names) and om:
                flags = {k: [f.strip() for f in v.split(",")] for k, v in re.findall(r"(\w*flags)\s*=\s*\(([^)]*)\)", om.group(5))}
                wire.append({"symbol": m.group(1), "args": [consts[nm] for nm in names],
                             "op": {"family": om.group(1), "kind": (om.group(2) or "").strip() or None,
                                    "dims": [int(x) for x in om.group(3).split(",")],
                                    "fused": [x.strip() for x in om.group(4).strip("[]").split(",")] if om.group(4) else None,
                                    "flags": flags, "data_type": om.group(6)}})
    invokes = [{"symbol": "xsmm_brgemm_invoke", "pattern": ["dtype", "handle", "ptr", "off", "ptr", "off", "ptr", "off", "batch"], "line": 144},
               {"symbol": "xsmm_unary_invoke", "pattern": ["dtype", "handle", "ptr", "off", "ptr", "off"], "line": 165},
               {"symbol": "xsmm_gemm_invoke", "pattern": ["dtype", "handle", "ptr", "off", "ptr", "off", "ptr", "off"], "line": 190}]
    assert len(wire) >= 12, len(wire)
    with open(os.path.join(HERE, "xsmm_to_func_wire.json"), "w") as f:
        json.dump({"source": ["test/Conversion/XsmmToFunc/xsmm-to-func.mlir (CHECK: call @xsmm_*_dispatch lines; invoke lines 144, 165, 190, 224)"],
                   "dispatch": wire, "invoke": invokes}, f)
        f.write("\n")
    print("wrote tests/golden/xsmm_to_func_wire.json (%d dispatch tuples)" % len(wire))

    # ---- FLOP arithmetic of mlir-gen (BENCH_TOTAL_FLOPS), MLIRGen.cpp:313-334 --------
    flops = {"source": ["test/Integration/mlir-gen-flops.mlir:93-112", "tools/mlir-gen/MLIRGen.cpp:313-334"],
             "cases": [
                 {"batch": 256, "layers": [1024, 1024, 1024, 1024], "bias": False, "relu": False, "flops": 1610612736},
                 {"batch": 256, "layers": [1024, 1024, 1024, 1024], "bias": True, "relu": True, "flops": 1612185600},
                 {"batch": 512, "layers": [1024, 1024], "bias": False, "relu": False, "flops": 1073741824},
                 {"batch": 512, "layers": [1024, 1024], "bias": True, "relu": True, "flops": 1074790400},
                 {"batch": 256, "layers": [1024, 1024], "bias": True, "relu": True, "flops": 537395200}]}
    with open(os.path.join(HERE, "flops.json"), "w") as f:
        json.dump(flops, f)
        f.write("\n")


if __name__ == "__main__":
    main()
