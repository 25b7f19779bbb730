// rt_operands.h - the operands (footprints) of one invoke from its descriptor and pointers
// One of the subsystem units of runtime.cpp (round 6, VERDICT r5 next 7: the 3 000-line file split by subsystem, no behaviour
// change). The units are INCLUDED into the one translation unit runtime.cpp, in dependence order, inside its anonymous namespace:
// the per-invoke host path (14-18 ns: enqueue_item -> join_window -> Segment::mark) crosses four of them and is inlined across
// their borders - as separate objects without LTO it would pay a call per border. Not a stand-alone header: include runtime.cpp's way only.

// affinity mask of the thread that loaded the library (normally the main thread, before any OpenMP pinning)
cpu_set_t g_process_mask;
const bool g_have_process_mask = sched_getaffinity(0, sizeof(g_process_mask), &g_process_mask) == 0;

// ---- operands of one invoke, from its descriptor and the element-offset-applied pointers ----------------
// (shared by the invoke entry points and by the scheduler thread, which receives only descriptor + pointers)
struct QueuedOps {
  Operand op[4];
  int n_in;      // op[0 .. n_in) are read
  int out;       // index of the written operand
  bool vec_ok, out_ok, pair_ok; // 16-byte input pieces / 16-byte output pieces + 8-byte bias / even batch count (launch_gemm_grouped)
  QueuedOps() {} // members are filled by queued_operands (no zero-fill on the enqueue path)
};
__attribute__((always_inline)) inline void set_operand(Operand &o, void *ptr, size_t bytes, bool written) {
  o.ptr = ptr;
  o.bytes = bytes;
  o.written = written;
  o.dev = nullptr;
  o.rows = o.row_bytes = o.pitch = 0;
  o.read = true;
  o.host = false;
}
__attribute__((always_inline)) inline void gemm_operands(const GemmDesc *d, void *a, void *b, void *c, void *dp, int64_t br, Operand &A, Operand &B,
                          Operand &C, Operand &D) {
  const size_t es = esize(d->dtype);
  set_operand(A, a, 0, false);
  set_operand(B, b, 0, false);
  set_operand(C, c, (d->vnni_c ? span(d->m / 2, 2 * d->ldc, 2 * d->n) : span(d->m, d->ldc, d->n)) * es, true);
  set_operand(D, dp, d->bias ? (size_t)d->n * es : 0, false);
  if (d->vnni_c) C.shape(d->m / 2, (size_t)2 * d->n * es, (size_t)2 * d->ldc * es);
  else C.shape(d->m, (size_t)d->n * es, (size_t)d->ldc * es);
  if (br > 0 && d->k > 0) {
    A.bytes = ((size_t)(br - 1) * d->stride_a + span(d->m, d->lda, d->k)) * es;
    const int64_t vf = d->vnni_factor;
    const size_t bspan = d->vnni_b ? span((d->k + vf - 1) / vf, vf * d->ldb, vf * d->n) : d->b_trans ? span(d->n, d->ldb, d->k) : span(d->k, d->ldb, d->n);
    B.bytes = ((size_t)(br - 1) * d->stride_b + bspan) * es;
  }
}
// in == nullptr: scalar input or a ZERO op (nothing is read)
inline void unary_operands(const UnaryDesc *d, void *in, void *out, Operand &I, Operand &O) {
  const size_t es = esize(d->dtype);
  set_operand(I, nullptr, 0, false);
  set_operand(O, out, 0, true);
  if (d->op == XSMM_UNARY_TRANSPOSE) {
    O.bytes = span(d->n, d->ldo, d->m) * es;
    O.shape(d->n, (size_t)d->m * es, (size_t)d->ldo * es);
  } else if (d->op == XSMM_UNARY_VNNI2) {
    O.bytes = span(d->m / 2, 2 * d->ldo, 2 * d->n) * es;
    O.shape(d->m / 2, (size_t)2 * d->n * es, (size_t)2 * d->ldo * es);
  } else {
    O.bytes = span(d->m, d->ldo, d->n) * es;
    O.shape(d->m, (size_t)d->n * es, (size_t)d->ldo * es);
  }
  if (in && d->op != XSMM_UNARY_ZERO) {
    I.ptr = in;
    if (d->flags & XSMM_UNARY_FLAG_BCAST_SCALAR) I.bytes = es;
    else if (d->flags & XSMM_UNARY_FLAG_BCAST_ROW) I.bytes = span(d->m, d->ldi, 1) * es;
    else if (d->flags & XSMM_UNARY_FLAG_BCAST_COL) I.bytes = (size_t)d->n * es;
    else {
      I.bytes = span(d->m, d->ldi, d->n) * es;
      I.shape(d->m, (size_t)d->n * es, (size_t)d->ldi * es);
    }
  }
}
inline void binary_operands(const BinaryDesc *d, void *lhs, void *rhs, void *out, Operand &L, Operand &R, Operand &O) {
  const size_t es = esize(d->dtype);
  auto in_bytes = [&](int64_t row, int64_t col, int64_t sc, int64_t ld) -> size_t {
    if (d->flags & sc) return es;
    if (d->flags & row) return span(d->m, ld, 1) * es;
    if (d->flags & col) return (size_t)d->n * es;
    return span(d->m, ld, d->n) * es;
  };
  set_operand(L, lhs, in_bytes(1, 4, 16, d->ldi_lhs), false);
  set_operand(R, rhs, in_bytes(2, 8, 32, d->ldi_rhs), false);
  set_operand(O, out, span(d->m, d->ldo, d->n) * es, true);
  O.shape(d->m, (size_t)d->n * es, (size_t)d->ldo * es);
  if (!(d->flags & (1 | 4 | 16))) L.shape(d->m, (size_t)d->n * es, (size_t)d->ldi_lhs * es);
  if (!(d->flags & (2 | 8 | 32))) R.shape(d->m, (size_t)d->n * es, (size_t)d->ldi_rhs * es);
}
// the operands of a queued work item (kind from the descriptor's first field), as the queue's bookkeeping wants them
__attribute__((always_inline)) inline void queued_operands(const void *desc, const WorkItem &w, QueuedOps &q) {
  const int kind = *(const int *)desc;
  if (kind == KIND_GEMM) {
    gemm_operands((const GemmDesc *)desc, (void *)w.A, (void *)w.B, w.C, (void *)w.D, w.br, q.op[0], q.op[1], q.op[3], q.op[2]);
    q.n_in = 3; // A, B, D read; op[3] = C written (and read when the op accumulates - a superset is harmless)
    q.out = 3;
    q.vec_ok = (((uintptr_t)w.A | (uintptr_t)w.B) & 15) == 0;
    // (... and a batch count of at least one: the loader-wave kernels assume a chunk; a group with an empty batch in it - C = epilogue
    // of nothing - takes the generic kernel like a single such invoke does)
    q.out_ok = (((uintptr_t)w.C) & 15) == 0 && (((uintptr_t)w.D) & 7) == 0 && w.br >= 1;
    q.pair_ok = !(w.br & 1);
  } else if (kind == KIND_UNARY) {
    unary_operands((const UnaryDesc *)desc, (void *)w.A, w.C, q.op[0], q.op[1]);
    q.n_in = 1;
    q.out = 1;
    q.vec_ok = q.out_ok = q.pair_ok = true;
  } else {
    binary_operands((const BinaryDesc *)desc, (void *)w.A, (void *)w.B, w.C, q.op[0], q.op[1], q.op[2]);
    q.n_in = 2;
    q.out = 2;
    q.vec_ok = q.out_ok = q.pair_ok = true;
  }
}
