// rt_registry.h - dispatch side: hash-consed descriptors, validation like the dialect verifier + libxsmm (XsmmRunnerUtils.cpp:95-246, 308-457)
// One of the subsystem units of runtime.cpp (round 6, VERDICT r5 next 7: the 3 000-line file split by subsystem, no behaviour
// change). The units are INCLUDED into the one translation unit runtime.cpp, in dependence order, inside its anonymous namespace:
// the per-invoke host path (14-18 ns: enqueue_item -> join_window -> Segment::mark) crosses four of them and is inlined across
// their borders - as separate objects without LTO it would pay a call per border. Not a stand-alone header: include runtime.cpp's way only.

// ---- handle registry: hash-cons descriptors by their dispatch tuple -----------------
std::mutex g_mu;
std::map<std::vector<int64_t>, void *> g_registry;

template <typename Make> void *intern(const std::vector<int64_t> &key, Make make) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_registry.find(key);
  if (it != g_registry.end()) return it->second;
  void *p = make();
  g_registry.emplace(key, p);
  return p;
}

size_t esize(int64_t dtype) { return dtype == DT_F32 ? 4 : 2; }

void check_dtype(const char *who, int64_t dtype) {
  if (dtype != DT_F32 && dtype != DT_BF16) die("%s: unhandled data type %ld", who, (long)dtype);
}


template <typename D> const D *as_desc(int64_t handle, int kind, const char *who) {
  const D *d = reinterpret_cast<const D *>(handle);
  if (!d || d->kind != kind) die("%s: handle %ld was not produced by the matching dispatch", who, (long)handle);
  return d;
}

size_t span(int64_t rows, int64_t ld, int64_t cols) { // elements of a rows x cols view
  return rows <= 0 || cols <= 0 ? 0 : (size_t)((rows - 1) * ld + cols);
}

int64_t gemm_dispatch_common(const char *who, int has_batch, int fused, int64_t dtype, int64_t m, int64_t n,
                             int64_t k, int64_t lda, int64_t ldb, int64_t ldc, int64_t stride_a,
                             int64_t stride_b, int64_t flags, int64_t unary_flags, int64_t unary_kind,
                             int64_t binary_flags, int64_t binary_kind) {
  check_dtype(who, dtype);
  if (m < 0 || n < 0 || k < 0 || lda < 0 || ldb < 0 || ldc < 0 || stride_a < 0 || stride_b < 0)
    die("%s: negative dimension (m %ld n %ld k %ld lda %ld ldb %ld ldc %ld)", who, (long)m, (long)n, (long)k,
        (long)lda, (long)ldb, (long)ldc);
  // XsmmOps.cpp:335-340: lda >= k, ldb >= n, ldc >= n
  if (lda < k || ldb < n || ldc < n)
    die("%s: failed to generate func: expect lda >= k, ldb >= n, ldc >= n (M: %ld N: %ld K: %ld lda: %ld ldb: %ld ldc: %ld)",
        who, (long)m, (long)n, (long)k, (long)lda, (long)ldb, (long)ldc);
  const int64_t known = XSMM_GEMM_FLAG_BETA_0 | XSMM_GEMM_FLAG_NO_RESET_TILECONFIG |
                        XSMM_GEMM_FLAG_NO_SETUP_TILECONFIG | XSMM_GEMM_WIRE_VNNI_B | XSMM_GEMM_WIRE_VNNI_A |
                        XSMM_GEMM_FLAG_VNNI_C;
  if (flags & ~known) die("%s: unsupported gemm flags %ld", who, (long)flags);
  const bool vnni_b = (flags & XSMM_GEMM_WIRE_VNNI_B) != 0;
  // wire 4096 = dialect vnni_a: A is [m][k/2][2] (VNNIUtils.cpp:75-77), byte-identical to row-major [m][k]: accepted,
  // nothing to do. wire 8192 = vnni_c: C is stored (and, without BETA_0, read) as VNNI-2 [m/2][n][2].
  const bool vnni_c = (flags & XSMM_GEMM_FLAG_VNNI_C) != 0;
  if ((flags & (XSMM_GEMM_WIRE_VNNI_B | XSMM_GEMM_WIRE_VNNI_A | XSMM_GEMM_FLAG_VNNI_C)) && dtype != DT_BF16)
    die("%s: VNNI flags require bf16 (XsmmOps.cpp:292-298)", who);
  // The blocking factor of a VNNI B operand is not on the wire: the reference's compiler and its runtime library both ask
  // libxsmm_cpuid_dot_pack_factor (VNNIUtils.cpp:25-45; `--vnni=4` in benchmarks/config/omp/mlir-bf16.json:68-100). Its stand-in
  // here is a process-wide setting read at dispatch time (xsmm_hip_set_vnni_factor / TPP_HIP_VNNI_FACTOR, default 2).
  //  The setting is read ONCE per dispatch, here; the handle keeps the factor it was dispatched with (it is part of the
  // descriptor key). A harness sets it before it dispatches - a thread that changes it while another one dispatches gets whichever
  // value is current; with TPP_HIP_TRACE a change between two VNNI dispatches is reported.
  const int vf_now = cfg().vnni_factor.load(std::memory_order_relaxed);
  const int vf = vnni_b ? vf_now : 2;
  if (vnni_b && (k % vf)) die("%s: VNNI-%d B operand needs k to be a multiple of %d, got %ld", who, vf, vf, (long)k);
  // a VNNI A operand [m][k/v][v] is byte-identical to the flat row for every v that divides k: the same factor as B's
  if ((flags & XSMM_GEMM_WIRE_VNNI_A) && (k % vf_now)) die("%s: VNNI-%d A operand needs k to be a multiple of %d, got %ld", who, vf_now, vf_now, (long)k);
  if (flags & (XSMM_GEMM_WIRE_VNNI_B | XSMM_GEMM_WIRE_VNNI_A)) {
    static std::atomic<int> last_vf{0};
    const int prev = last_vf.exchange(vf_now, std::memory_order_relaxed);
    if (prev && prev != vf_now && cfg().trace)
      fprintf(stderr, "[tpp-xsmm-hip] %s: the VNNI factor changed from %d to %d between two VNNI dispatches (handles keep the factor they were "
                      "dispatched with)\n", who, prev, vf_now);
  }
  if (vnni_c && (m & 1)) die("%s: VNNI-2 C operand needs an even m, got %ld", who, (long)m);
  if (fused) {
    if (unary_flags != 0) die("%s: unsupported unary flags %ld on a fused brgemm", who, (long)unary_flags);
    if (unary_kind != XSMM_UNARY_NONE && unary_kind != XSMM_UNARY_RELU)
      die("%s: unsupported fused unary kind %ld (only none/relu reach the runtime)", who, (long)unary_kind);
    // ConvertXsmmToFunc.cpp:405-421: fused ADD is only lowered with bcast_col_in0
    if (binary_kind == XSMM_BINARY_NONE) {
      if (binary_flags != 0) die("%s: binary flags %ld without a binary op", who, (long)binary_flags);
    } else if (!(binary_kind == XSMM_BINARY_ADD && binary_flags == XSMM_BINARY_FLAG_BCAST_COL_IN_0)) {
      die("%s: unsupported fused binary op %ld with flags %ld (only add + bcast_col_in0)", who, (long)binary_kind,
          (long)binary_flags);
    }
  }
  std::vector<int64_t> key = {KIND_GEMM, has_batch, fused, dtype, m, n, k, lda, ldb, ldc, stride_a, stride_b,
                              flags & (XSMM_GEMM_FLAG_BETA_0 | XSMM_GEMM_WIRE_VNNI_B | XSMM_GEMM_FLAG_VNNI_C), unary_kind, binary_kind,
                              cfg().forced_variant.load(), vf};
  void *h = intern(key, [&]() {
    GemmDesc *d = new GemmDesc();
    memset(d, 0, sizeof(*d));
    d->kind = KIND_GEMM;
    d->has_batch = has_batch;
    d->fused = fused;
    d->dtype = dtype; d->m = m; d->n = n; d->k = k; d->lda = lda; d->ldb = ldb; d->ldc = ldc;
    d->stride_a = stride_a; d->stride_b = stride_b; d->wire_flags = flags;
    d->beta0 = (flags & XSMM_GEMM_FLAG_BETA_0) != 0;
    d->vnni_b = vnni_b;
    d->vnni_c = vnni_c;
    d->vnni_factor = vf;
    d->bias = fused && binary_kind == XSMM_BINARY_ADD;
    d->relu = fused && unary_kind == XSMM_UNARY_RELU;
    plan_gemm(*d, cfg().forced_variant.load());
    snprintf(d->trace, sizeof(d->trace), "%s[%ld,%ld,%ld,%ld,%ld,%ld,%ld,%ld] dt%ld flags%ld %s", fused ? "fused_brgemm" : has_batch ? "brgemm" : "gemm",
             (long)m, (long)n, (long)k, (long)lda, (long)ldb, (long)ldc, (long)stride_a, (long)stride_b, (long)dtype, (long)flags, d->name);
    if (cfg().trace)
      fprintf(stderr, "[tpp-xsmm-hip] %s dtype %ld m %ld n %ld k %ld lda %ld ldb %ld ldc %ld sa %ld sb %ld flags %ld -> %s\n",
              who, (long)dtype, (long)m, (long)n, (long)k, (long)lda, (long)ldb, (long)ldc, (long)stride_a,
              (long)stride_b, (long)flags, d->name);
    return (void *)d;
  });
  return reinterpret_cast<int64_t>(h);
}
