// runtime.cpp - the dispatch/invoke C-ABI of tpp-mlir's runtime/Xsmm, served by
// gfx950 HIP kernels. Mirrors /root/reference/runtime/Xsmm/XsmmRunnerUtils.cpp entry
// point by entry point (same names, argument order and error behaviour:
// message on stderr + exit(-1)); see include/tpp_xsmm_abi.h for the citations.
//
// Differences that follow from running on a discrete GPU (documented in DESIGN.md):
//  * a handle is a pointer to an immutable descriptor (hash-consed, never freed)
//    instead of a JIT'd function pointer;
//  * data pointers are classified per invoke: device memory is used in place; host
//    memory is mirrored (H2D, kernel, D2H) so host callers keep the reference's
//    "results visible on return" contract;
//  * there is NO CPU fallback: without a HIP device every invoke fails loudly.
#include "../../include/tpp_xsmm_abi.h"
#include "xsmm_desc.h"
#include "chain_args.h"
#include "host_cache.h"

#include <dlfcn.h>
#include <linux/futex.h>
#include <linux/membarrier.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <time.h>
#include <unordered_map>
#include <vector>

using namespace tpp;

namespace {
// The subsystems (each file says what it holds; ONE translation unit: see the note at the top of any of them):
#include "rt_core.h"      // die / HIP_OK, Config + environment switches, tracing
#include "rt_mirror.h"    // device-pointer classification, per-invoke host mirror, host residents   (before the registry: Operand)
#include "rt_registry.h"  // descriptors: hash-consing, dispatch-time validation
#include "rt_operands.h"  // operands / footprints of one invoke
#include "rt_launcher.h"  // the launch thread: complete replayed groups are launched off the calling thread
#include "rt_tile_queue.h" // tile queue state: footprints, trace cache (segments), direct window, group bookkeeping
#include "rt_scheduler.h"  // per-caller rings merged by ONE scheduler thread
#include "rt_enqueue.h"    // the ways into the queue, caller state, enqueue_item (the per-invoke host path)
#include "rt_rewrites.h"  // grid merge, deferred transposes
#include "rt_invoke.h"    // strict-mode items, host-cache scope, gemm_invoke_common
#include "rt_chain.h"     // layer chains: launch, probation, journal, re-run

} // namespace

// =============================== dispatch ==========================================
extern "C" int64_t xsmm_gemm_dispatch(int64_t dtype, int64_t m, int64_t n, int64_t k, int64_t lda, int64_t ldb,
                                      int64_t ldc, int64_t flags) {
  return gemm_dispatch_common("xsmm_gemm_dispatch", 0, 0, dtype, m, n, k, lda, ldb, ldc, 0, 0, flags, 0, 0, 0, 0);
}

extern "C" int64_t xsmm_brgemm_dispatch(int64_t dtype, int64_t m, int64_t n, int64_t k, int64_t lda, int64_t ldb,
                                        int64_t ldc, int64_t stride_a, int64_t stride_b, int64_t flags) {
  return gemm_dispatch_common("xsmm_brgemm_dispatch", 1, 0, dtype, m, n, k, lda, ldb, ldc, stride_a, stride_b,
                              flags, 0, 0, 0, 0);
}

extern "C" int64_t xsmm_fused_brgemm_dispatch(int64_t dtype, int64_t m, int64_t n, int64_t k, int64_t lda,
                                              int64_t ldb, int64_t ldc, int64_t stride_a, int64_t stride_b,
                                              int64_t gemm_flags, int64_t unary_flags, int64_t unary_kind,
                                              int64_t binary_flags, int64_t binary_kind) {
  return gemm_dispatch_common("xsmm_fused_brgemm_dispatch", 1, 1, dtype, m, n, k, lda, ldb, ldc, stride_a,
                              stride_b, gemm_flags, unary_flags, unary_kind, binary_flags, binary_kind);
}

extern "C" int64_t xsmm_unary_dispatch(int64_t kind, int64_t dtype, int64_t m, int64_t n, int64_t ldi,
                                       int64_t ldo, int64_t flags) {
  const char *who = "xsmm_unary_dispatch";
  check_dtype(who, dtype);
  if (m < 0 || n < 0 || ldi < 0 || ldo < 0) die("%s: negative dimension", who);
  switch (kind) {
  case XSMM_UNARY_IDENTITY: case XSMM_UNARY_ZERO: case XSMM_UNARY_RELU:
    if (flags != 0 && flags != XSMM_UNARY_FLAG_BCAST_ROW && flags != XSMM_UNARY_FLAG_BCAST_COL &&
        flags != XSMM_UNARY_FLAG_BCAST_SCALAR)
      die("failed to generate unary func\nop_type: %ld\nflags: %ld", (long)kind, (long)flags);
    if (ldo < n) die("%s: ldo %ld < n %ld", who, (long)ldo, (long)n);
    if (flags == 0 && kind != XSMM_UNARY_ZERO && ldi < n) die("%s: ldi %ld < n %ld", who, (long)ldi, (long)n);
    break;
  case XSMM_UNARY_TRANSPOSE: // m, n are the INPUT dims; output is n x m
    if (flags != 0) die("%s: transpose takes no broadcast flags", who);
    if (ldi < n || ldo < m) die("%s: transpose expects ldi >= n and ldo >= m (m %ld n %ld ldi %ld ldo %ld)", who,
                               (long)m, (long)n, (long)ldi, (long)ldo);
    break;
  case XSMM_UNARY_VNNI2:
    if (dtype != DT_BF16) die("%s: VNNI-2 packing is defined for bf16 only", who);
    if (flags != 0) die("%s: vnni_2 takes no broadcast flags", who);
    if (m & 1) die("%s: VNNI-2 packing needs an even number of rows, got %ld", who, (long)m);
    if (ldi < n || ldo < n) die("%s: vnni_2 expects ldi >= n and ldo >= n", who);
    break;
  default:
    die("failed to generate unary func\nop_type: %ld\nflags: %ld", (long)kind, (long)flags);
  }
  std::vector<int64_t> key = {KIND_UNARY, kind, dtype, m, n, ldi, ldo, flags};
  void *h = intern(key, [&]() {
    UnaryDesc *d = new UnaryDesc{KIND_UNARY, kind, dtype, m, n, ldi, ldo, flags, {0}};
    snprintf(d->trace, sizeof(d->trace), "unary kind%ld [%ld,%ld,%ld,%ld] dt%ld flags%ld", (long)kind, (long)m, (long)n, (long)ldi, (long)ldo, (long)dtype, (long)flags);
    if (cfg().trace) fprintf(stderr, "[tpp-xsmm-hip] xsmm_unary_dispatch %s\n", d->trace);
    return (void *)d;
  });
  return reinterpret_cast<int64_t>(h);
}

extern "C" int64_t xsmm_binary_dispatch(int64_t kind, int64_t dtype, int64_t m, int64_t n, int64_t ldi_lhs,
                                        int64_t ldi_rhs, int64_t ldo, int64_t flags) {
  const char *who = "xsmm_binary_dispatch";
  check_dtype(who, dtype);
  if (kind < XSMM_BINARY_ADD || kind > XSMM_BINARY_DIV)
    die("failed to generate binary func\nop_type: %ld\nflags: %ld", (long)kind, (long)flags);
  if (m < 0 || n < 0 || ldi_lhs < 0 || ldi_rhs < 0 || ldo < 0) die("%s: negative dimension", who);
  if (flags & ~int64_t(63)) die("failed to generate binary func\nop_type: %ld\nflags: %ld", (long)kind, (long)flags);
  auto one = [&](int64_t f) { return (f & (f - 1)) == 0; }; // at most one broadcast per operand
  const int64_t f0 = flags & (1 | 4 | 16), f1 = flags & (2 | 8 | 32);
  if (!one(f0) || !one(f1)) die("%s: conflicting broadcast flags %ld", who, (long)flags);
  if (ldo < n) die("%s: ldo %ld < n %ld", who, (long)ldo, (long)n);
  if (f0 == 0 && ldi_lhs < n) die("%s: ldi lhs %ld < n %ld", who, (long)ldi_lhs, (long)n);
  if (f1 == 0 && ldi_rhs < n) die("%s: ldi rhs %ld < n %ld", who, (long)ldi_rhs, (long)n);
  std::vector<int64_t> key = {KIND_BINARY, kind, dtype, m, n, ldi_lhs, ldi_rhs, ldo, flags};
  void *h = intern(key, [&]() {
    BinaryDesc *d = new BinaryDesc{KIND_BINARY, kind, dtype, m, n, ldi_lhs, ldi_rhs, ldo, flags, {0}};
    snprintf(d->trace, sizeof(d->trace), "binary kind%ld [%ld,%ld,%ld,%ld,%ld] dt%ld flags%ld", (long)kind, (long)m, (long)n, (long)ldi_lhs, (long)ldi_rhs, (long)ldo, (long)dtype, (long)flags);
    if (cfg().trace) fprintf(stderr, "[tpp-xsmm-hip] xsmm_binary_dispatch %s\n", d->trace);
    return (void *)d;
  });
  return reinterpret_cast<int64_t>(h);
}

extern "C" int64_t xsmm_intel_amx_tile_config_dispatch(int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,
                                                       int64_t, int64_t, int64_t, int64_t) {
  // AMX tile configuration has no meaning on CDNA4; the compiler emits these calls
  // around every bf16 brgemm (IntelAMXTileConfig.cpp:36-118), so they must exist.
  static AmxDesc amx{KIND_AMX};
  return reinterpret_cast<int64_t>(&amx);
}

// =============================== invoke ============================================
extern "C" void xsmm_gemm_invoke(int64_t dtype, int64_t handle, void *a, int64_t off_a, void *b, int64_t off_b,
                                 void *c, int64_t off_c) {
  gemm_invoke_common("xsmm_gemm_invoke", false, dtype, handle, a, off_a, b, off_b, c, off_c, nullptr, 0, 1);
}

extern "C" void xsmm_brgemm_invoke(int64_t dtype, int64_t handle, void *a, int64_t off_a, void *b, int64_t off_b,
                                   void *c, int64_t off_c, int64_t num_batches) {
  gemm_invoke_common("xsmm_brgemm_invoke", false, dtype, handle, a, off_a, b, off_b, c, off_c, nullptr, 0,
                     num_batches);
}

extern "C" void xsmm_fused_brgemm_invoke(int64_t dtype, int64_t handle, void *a, int64_t off_a, void *b,
                                         int64_t off_b, void *c, int64_t off_c, void *d, int64_t off_d,
                                         int64_t num_batches) {
  gemm_invoke_common("xsmm_fused_brgemm_invoke", true, dtype, handle, a, off_a, b, off_b, c, off_c, d, off_d,
                     num_batches);
}

__attribute__((always_inline)) static inline void unary_invoke_common(const char *who, int64_t dtype, int64_t handle, void *in, int64_t off_in,
                                float scalar, bool use_scalar, void *out, int64_t off_out) {
  const UnaryDesc *d = as_desc<UnaryDesc>(handle, KIND_UNARY, who);
  if (d->dtype != dtype) die("%s: invoke dtype %ld != dispatch dtype %ld", who, (long)dtype, (long)d->dtype);
  if (d->m == 0 || d->n == 0) return;
  TraceRange trace_range(who, d->trace);
  const size_t es = esize(dtype);
  if (use_scalar && (d->op == XSMM_UNARY_TRANSPOSE || d->op == XSMM_UNARY_VNNI2))
    die("%s: scalar input is meaningless for op %ld", who, (long)d->op);
  void *pi = use_scalar || d->op == XSMM_UNARY_ZERO ? nullptr : (char *)in + off_in * es, *po = (char *)out + off_out * es;
  HcScope hcs;
  if (hc_on()) {
    Operand I, O;
    unary_operands(d, pi, po, I, O);
    hcs.add(&pi, I, true, false);
    hcs.add(&po, O, false, true);
    hcs.go(cfg().stream.load(std::memory_order_relaxed));
  }
  // (the steady state of a transpose-then-gemm loop: the record's source is replaced, nothing else - rt_rewrites.h)
  if (d->op == XSMM_UNARY_TRANSPOSE && pi && !hcs.hits && dt_defer_fast(d, pi, po, cfg().stream.load(std::memory_order_relaxed))) return;
  unary_invoke_core(d, pi, scalar, use_scalar, po, true);
}
namespace {
void unary_invoke_core(const UnaryDesc *d, void *pi, float scalar, bool use_scalar, void *po, bool may_defer) {
  hipStream_t s = cfg().stream.load(std::memory_order_relaxed);
  if (may_defer) {
    if (d->op == XSMM_UNARY_TRANSPOSE && pi && dt_defer(d, pi, po, s)) return;
    if (g_dt_pending.load(std::memory_order_acquire)) {
      Operand I, O;
      unary_operands(d, pi, po, I, O);
      const void *rd[1] = {I.ptr};
      const size_t rb[1] = {I.bytes};
      dt_other(rd, rb, 1, po, O.bytes);
    }
  }
  if (cfg().tile_queue.load(std::memory_order_relaxed)) {
    // small tiles of tensor.pack / unpack lowering and bias broadcasts: queued like the GEMM tiles
    if (queue_active() && !use_scalar && d->m <= 64 && d->n <= 64) {
      const void *ptrs[2] = {pi, po};
      if (enqueue_item(d, WorkItem{pi, nullptr, po, nullptr, 0}, ptrs, 2, s)) return;
    }
    flush_tile_queue();
  }
  Operand I, O;
  unary_operands(d, pi, po, I, O);
  O.read = false; // an in-place input is uploaded through I
  std::vector<Operand *> ops = {&I, &O};
  stage_in(ops, s);
  HIP_OK(launch_unary(*d, I.dev, scalar, use_scalar, O.dev, s));
  finish(ops, s);
}
} // namespace

extern "C" void xsmm_unary_invoke(int64_t dtype, int64_t handle, void *in, int64_t off_in, void *out,
                                  int64_t off_out) {
  unary_invoke_common("xsmm_unary_invoke", dtype, handle, in, off_in, 0.0f, false, out, off_out);
}

extern "C" void xsmm_unary_scalar_invoke(int64_t dtype, int64_t handle, float scalar, void *out, int64_t off_out) {
  unary_invoke_common("xsmm_unary_scalar_invoke", dtype, handle, nullptr, 0, scalar, true, out, off_out);
}

extern "C" void xsmm_binary_invoke(int64_t dtype, int64_t handle, void *lhs, int64_t off_lhs, void *rhs,
                                   int64_t off_rhs, void *out, int64_t off_out) {
  const char *who = "xsmm_binary_invoke";
  const BinaryDesc *d = as_desc<BinaryDesc>(handle, KIND_BINARY, who);
  if (d->dtype != dtype) die("%s: invoke dtype %ld != dispatch dtype %ld", who, (long)dtype, (long)d->dtype);
  if (d->m == 0 || d->n == 0) return;
  TraceRange trace_range(who, d->trace);
  const size_t es = esize(dtype);
  void *pl = (char *)lhs + off_lhs * es, *pr = (char *)rhs + off_rhs * es, *po = (char *)out + off_out * es;
  hipStream_t s = cfg().stream.load(std::memory_order_relaxed);
  HcScope hcs;
  if (hc_on()) {
    Operand L, R, O;
    binary_operands(d, pl, pr, po, L, R, O);
    hcs.add(&pl, L, true, false);
    hcs.add(&pr, R, true, false);
    hcs.add(&po, O, false, true);
    hcs.go(s);
  }
  if (g_dt_pending.load(std::memory_order_acquire)) {
    Operand L, R, O;
    binary_operands(d, pl, pr, po, L, R, O);
    const void *rd[2] = {L.ptr, R.ptr};
    const size_t rb[2] = {L.bytes, R.bytes};
    dt_other(rd, rb, 2, po, O.bytes);
  }
  if (cfg().tile_queue.load(std::memory_order_relaxed)) {
    if (queue_active() && d->m <= 64 && d->n <= 64) {
      const void *ptrs[3] = {pl, pr, po};
      if (enqueue_item(d, WorkItem{pl, pr, po, nullptr, 0}, ptrs, 3, s)) return;
    }
    flush_tile_queue();
  }
  Operand L, R, O;
  binary_operands(d, pl, pr, po, L, R, O);
  O.read = false; // out == lhs / rhs is uploaded through that operand
  std::vector<Operand *> ops = {&L, &R, &O};
  stage_in(ops, s);
  HIP_OK(launch_binary(*d, L.dev, R.dev, O.dev, s));
  finish(ops, s);
}

extern "C" void xsmm_intel_amx_tile_config_invoke(int64_t, int64_t, void *, int64_t) {}

// =============================== timers ============================================
// runtime/PerfRunnerUtils.cpp:23-35, plus a device drain so queued launches count.
extern "C" int64_t perf_start_timer(void) {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(
             std::chrono::high_resolution_clock::now().time_since_epoch())
      .count();
}

extern "C" double perf_stop_timer(int64_t start) {
  flush_tile_queue();
  g_devmem_epoch.fetch_add(1, std::memory_order_relaxed);
  if (cfg().async.load()) {
    HIP_OK(hipStreamSynchronize(cfg().stream.load())); // an asynchronous kernel fault must not read as a timing
    check_chain_errors();
    hc::on_sync_point(cfg().stream.load()); // host cache: what the kernels of this region wrote goes back to the host now
  }
  const int64_t now = std::chrono::duration_cast<std::chrono::nanoseconds>(
                          std::chrono::high_resolution_clock::now().time_since_epoch())
                          .count();
  return (double)(now - start) * 1e-9;
}

// =============================== extensions ========================================
extern "C" int xsmm_hip_set_async(int enable) {
  flush_tile_queue();
  const int prev = cfg().async.exchange(enable != 0);
  if (prev && !enable) { // leaving async mode restores "results visible on return" for everything already enqueued
    HIP_OK(hipStreamSynchronize(cfg().stream.load()));
    check_chain_errors();
    hc::on_sync_point(cfg().stream.load());
    g_devmem_epoch.fetch_add(1, std::memory_order_relaxed);
  }
  return prev;
}
extern "C" void xsmm_hip_set_stream(void *s) {
  flush_tile_queue();
  const hipStream_t old = cfg().stream.exchange((hipStream_t)s);
  // xsmm_hip_synchronize / perf_stop_timer / leaving async mode drain the CURRENT stream only, and the header promises that
  // operands may be freed after they return: work enqueued on the stream being left must not outlive that promise
  if (old != (hipStream_t)s && cfg().async.load(std::memory_order_relaxed)) {
    HIP_OK(hipStreamSynchronize(old));
    check_chain_errors(old);
    hc::on_sync_point(old);
  }
}
extern "C" int xsmm_hip_set_tile_queue(int enable) {
  flush_tile_queue();
  const int mode = enable < 0 ? 0 : enable > 2 ? 2 : enable; // 2: several callers always go through the scheduler thread
  const int prev = cfg().tile_queue.exchange(mode);
  if (prev == 2 && mode != 2) {
    // leaving mode 2: back to the inline / direct path (a mode switch happens between bursts - no invoke is in flight - and the
    // flush above has drained the scheduler's rings)
    InlineQueue &iq = inl();
    iq.dw.touch(thread_token());
    std::lock_guard<SpinLock> lk(iq.mu);
    iq.scheduled.store(false, std::memory_order_release);
    iq.owner = 0;
    iq.foreign = 0;
    iq.multi = false;
    iq.slow = 0;
  }
  return prev;
}
extern "C" void xsmm_hip_flush(void) { flush_tile_queue(); }
// n fused_brgemm invokes in one call: exactly the effect of xsmm_fused_brgemm_invoke(dtype, handles[i], ...) for i = 0 .. n-1 in
// order. When the calls form a chain the chip can run as one launch (see include/tpp_xsmm_abi.h) they run as ONE kernel.
extern "C" int xsmm_hip_fused_brgemm_chain_invoke(int64_t dtype, int64_t n, const int64_t *handles, void *const *a, const int64_t *off_a,
                                                  void *const *b, const int64_t *off_b, void *const *c, const int64_t *off_c,
                                                  void *const *d, const int64_t *off_d, const int64_t *num_batches) {
  const char *who = "xsmm_hip_fused_brgemm_chain_invoke";
  if (n <= 0) return 0;
  if (n <= CH_MAXL) {
    const GemmDesc *desc[CH_MAXL];
    void *pa[CH_MAXL], *pb[CH_MAXL], *pc[CH_MAXL], *pd[CH_MAXL];
    bool ok = true;
    const size_t es = esize(dtype);
    for (int64_t i = 0; i < n; ++i) {
      desc[i] = as_desc<GemmDesc>(handles[i], KIND_GEMM, who);
      if (desc[i]->dtype != dtype) die("%s: invoke dtype %ld != dispatch dtype %ld", who, (long)dtype, (long)desc[i]->dtype);
      if (!desc[i]->fused) die("%s: handle %ld was not dispatched by xsmm_fused_brgemm_dispatch", who, (long)i);
      if (num_batches[i] < 0) die("%s: negative batch count %ld", who, (long)num_batches[i]);
      pa[i] = (char *)a[i] + off_a[i] * es;
      pb[i] = (char *)b[i] + off_b[i] * es;
      pc[i] = (char *)c[i] + off_c[i] * es;
      pd[i] = d[i] ? (char *)d[i] + off_d[i] * es : nullptr;
      if (desc[i]->bias && !d[i]) die("%s: fused bias operand of call %ld is null", who, (long)i);
      ok = ok && desc[i]->m > 0 && desc[i]->n > 0;
    }
    if (ok) {
      flush_tile_queue();
      TraceRange trace_range(who, desc[0]->trace);
      if (try_chain_launch((int)n, desc, pa, pb, pc, pd, num_batches, cfg().stream.load(std::memory_order_relaxed))) return 1;
    }
  }
  for (int64_t i = 0; i < n; ++i)
    xsmm_fused_brgemm_invoke(dtype, handles[i], a[i], off_a[i], b[i], off_b[i], c[i], off_c[i], d[i], off_d[i], num_batches[i]);
  return 0;
}
extern "C" int64_t xsmm_hip_chain_status(void) { return g_chain_repairs.load(std::memory_order_relaxed); }
extern "C" void xsmm_hip_tile_queue_stats(int64_t out[5]) {
  out[0] = g_q_launches.load(std::memory_order_relaxed);
  out[1] = g_q_checked.load(std::memory_order_relaxed);
  out[2] = g_q_replayed.load(std::memory_order_relaxed);
  out[3] = g_q_terminated.load(std::memory_order_relaxed);
  out[4] = g_q_abandoned.load(std::memory_order_relaxed);
}
extern "C" void *xsmm_hip_get_stream(void) { return (void *)cfg().stream.load(); }
extern "C" void xsmm_hip_synchronize(void) {
  flush_tile_queue();
  g_devmem_epoch.fetch_add(1, std::memory_order_relaxed);
  HIP_OK(hipStreamSynchronize(cfg().stream.load()));
  check_chain_errors();
  hc::on_sync_point(cfg().stream.load());
}
// ---- host residents (see the comment at Resident) ---------------------------------------------------------
extern "C" int xsmm_hip_host_resident(const void *ptr, int64_t bytes) {
  if (!ptr || bytes <= 0) return -1;
  hipStream_t s = cfg().stream.load();
  std::lock_guard<std::mutex> lk(g_res_mu);
  for (const Resident &r : g_residents)
    if ((const char *)ptr < r.host + r.bytes && r.host < (const char *)ptr + bytes) return -1; // overlaps an existing resident
  Resident r{(char *)ptr, (size_t)bytes, nullptr};
  HIP_OK(hipMalloc((void **)&r.dev, (size_t)bytes + 256));
  r.dev += ((uintptr_t)ptr) & 255; // keep the caller's alignment class (kernel choices depend on it)
  HIP_OK(hipMemcpyAsync(r.dev, ptr, (size_t)bytes, hipMemcpyHostToDevice, s));
  HIP_OK(hipStreamSynchronize(s));
  g_residents.push_back(r);
  g_n_residents.store((int)g_residents.size(), std::memory_order_release);
  return 0;
}
extern "C" int xsmm_hip_host_update(const void *ptr) {
  hipStream_t s = cfg().stream.load();
  std::lock_guard<std::mutex> lk(g_res_mu);
  for (const Resident &r : g_residents)
    if (r.host == (const char *)ptr) {
      HIP_OK(hipMemcpyAsync(r.dev, r.host, r.bytes, hipMemcpyHostToDevice, s));
      HIP_OK(hipStreamSynchronize(s));
      return 0;
    }
  return -1;
}
extern "C" int xsmm_hip_host_release(const void *ptr) {
  flush_tile_queue();
  std::lock_guard<std::mutex> lk(g_res_mu);
  for (size_t i = 0; i < g_residents.size(); ++i)
    if (g_residents[i].host == (const char *)ptr) {
      HIP_OK(hipStreamSynchronize(cfg().stream.load()));
      HIP_OK(hipFree(g_residents[i].dev - (((uintptr_t)ptr) & 255)));
      g_residents.erase(g_residents.begin() + i);
      g_n_residents.store((int)g_residents.size(), std::memory_order_release);
      return 0;
    }
  return -1;
}
// ---- host cache (host_cache.h): host operands kept on the device between invokes, re-uploaded only where the host wrote -----
extern "C" int xsmm_hip_set_host_cache(int enable) {
  (void)hc_on(); // hooks + the environment switch first
  if (!enable && hc::enabled()) { // off: launch what is queued, drain, write everything back, forget the mirrors
    flush_tile_queue();
    HIP_OK(hipStreamSynchronize(cfg().stream.load()));
    check_chain_errors();
    hc::on_sync_point(cfg().stream.load());
    g_devmem_epoch.fetch_add(1, std::memory_order_relaxed);
  }
  return hc::set_enabled(enable != 0);
}
extern "C" void xsmm_hip_host_cache_stats(int64_t out[10]) { hc::stats(out); }
extern "C" int xsmm_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
extern "C" const char *xsmm_hip_kernel_name(int64_t handle) {
  const GemmDesc *d = reinterpret_cast<const GemmDesc *>(handle);
  return (d && d->kind == KIND_GEMM) ? d->name : "";
}
extern "C" const char *xsmm_hip_last_grouped_kernel(void) {
  const char *m = g_last_merged.load(std::memory_order_relaxed);
  return m ? m : last_grouped_kernel();
}
extern "C" const char *xsmm_hip_last_refined_kernel(void) { return last_refined_kernel(); }
extern "C" void xsmm_hip_force_variant(int v) { cfg().forced_variant.store(v); }
// strict mode (see Config::strict). Meant to be chosen before the first invoke (TPP_HIP_STRICT=1): groups recorded by the tile queue's
// trace cache under the other setting would replay on the kernel they were recorded for - a change after the queue has recorded
// a group is refused (-1).
extern "C" int xsmm_hip_set_strict(int enable) {
  flush_tile_queue();
  const int prev = cfg().strict.load();
  if ((enable != 0) == (prev != 0)) return prev;
  {
    InlineQueue &iq = inl();
    iq.dw.touch(thread_token());
    std::lock_guard<SpinLock> lk(iq.mu);
    if (!iq.q.segs.empty()) {
      fprintf(stderr, "[tpp-xsmm-hip] xsmm_hip_set_strict(%d) refused: the tile queue has recorded groups under the other setting (choose the mode "
                      "before the first queued invoke, or with TPP_HIP_STRICT)\n", enable);
      return -1;
    }
    cfg().strict.store(enable != 0);
    tpp::set_strict_kernels(enable != 0);
  }
  return prev;
}
extern "C" int xsmm_hip_get_strict(void) { return cfg().strict.load(); }
// the launch thread (rt_launcher.h): returns the previous setting; out[0] = launches handed over since process start, out[1] = the
// thread exists right now
extern "C" int xsmm_hip_set_launch_thread(int enable) {
  flush_tile_queue(); // (drains it)
  return launcher().on.exchange(enable != 0);
}
extern "C" void xsmm_hip_launch_thread_stats(int64_t out[2]) {
  out[0] = launcher().handed.load(std::memory_order_relaxed);
  out[1] = launcher().running.load(std::memory_order_relaxed) ? 1 : 0;
}
extern "C" int xsmm_hip_set_fold_transpose(int enable) {
  flush_tile_queue(); // (launches a remembered transpose)
  return cfg().fold_transpose.exchange(enable != 0);
}
extern "C" void xsmm_hip_fold_transpose_stats(int64_t out[3]) {
  out[0] = out[1] = 0;
  for (int i = 0; i < DT_SLOTS; ++i) {
    out[0] += g_dt_slots[i].folded.load(std::memory_order_relaxed);  // gemm invokes that read a remembered transpose's source
    out[1] += g_dt_slots[i].dropped.load(std::memory_order_relaxed); // remembered transposes that were overwritten before anything else could read them
  }
  out[2] = g_dt_launched.load(std::memory_order_relaxed); // remembered transposes that were launched after all
}
extern "C" int xsmm_hip_force_split(int v) { return tpp::force_gemm_split(v); }
// the VNNI blocking factor of bf16 B operands dispatched from now on (2 or 4); returns the previous one, -1 for an invalid factor
extern "C" int xsmm_hip_set_vnni_factor(int v) {
  if (v != 2 && v != 4) return -1;
  return cfg().vnni_factor.exchange(v);
}
extern "C" int xsmm_hip_get_vnni_factor(void) { return cfg().vnni_factor.load(); }
extern "C" const char *xsmm_hip_version(void) { return "tpp-xsmm-hip 0.1 (gfx950)"; }
