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

[[noreturn]] void die(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
  fflush(stderr);
  exit(-1); // XsmmRunnerUtils.cpp:132-137 convention
}

#define HIP_OK(expr)                                                                               \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) die("tpp-xsmm-hip: %s failed: %s (no CPU fallback exists)", #expr, hipGetErrorString(e_)); \
  } while (0)

inline void cpu_relax() {
#if defined(__x86_64__)
  __builtin_ia32_pause();
#else
  asm volatile("" ::: "memory");
#endif
}

struct Config {
  std::atomic<int> async{0};
  std::atomic<hipStream_t> stream{nullptr};
  std::atomic<int> forced_variant{-1};
  std::atomic<int> tile_queue{0};
  std::atomic<int> vnni_factor{2}; // blocking factor of VNNI B operands dispatched from now on (xsmm_hip_set_vnni_factor / TPP_HIP_VNNI_FACTOR)
  int trace = 0; // TPP_HIP_TRACE: 1 = one stderr line per dispatch + a roctx range per invoke, 2 = also one stderr line per invoke
  std::atomic<int> fold_transpose{1}; // TPP_HIP_FOLD_TRANSPOSE / xsmm_hip_set_fold_transpose: transposes that feed a gemm's B operand are folded into it
  // TPP_HIP_STRICT / xsmm_hip_set_strict (round 6): the kernel an invoke runs on is a function of its descriptor,
  // batch count and own pointer alignment only - no grid merge, no folded transposes, no kernel family chosen by the size of the
  // queued group, groups of one alignment class and one batch count only. The same invoke on the same data then returns the same
  // bits whether it runs alone, in the first pass of a queued group or in a replay (libxsmm's JIT'd kernel is a function of the
  // dispatch tuple: XsmmRunnerUtils.cpp:288-306).
  std::atomic<int> strict{0};
  Config() {
    if (const char *e = getenv("TPP_HIP_STRICT")) {
      strict = atoi(e) != 0;
      tpp::set_strict_kernels(strict.load());
    }
    if (const char *e = getenv("TPP_HIP_FOLD_TRANSPOSE")) fold_transpose = atoi(e) != 0;
    if (const char *e = getenv("TPP_HIP_ASYNC")) async = atoi(e) != 0;
    if (const char *e = getenv("TPP_HIP_TRACE")) trace = atoi(e);
    if (const char *e = getenv("TPP_HIP_VARIANT")) forced_variant = atoi(e);
    if (const char *e = getenv("TPP_HIP_TILE_QUEUE")) tile_queue = atoi(e) < 0 ? 0 : atoi(e) > 2 ? 2 : atoi(e);
    if (const char *e = getenv("TPP_HIP_VNNI_FACTOR")) {
      if (atoi(e) == 2 || atoi(e) == 4) vnni_factor = atoi(e);
      else fprintf(stderr, "[tpp-xsmm-hip] TPP_HIP_VNNI_FACTOR=%s ignored: the factor is 2 or 4\n", e);
    }
  }
};
Config &cfg() {
  static Config c;
  return c;
}
// the stream an invoke of THIS thread launches on: the process-wide setting, unless the thread is re-running a journaled chain
// launch on that launch's stream (check_chain_errors; ADVICE r5: the re-run must not change the setting other threads read)
thread_local hipStream_t tl_stream_override = nullptr; 
thread_local bool tl_has_stream_override = false;      
inline hipStream_t invoke_stream() { return tl_has_stream_override ? tl_stream_override : cfg().stream.load(std::memory_order_relaxed); }

// ---- tracing: with TPP_HIP_TRACE >= 1 every invoke runs inside a roctx range named after its
// dispatch tuple and kernel, so `rocprofv3 --marker-trace --kernel-trace` timelines show which xsmm call a kernel
// belongs to. libroctx64 is looked up at run time (profiling tool, not a link dependency of the product).
struct Roctx {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    if (cfg().trace < 1) return;
    // rocprofv3 (rocprofiler-sdk) traces the SDK's roctx library; the classic libroctx64 serves older tools
    void *h = nullptr;
    for (const char *name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "/opt/rocm/lib/librocprofiler-sdk-roctx.so",
                             "libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so"})
      if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return;
    push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
    pop = (int (*)())dlsym(h, "roctxRangePop");
    if (!push || !pop) push = nullptr, pop = nullptr;
  }
};
Roctx &roctx() {
  static Roctx r;
  return r;
}
struct TraceRange {
  bool on = false;
  TraceRange(const char *who, const char *what) {
    if (cfg().trace < 1) return;
    if (cfg().trace >= 2) fprintf(stderr, "[tpp-xsmm-hip] %s %s\n", who, what);
    if (roctx().push) on = roctx().push(what) >= 0;
  }
  ~TraceRange() {
    if (on) roctx().pop();
  }
};

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

// ---- device scratch for mirroring host operands (per thread, grow only) -------------
struct Arena {
  char *base = nullptr;
  size_t cap = 0, used = 0;
  char *alloc(size_t bytes) {
    used = (used + 255) & ~size_t(255);
    char *p = base + used;
    used += bytes;
    return p;
  }
  void reserve(size_t bytes, hipStream_t s) {
    used = 0;
    if (bytes <= cap) return;
    if (base) {
      HIP_OK(hipStreamSynchronize(s));
      HIP_OK(hipFree(base));
    }
    cap = std::max(bytes, cap * 2);
    HIP_OK(hipMalloc((void **)&base, cap));
  }
};
thread_local Arena t_arena;

bool is_device_ptr(const void *p) {
  if (!p) return true; // nothing to mirror
  hipPointerAttribute_t attr;
  memset(&attr, 0, sizeof(attr));
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void)hipGetLastError(); // plain malloc'd memory on older runtimes: invalid value
    return false;
  }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged || attr.type == hipMemoryTypeArray;
}

struct Range {
  uintptr_t b, e;
};

// Device allocations seen so far ([base, base+size) from hipMemGetAddressRange), one cache per calling thread.
// Callers issue hundreds of invokes per layer on the same few allocations, and one driver query per operand
// per invoke would dominate the host time (~1 us each). The epoch is bumped at the explicit synchronisation
// points (xsmm_hip_synchronize, perf_stop_timer): the caller may free and re-allocate buffers after those, so
// cached ranges are only trusted within one epoch (and never in synchronous mode, see stage_in).
std::atomic<uint64_t> g_devmem_epoch{1};

struct DeviceRanges {
  std::vector<Range> known;
  uint64_t epoch = 0;
  bool refresh() { // true: a new epoch began (first use on this thread since the last synchronisation point)
    const uint64_t e = g_devmem_epoch.load(std::memory_order_relaxed);
    if (e == epoch) return false;
    known.clear();
    epoch = e;
    return true;
  }
  // index of the last hit PER OPERAND POSITION (A, B, C, D of consecutive invokes each stay in their own allocation;
  // one shared index would miss on every operand and fall into the scan)
  mutable size_t mru[4] = {0, 0, 0, 0};
  bool contains(const void *p, int pos = 0) const {
    const uintptr_t a = (uintptr_t)p;
    size_t &m = mru[pos & 3];
    if (m < known.size() && a >= known[m].b && a < known[m].e) return true;
    for (size_t i = 0; i < known.size(); ++i)
      if (a >= known[i].b && a < known[i].e) {
        m = i;
        return true;
      }
    return false;
  }
  Range range_of(const void *p) const { // the allocation that holds p, {0, 0} if unknown
    const uintptr_t a = (uintptr_t)p;
    for (const Range &r : known)
      if (a >= r.b && a < r.e) return r;
    return Range{0, 0};
  }
  uintptr_t base_of(const void *p) const { // allocation base, 0 if unknown
    const uintptr_t a = (uintptr_t)p;
    for (const Range &r : known)
      if (a >= r.b && a < r.e) return r.b;
    return 0;
  }
  bool is_device(const void *p, int pos = 0) {
    if (!p || contains(p, pos)) return true;
    if (!is_device_ptr(p)) return false;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) == hipSuccess && size) {
      if (known.size() >= 64) known.erase(known.begin());
      known.push_back(Range{(uintptr_t)base, (uintptr_t)base + size});
    } else {
      (void)hipGetLastError();
    }
    return true;
  }
};

// One operand of an invoke: [ptr, ptr + bytes), read and/or written by the kernel.
struct Operand {
  void *ptr;
  size_t bytes;
  bool written;
  void *dev; // resolved device pointer
  // optional 2-D shape of the footprint (rows of row_bytes every pitch bytes); 0 = one flat range.
  // Neighbouring tiles of one row-major buffer have interleaved rows, so their bounding ranges overlap
  // although the tiles do not: the tile queue's dependence tracking and the host mirror (which must
  // copy back ONLY the bytes the kernel writes) both work on this shape.
  size_t rows = 0, row_bytes = 0, pitch = 0;
  bool read = true;   // the kernel reads it (false: pure outputs, e.g. C under BETA_0)
  bool host = false;  // set by stage_in: the operand is host memory and `dev` points into a mirror
  void shape(int64_t r, size_t rb, size_t p) {
    if (r > 1 && p > rb) rows = (size_t)r, row_bytes = rb, pitch = p;
  }
};

// ---- host residents (extension): host buffers the harness declares stable -----------------------------
// The reference's callers pass host pointers and the ABI has no allocation / free hook, so a mirror can
// never be cached behind the caller's back (a freed and re-allocated range would alias a stale copy).
// A harness that knows a host buffer is long-lived (weights, inputs of a timing loop) can say so:
// xsmm_hip_host_resident(ptr, bytes) uploads it once and keeps a device copy; invokes whose operands lie
// inside a resident range use that copy without any upload (written operands are still copied back, so
// the host view stays current); xsmm_hip_host_update(ptr) re-uploads after the host changed the buffer;
// xsmm_hip_host_release(ptr) drops it.
struct Resident {
  char *host;
  size_t bytes;
  char *dev;
};
std::mutex g_res_mu;
std::vector<Resident> g_residents;
std::atomic<int> g_n_residents{0};

char *resident_dev(const void *p, size_t bytes) {
  if (!g_n_residents.load(std::memory_order_acquire)) return nullptr;
  std::lock_guard<std::mutex> lk(g_res_mu);
  for (const Resident &r : g_residents)
    if ((const char *)p >= r.host && (const char *)p + bytes <= r.host + r.bytes) return r.dev + ((const char *)p - r.host);
  return nullptr;
}

// pinned staging for the copy-back of small strided tiles (per thread, grow only)
struct Staging {
  char *base = nullptr;
  size_t cap = 0;
  char *get(size_t bytes) {
    if (bytes > cap) {
      if (base) HIP_OK(hipHostFree(base));
      cap = std::max(bytes, cap * 2);
      HIP_OK(hipHostMalloc((void **)&base, cap, hipHostMallocDefault));
    }
    return base;
  }
};
thread_local Staging t_staging;

// Resolve every operand to a device pointer. Device memory is used in place. Host operands are mirrored:
// overlapping host ranges (in-place relu, binary with out == lhs) share one mirror allocation, every
// operand the kernel READS is uploaded with its own shape (rows x row_bytes at the host pitch - the mirror
// keeps the host layout, gaps are never touched), pure outputs are not uploaded at all (C under BETA_0).
void stage_in(std::vector<Operand *> &ops, hipStream_t s) {
  std::vector<Operand *> host_ops;
  // async mode: device allocations seen in this synchronisation epoch cost one driver query each. In the
  // (default) synchronous mode every invoke is a point after which the caller may free buffers: query each time.
  thread_local DeviceRanges devmem;
  if (cfg().async.load(std::memory_order_relaxed)) devmem.refresh();
  else devmem.known.clear();
  for (Operand *o : ops) {
    o->host = false;
    if (!o->ptr || o->bytes == 0 || devmem.is_device(o->ptr)) o->dev = o->ptr;
    else host_ops.push_back(o);
  }
  if (host_ops.empty()) return;
  struct Span {
    char *host;
    size_t bytes;
    char *dev;
  };
  std::vector<Span> spans;
  std::vector<Operand *> mirrored;
  for (Operand *o : host_ops) {
    o->host = true;
    if (char *d = resident_dev(o->ptr, o->bytes)) o->dev = d, o->read = false; // device copy is current: nothing to upload
    else mirrored.push_back(o);
  }
  std::sort(mirrored.begin(), mirrored.end(), [](Operand *a, Operand *b) { return a->ptr < b->ptr; });
  for (Operand *o : mirrored) {
    char *b = (char *)o->ptr;
    if (!spans.empty() && b < spans.back().host + spans.back().bytes)
      spans.back().bytes = std::max(spans.back().bytes, (size_t)(b + o->bytes - spans.back().host));
    else spans.push_back({b, o->bytes, nullptr});
  }
  size_t total = 0;
  for (Span &m : spans) total += m.bytes + 512;
  t_arena.reserve(total, s);
  for (Span &m : spans) // keep the host address's offset within 256 B so alignment-dependent kernel choices see the caller's real alignment
    m.dev = t_arena.alloc(m.bytes + 256) + (((uintptr_t)m.host) & 255);
  for (Operand *o : mirrored)
    for (Span &m : spans)
      if ((char *)o->ptr >= m.host && (char *)o->ptr < m.host + m.bytes) {
        o->dev = m.dev + ((char *)o->ptr - m.host);
        break;
      }
  for (Operand *o : mirrored) {
    // an operand that is written AND overlaps a read operand (in-place ops) is covered by that operand's upload
    if (!o->read) continue;
    if (o->rows && o->rows * o->row_bytes * 2 < o->bytes)
      HIP_OK(hipMemcpy2DAsync(o->dev, o->pitch, o->ptr, o->pitch, o->row_bytes, o->rows, hipMemcpyHostToDevice, s));
    else
      HIP_OK(hipMemcpyAsync(o->dev, o->ptr, o->bytes, hipMemcpyHostToDevice, s));
  }
}

// Copy back what the kernel wrote - and only that: rows x row_bytes of a strided tile, never the gap bytes
// between its rows (they belong to neighbouring tiles other threads may be writing right now). Small tiles go
// through a pinned staging buffer + row-wise memcpy on this thread; large ones through hipMemcpy2DAsync.
void finish(std::vector<Operand *> &ops, hipStream_t s) {
  bool any_host = false;
  struct Late {
    Operand *o;
    char *stage;
  };
  Late late[4];
  int n_late = 0;
  size_t stage_bytes = 0;
  for (Operand *o : ops)
    if (o->host && o->written && o->rows && o->bytes <= (1u << 20)) stage_bytes += o->bytes;
  char *stage = stage_bytes ? t_staging.get(stage_bytes) : nullptr;
  for (Operand *o : ops) {
    if (!o->host) continue;
    any_host = true;
    if (!o->written) continue;
    if (!o->rows) {
      HIP_OK(hipMemcpyAsync(o->ptr, o->dev, o->bytes, hipMemcpyDeviceToHost, s));
    } else if (o->bytes <= (1u << 20) && n_late < 4) {
      HIP_OK(hipMemcpyAsync(stage, o->dev, o->bytes, hipMemcpyDeviceToHost, s));
      late[n_late++] = Late{o, stage};
      stage += o->bytes;
    } else {
      HIP_OK(hipMemcpy2DAsync(o->ptr, o->pitch, o->dev, o->pitch, o->row_bytes, o->rows, hipMemcpyDeviceToHost, s));
    }
  }
  if (any_host || !cfg().async.load(std::memory_order_relaxed)) HIP_OK(hipStreamSynchronize(s));
  for (int i = 0; i < n_late; ++i)
    for (size_t r = 0; r < late[i].o->rows; ++r)
      memcpy((char *)late[i].o->ptr + r * late[i].o->pitch, late[i].stage + r * late[i].o->pitch, late[i].o->row_bytes);
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

// ---- tile queue -------------------------------------------------------------------
// The compiler's native granularity is hundreds of invokes per layer on 32x32 tiles from
// OpenMP workers; one launch per invoke would be pure launch latency on a GPU. In async
// mode with the tile queue on, invokes of ONE small-tile GEMM handle on device pointers are
// appended to a work list and run as ONE grouped launch (brgemm_grouped) when something
// forces a flush: another handle or op, a data dependence on a queued output, capacity, a
// synchronize / perf_stop_timer, or leaving async mode. Program order is preserved: a new
// invoke that reads or overwrites anything a queued invoke writes (or overwrites anything a
// queued invoke reads) flushes first, so queued invokes are always mutually independent.
// union of half-open intervals: a sorted vector of disjoint ranges (a handful in practice - queued
// operands of one layer merge into a few runs - so a contiguous array beats a node-based map; the
// enqueue path runs 9 of these operations per invoke and is the throughput limit of the tile queue)
struct IntervalSet {
  std::vector<Range> iv; // sorted by begin, disjoint and non-touching
  void clear() { iv.clear(); }
  // index of the first interval whose begin is > x
  size_t upper(uintptr_t x) const {
    size_t lo = 0, hi = iv.size();
    if (hi <= 8) { // linear scan from the back: new operands are usually at or near the last run
      while (hi > 0 && iv[hi - 1].b > x) --hi;
      return hi;
    }
    while (lo < hi) {
      const size_t mid = (lo + hi) / 2;
      if (iv[mid].b > x) hi = mid;
      else lo = mid + 1;
    }
    return lo;
  }
  bool overlaps(const Range &r) const {
    if (r.b >= r.e || iv.empty()) return false;
    const size_t i = upper(r.e - 1); // intervals [0, i) begin before r.e
    return i > 0 && iv[i - 1].e > r.b;
  }
  void insert(Range r) {
    if (r.b >= r.e) return;
    size_t i = upper(r.b); // iv[i-1].b <= r.b < iv[i].b
    if (i > 0 && iv[i - 1].e >= r.b) { // r starts inside (or right at the end of) its predecessor
      if (iv[i - 1].e >= r.e) return;  // already covered: the common case for re-read operands
      --i;
      r.b = iv[i].b;
    }
    size_t j = i; // [i, j) are swallowed by r
    while (j < iv.size() && iv[j].b <= r.e) {
      r.e = std::max(r.e, iv[j].e);
      ++j;
    }
    if (j == i) iv.insert(iv.begin() + i, r);
    else {
      iv[i] = r;
      if (j > i + 1) iv.erase(iv.begin() + i + 1, iv.begin() + j);
    }
  }
};

// Footprint of the queued invokes' reads (or writes). Flat ranges are kept exactly in an interval
// set. A 2-D tile (rows x row_bytes, pitch) is kept as a rectangle in the "plane" (allocation base,
// pitch) it lives in - exact overlap tests between tiles of one row-major buffer, which is what pack /
// unpack tiles and the C tiles of a flat layer are - with a 64-row x 256-byte cell hash so a test
// touches a handful of rectangles. Anything that does not fit a plane falls back to its bounding
// range, and tests across different planes / against flat ranges use bounding ranges: conservative
// (may flush early), never unsafe.
struct Footprint {
  struct Rect { uint32_t r0, r1, c0, c1; };
  struct Plane {
    uintptr_t anchor;
    size_t pitch;
    IntervalSet bound;
    std::unordered_multimap<uint64_t, Rect> cells;
  };
  IntervalSet flat;
  std::vector<Plane> planes;
  void clear() { flat.clear(); planes.clear(); }
  static Range bounding(const Operand &o) { return Range{(uintptr_t)o.ptr, (uintptr_t)o.ptr + o.bytes}; }
  // rectangle of o in the plane (anchor, o.pitch); false if o is flat / wraps / is too wide for the hash
  static bool to_rect(const Operand &o, uintptr_t anchor, Rect &r) {
    if (!o.rows || !anchor) return false;
    const uintptr_t off = (uintptr_t)o.ptr - anchor;
    const uintptr_t r0 = off / o.pitch, c0 = off % o.pitch;
    if (c0 + o.row_bytes > o.pitch || o.row_bytes > 2048 || o.rows > 512 || r0 + o.rows > 0xffffffffu) return false;
    r = Rect{(uint32_t)r0, (uint32_t)(r0 + o.rows), (uint32_t)c0, (uint32_t)(c0 + o.row_bytes)};
    return true;
  }
  template <typename F> static void for_cells(const Rect &r, F f) {
    for (uint32_t cr = r.r0 / 64; cr <= (r.r1 - 1) / 64; ++cr)
      for (uint32_t cc = r.c0 / 256; cc <= (r.c1 - 1) / 256; ++cc) f(((uint64_t)cr << 32) | cc);
  }
  bool overlaps(const Operand &o, uintptr_t anchor) const {
    if (!o.ptr || !o.bytes) return false;
    const Range b = bounding(o);
    if (flat.overlaps(b)) return true;
    Rect r{0, 0, 0, 0};
    const bool is_rect = to_rect(o, anchor, r);
    for (const Plane &p : planes) {
      if (!p.bound.overlaps(b)) continue;
      if (!is_rect || p.anchor != anchor || p.pitch != o.pitch) return true;
      bool hit = false;
      for_cells(r, [&](uint64_t key) {
        auto range = p.cells.equal_range(key);
        for (auto it = range.first; it != range.second && !hit; ++it) {
          const Rect &q = it->second;
          hit = q.r0 < r.r1 && r.r0 < q.r1 && q.c0 < r.c1 && r.c0 < q.c1;
        }
      });
      if (hit) return true;
    }
    return false;
  }
  void insert(const Operand &o, uintptr_t anchor) {
    if (!o.ptr || !o.bytes) return;
    Rect r;
    if (!to_rect(o, anchor, r)) {
      flat.insert(bounding(o));
      return;
    }
    Plane *pl = nullptr;
    for (Plane &p : planes)
      if (p.anchor == anchor && p.pitch == o.pitch) pl = &p;
    if (!pl) {
      planes.push_back(Plane{anchor, o.pitch, {}, {}});
      pl = &planes.back();
    }
    pl->bound.insert(bounding(o));
    for_cells(r, [&](uint64_t key) { pl->cells.emplace(key, r); });
  }
};

// tile-queue counters (xsmm_hip_tile_queue_stats): launches, invokes queued with full bookkeeping / by replay, abandoned replays
std::atomic<int64_t> g_q_launches{0}, g_q_checked{0}, g_q_replayed{0}, g_q_abandoned{0}, g_q_terminated{0};
// (bumped only by whoever owns the queue state at that moment - the inline queue's lock holder or the scheduler thread: a
// plain load + store, not a locked read-modify-write on the enqueue path)
inline void bump(std::atomic<int64_t> &c) { c.store(c.load(std::memory_order_relaxed) + 1, std::memory_order_relaxed); }

// One queued invoke as the trace cache remembers it.
struct TraceItem {
  const void *desc = nullptr;
  WorkItem w{};
  hipStream_t stream = nullptr;
  bool same(const void *d, const WorkItem &x, hipStream_t s) const {
    return desc == d && w.A == x.A && w.B == x.B && w.C == x.C && w.D == x.D && w.br == x.br && stream == s;
  }
};
// A group as it was once collected: its invokes (in the order of that collection, and as a hash set) and the invokes that have
// been seen to end it by conflicting with it.
struct Segment {
  std::vector<TraceItem> items;
  std::vector<TraceItem> terminators;
  std::vector<uint32_t> seen; // round in which items[i] was last replayed (an invoke may join a group once); marked with atomic
                              // exchanges: callers mark their own arrivals while a direct window is open (DirectWindow)
  std::vector<int32_t> table; // open addressing over items, -1 = empty
  uint32_t round = 0;
  bool vec_ok = true, out_ok = true, pair_ok = true;
  uint64_t last_use = 0;
  // The group's work list as the grouped kernels read it: items[i].w in recorded order, in pinned host memory, written once when
  // the group is first replayed. A replay in which EVERY member arrives launches straight from it - nobody copies a work item.
  WorkItem *list = nullptr, *list_dev = nullptr; // ... and its copy in device memory (what the launches read: no PCIe round trip at the head of every workgroup)
  size_t list_cap = 0;
  bool list_valid = false, list_used = false; // holds items[] of THIS recording / a launch may still be reading it
  hipStream_t list_stream = nullptr;           // ... on this stream
  // GRID (round 5, detect_grid below): the group's gemm invokes tile ONE flat problem - a complete replay is then ONE launch of the
  // merged problem's own kernel. 0: not looked at yet, 1: grid_desc / grid_w hold the merged problem, -1: not a grid
  int grid_state = 0;
  const GemmDesc *grid_desc = nullptr;
  WorkItem grid_w{};
  // Called with the inline queue's lock held, once per RECORDING (a steady-state replay never comes here). The buffers are sized
  // for the largest group (TileQueue::CAP) the first time a segment needs them and then travel with it (store_recording swaps
  // segments, so at most NSEG + 1 sets exist per queue: allocation is a start-up cost, not a per-recording one); they live as long
  // as the process (like the pinned work-list slots: no HIP call at exit). While the stream is being CAPTURED into a graph no list
  // is built (allocation / synchronisation are not legal there): the replay then gathers its members into a pinned slot at the
  // flush like an incomplete group (TileQueue::flush) - returns false.
  bool ensure_list(hipStream_t stream, size_t cap) {
    if (list_valid) return true;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess) (void)hipGetLastError();
    else if (cs != hipStreamCaptureStatusNone) return false;
    if (list_used) HIP_OK(hipStreamSynchronize(list_stream)); // the buffers carried another recording's items: its last launch must be done
    list_used = false;
    if (list_cap < items.size()) {
      if (list) HIP_OK(hipHostFree(list));
      if (list_dev) HIP_OK(hipFree(list_dev));
      list_cap = items.size() < cap ? cap : items.size();
      HIP_OK(hipHostMalloc((void **)&list, sizeof(WorkItem) * list_cap, hipHostMallocDefault));
      HIP_OK(hipMalloc((void **)&list_dev, sizeof(WorkItem) * list_cap));
    }
    for (size_t i = 0; i < items.size(); ++i) list[i] = items[i].w;
    HIP_OK(hipMemcpyAsync(list_dev, list, sizeof(WorkItem) * items.size(), hipMemcpyHostToDevice, stream));
    list_used = true; // (the copy reads `list`)
    list_stream = stream;
    list_valid = true;
    return true;
  }
  // Proof that every recorded pointer is device memory, per synchronisation epoch (DeviceRanges): the (few) allocations that hold
  // them, collected the first time the group is replayed and re-verified - same base, same extent - once per epoch by whoever
  // opens the group's window, under the queue's lock. A caller whose invoke matches a recorded member while dev_epoch is the
  // current epoch skips its own four range checks (a quarter of the lock-free path); without a proof it checks as before.
  static constexpr int MAX_ALLOC = 12;
  Range alloc[MAX_ALLOC];
  int n_alloc = -1;       // -1: not collected
  uint64_t dev_epoch = 0; // written under the lock before the window opens, read inside the window
  bool prove(DeviceRanges &dm, uint64_t epoch) {
    if (dev_epoch == epoch) return true;
    dev_epoch = 0;
    if (n_alloc >= 0) {
      bool same = true;
      for (int i = 0; i < n_alloc && same; ++i) {
        const Range r = dm.is_device((const void *)alloc[i].b) ? dm.range_of((const void *)alloc[i].b) : Range{0, 0};
        same = r.b == alloc[i].b && r.e == alloc[i].e;
      }
      if (same) {
        dev_epoch = epoch;
        return true;
      }
      n_alloc = -1; // an allocation went away or changed: collect again
    }
    int n = 0, last = 0;
    for (const TraceItem &t : items) {
      const void *ptrs[4] = {t.w.A, t.w.B, t.w.C, t.w.D};
      for (const void *q : ptrs) {
        if (!q) continue;
        const uintptr_t a = (uintptr_t)q;
        if (n && a >= alloc[last].b && a < alloc[last].e) continue;
        int j = 0;
        while (j < n && !(a >= alloc[j].b && a < alloc[j].e)) ++j;
        if (j == n) {
          if (n == MAX_ALLOC || !dm.is_device(q)) return false;
          const Range r = dm.range_of(q);
          if (!r.e) return false; // (device memory without an address range: not provable, the callers keep checking)
          alloc[n++] = r;
        }
        last = j;
      }
    }
    n_alloc = n;
    dev_epoch = epoch;
    return true;
  }
  bool mark(int idx) { return __atomic_exchange_n(&seen[idx], round, __ATOMIC_RELAXED) != round; } // false: joined this round already
  bool mark_solo(int idx) { // one caller in the whole process (DirectWindow, SOLO): nobody else marks
    if (__atomic_load_n(&seen[idx], __ATOMIC_RELAXED) == round) return false;
    __atomic_store_n(&seen[idx], round, __ATOMIC_RELAXED);
    return true;
  }
  static size_t hash(const WorkItem &w) {
    uint64_t h = (uint64_t)(uintptr_t)w.C * 0x9E3779B97F4A7C15ull;
    h ^= ((uint64_t)(uintptr_t)w.A >> 4) * 0xC2B2AE3D27D4EB4Full;
    h ^= ((uint64_t)(uintptr_t)w.B >> 4) * 0x165667B19E3779F9ull;
    return (size_t)(h ^ (h >> 29));
  }
  void build() {
    // by output address: membership is all a replay needs, and with the reference's static schedules a caller's invokes then sit
    // next to each other - its arrival marks in `seen` share cache lines with its own marks only, and "the one after my last" is
    // usually the next invoke (DirectWindow::Caller::hint) without a hash lookup
    std::stable_sort(items.begin(), items.end(), [](const TraceItem &a, const TraceItem &b) { return (uintptr_t)a.w.C < (uintptr_t)b.w.C; });
    size_t cap = 16;
    while (cap < 2 * items.size()) cap *= 2;
    table.assign(cap, -1);
    for (size_t i = 0; i < items.size(); ++i) {
      size_t at = hash(items[i].w) & (cap - 1);
      while (table[at] >= 0) at = (at + 1) & (cap - 1);
      table[at] = (int32_t)i;
    }
    seen.assign(items.size(), 0);
    round = 0;
    list_valid = false;
    n_alloc = -1;
    dev_epoch = 0;
    grid_state = 0;
  }
  int index_of(const void *d, const WorkItem &w, hipStream_t st) const {
    if (table.empty()) return -1;
    const size_t mask = table.size() - 1;
    for (size_t at = hash(w) & mask; table[at] >= 0; at = (at + 1) & mask)
      if (items[table[at]].same(d, w, st)) return table[at];
    return -1;
  }
  bool is_terminator(const void *d, const WorkItem &w, hipStream_t st) const {
    for (const TraceItem &t : terminators)
      if (t.same(d, w, st)) return true;
    return false;
  }
};

// GRID MERGE (round 5). The compiler tiles a contraction over FLAT operands into one gemm invoke per output tile (the mha projection,
// benchmarks/mlir/fp32-projection.mlir: 64 x 8 invokes of [32,64,512,512,512,512] = one 2048 x 512 x 512 problem with lda = ldb = ldc =
// 512). When a recorded group is exactly such a grid - every item the same f32 descriptor and batch count, A a function of the tile row
// only (A0 + r m lda), B of the tile column only (B0 + c n, all columns inside one row of B: cols n <= ldb), C = C0 + r m ldc + c n, the
// bias D0 + c n, every (r, c) once - a complete replay of it is launched as ONE invoke of the merged problem (rows m, cols n) on the
// kernel plan_gemm picks for THAT shape (64x64 tiles instead of 1024 workgroups of 32x32 with 8 chunks each: 13.0 -> ~10.5 us). Same
// reads, same writes, the same sums per element in a different (fixed) order: like every kernel choice that depends on the group.
// Packed block layouts (mlir-gen's tiles) are never grids: their B tiles are not columns of one row. TPP_HIP_GRID_MERGE=0: off.
static bool grid_merge_on() {
  static const bool on = [] {
    const char *e = getenv("TPP_HIP_GRID_MERGE");
    return !e || atoi(e) != 0;
  }();
  return on && !cfg().strict.load(std::memory_order_relaxed); // (a merged grid sums in the merged problem's order: not in strict mode)
}
std::atomic<const char *> g_last_merged{nullptr}; // trace text of the merged descriptor if the most recent group launch was a merged one
inline void detect_grid(Segment &S) {
  S.grid_state = -1;
  const size_t n = S.items.size();
  if (!grid_merge_on() || n < 4) return;
  const void *desc = S.items[0].desc;
  if (*(const int *)desc != KIND_GEMM) return;
  const GemmDesc *d = (const GemmDesc *)desc;
  if (d->dtype != DT_F32 || d->vnni_b || d->vnni_c || d->b_trans || d->generic_forced || d->variant_forced || d->m <= 0 || d->n <= 0 || d->k <= 0) return;
  const int64_t br = S.items[0].w.br;
  std::vector<uintptr_t> ua, ub;
  ua.reserve(n);
  ub.reserve(n);
  for (const TraceItem &t : S.items) {
    if (t.desc != desc || t.w.br != br || t.stream != S.items[0].stream) return;
    ua.push_back((uintptr_t)t.w.A);
    ub.push_back((uintptr_t)t.w.B);
  }
  std::sort(ua.begin(), ua.end());
  ua.erase(std::unique(ua.begin(), ua.end()), ua.end());
  std::sort(ub.begin(), ub.end());
  ub.erase(std::unique(ub.begin(), ub.end()), ub.end());
  const size_t R = ua.size(), Cn = ub.size();
  if (R * Cn != n || br < 1) return;
  const uintptr_t sa = (uintptr_t)d->m * (uintptr_t)d->lda * 4, sb = (uintptr_t)d->n * 4;
  for (size_t r = 0; r < R; ++r)
    if (ua[r] != ua[0] + r * sa) return;
  for (size_t c = 0; c < Cn; ++c)
    if (ub[c] != ub[0] + c * sb) return;
  if ((int64_t)Cn * d->n > d->ldb || (int64_t)Cn * d->n > d->ldc) return;
  uintptr_t c0 = 0, d0 = 0;
  for (const TraceItem &t : S.items)
    if ((uintptr_t)t.w.A == ua[0] && (uintptr_t)t.w.B == ub[0]) c0 = (uintptr_t)t.w.C, d0 = (uintptr_t)t.w.D;
  if (!c0) return;
  std::vector<char> seen(n, 0);
  for (const TraceItem &t : S.items) {
    const size_t r = ((uintptr_t)t.w.A - ua[0]) / sa, c = ((uintptr_t)t.w.B - ub[0]) / sb;
    if ((uintptr_t)t.w.C != c0 + ((uintptr_t)r * d->m * d->ldc + (uintptr_t)c * d->n) * 4) return;
    if (d->bias && (uintptr_t)t.w.D != d0 + (uintptr_t)c * d->n * 4) return;
    if (seen[r * Cn + c]++) return;
  }
  const int64_t M = (int64_t)R * d->m, N = (int64_t)Cn * d->n;
  std::vector<int64_t> key = {KIND_GEMM, -31, (int64_t)(uintptr_t)d, M, N};
  bool ok = true;
  const GemmDesc *e = (const GemmDesc *)intern(key, [&]() {
    GemmDesc *g = new GemmDesc(*d);
    g->m = M;
    g->n = N;
    ok = plan_gemm(*g, -1);
    snprintf(g->trace, sizeof(g->trace), "tile grid %zu x %zu of gemm[%ld,%ld,%ld] merged -> [%ld,%ld,%ld,%ld,%ld,%ld] %s", R, Cn, (long)d->m, (long)d->n,
             (long)d->k, (long)M, (long)N, (long)d->k, (long)d->lda, (long)d->ldb, (long)d->ldc, g->name);
    return (void *)g;
  });
  if (!ok || e->variant == GEMM_VARIANT_GENERIC) return; // (no fast tile for the merged shape: the grouped launch stays)
  S.grid_desc = e;
  S.grid_w = WorkItem{(const void *)ua[0], (const void *)ub[0], (void *)c0, d->bias ? (const void *)d0 : nullptr, br};
  S.grid_state = 1;
}

// DIRECT WINDOW: replayed members arrive without a lock. While the inline queue replays a recorded group, `cur` names it
// (generation << 7 | segment index + 1) and a caller whose invoke is a member marks it in the segment (Segment::mark) and counts it
// in its OWN cache line - no work item is written (the segment's pinned list already holds it) and no line is shared between
// callers except `cur`, which changes once per group. Whoever has to change the queue state - a terminator, an invoke the cache
// does not know, a flush point - holds the queue's lock, CLOSES the window (cur = 0) and waits until no caller is inside it:
//   caller: busy = cur (seq_cst); re-read cur (seq_cst); ... mark, count ...; busy = 0 (release)
//   closer: cur = 0 (seq_cst); for every caller: wait until busy == 0 (seq_cst / acquire), then read its count
// a Dekker pair per caller: either the caller sees the closed window and takes the locked path, or the closer sees it busy and waits
// for its arrival to be complete. Invokes are processed synchronously on this path (when xsmm_*_invoke returns, the invoke is in the
// group or launched), so everything that happened before an invoke is in the queue state when it arrives: program order and every
// happens-before between callers hold without time stamps. Membership was proven conflict-free when the group was recorded.
//
// SOLO: as long as ONE thread is all the queue has ever seen (tpp-run without OpenMP, the reference's default), the caller's half of
// the Dekker pair is plain stores and loads and the arrival mark a load + store: the two locked instructions (xchg for the seq_cst
// store of `busy`, xchg for the mark) are 35-40 cycles of an invoke that costs ~100. The fence moves to the side that runs ONCE: the
// first time a second thread touches the queue state (claims a caller slot, or takes the queue's lock) it sets `multi`, issues
// membarrier(PRIVATE_EXPEDITED) - a full barrier on every CPU running a thread of this process - and waits until no solo section
// is in flight (`seq` of every caller even; a section brackets itself with seq++ ... seq++, and a section that read multi = false
// before the barrier had made its seq store by then: stores are not reordered with OLDER loads' retirement, an interrupt discards
// a load that ran ahead of an unretired store). From then on, for good, the protocol above. No membarrier (seccomp): never solo.
static uintptr_t thread_token() {
  static thread_local char t;
  return (uintptr_t)&t;
}
struct DirectWindow {
  static constexpr int MAXC = 256;
  struct alignas(64) Caller {
    std::atomic<uint64_t> busy{0};
    uint64_t tag = 0;   // window the count belongs to   (written inside the busy section, read by the closer after it)
    uint32_t count = 0; // arrivals in that window
    uint32_t hint = 0;  // index after this caller's last arrival
    std::atomic<int> owned{0};
    std::atomic<uint64_t> seq{0}; // odd while the owner is inside a SOLO section (written by the owner only, relaxed)
  };
  alignas(64) std::atomic<uint64_t> cur{0};
  alignas(64) std::atomic<int> ncallers{0}; // high-water mark of claimed caller slots
  std::atomic<bool> multi{true};             // false: SOLO
  std::atomic<bool> multi_ready{true};       // the switch to multi has completed (nobody is inside a solo section any more)
  std::atomic<uintptr_t> solo_owner{0};      // thread_token() of the one thread
  Caller callers[MAXC];
  DirectWindow() {
    const char *e = getenv("TPP_HIP_QUEUE_SOLO"); // 0: the two-sided protocol from the start (A/B runs)
    if ((!e || atoi(e) != 0) && !getenv("TPP_HIP_NO_MEMBARRIER") && syscall(__NR_membarrier, MEMBARRIER_CMD_REGISTER_PRIVATE_EXPEDITED, 0) == 0) {
      multi.store(false, std::memory_order_relaxed);
      multi_ready.store(false, std::memory_order_relaxed);
    }
  }
  // every entry to the queue state that is not a solo section (claiming a slot, taking the queue's lock) says who it is
  void touch(uintptr_t me) {
    if (multi.load(std::memory_order_acquire)) {
      while (!multi_ready.load(std::memory_order_acquire)) cpu_relax(); // (another thread is switching right now)
      return;
    }
    uintptr_t o = solo_owner.load(std::memory_order_acquire);
    if (o == me) return;
    if (o == 0 && solo_owner.compare_exchange_strong(o, me, std::memory_order_seq_cst)) return;
    bool expect = false;
    if (multi.compare_exchange_strong(expect, true, std::memory_order_seq_cst)) {
      if (syscall(__NR_membarrier, MEMBARRIER_CMD_PRIVATE_EXPEDITED, 0) != 0) die("tpp-xsmm-hip: membarrier failed");
      for (int i = 0; i < MAXC; ++i)
        while (callers[i].seq.load(std::memory_order_acquire) & 1) cpu_relax();
      multi_ready.store(true, std::memory_order_release);
    } else {
      while (!multi_ready.load(std::memory_order_acquire)) cpu_relax();
    }
  }
  Caller *claim() {
    touch(thread_token());
    const int n = ncallers.load(std::memory_order_acquire);
    for (int i = 0; i < MAXC; ++i) {
      int expect = 0;
      if (callers[i].owned.load(std::memory_order_relaxed) == 0 && callers[i].owned.compare_exchange_strong(expect, 1, std::memory_order_seq_cst)) {
        int hw = n;
        while (hw < i + 1 && !ncallers.compare_exchange_weak(hw, i + 1, std::memory_order_seq_cst)) {
        }
        return &callers[i];
      }
    }
    return nullptr; // more caller threads than slots: this one always takes the locked path
  }
};

struct TileQueue {
  static constexpr int CAP = 4096, SLOTS = 32, GROUP = 8; // work-list slots; one completion event per GROUP slots
  // TRACE CACHE. Compiled code repeats itself: the same handles on the same pointers in the same order, iteration after
  // iteration (the timing loop of tpp-run, every layer of a model). Whether a group of queued invokes is conflict-free,
  // and whether the next invoke conflicts with it, is a pure function of that sequence of (descriptor, pointers, batch)
  // - the footprints follow from them - so a group that was collected once with full bookkeeping is REPLAYED the next
  // time one of its invokes shows up on an empty queue: each following invoke that is a MEMBER of the recorded group
  // (compared with the next recorded one first, else looked up in the group's hash set) and has not joined in this
  // round is appended to the work list, nothing else; an invoke that has been seen to end the group launches it.
  // Membership, not order: a group is conflict-free iff its invokes are pairwise so, in any order and for any subset -
  // which is what several OpenMP callers produce, whose interleaving changes from iteration to iteration. Any other
  // invoke rebuilds the footprints of what has been queued and drops back to the full bookkeeping: if it conflicts, it
  // is remembered as one more terminator of the group; if it joins, the new group is recorded, and the cache is left
  // alone for a growing number of groups. Replaying a flush is always safe, skipping the checks is safe because the
  // same set was proven conflict-free.
  static constexpr size_t NSEG = 64, MIN_SEG = 16; // recorded groups kept (a 20-layer model repeats ~20 groups per iteration)
  std::vector<Segment> segs;
  int replay = -1;        // index of the segment being replayed
  size_t rpos = 0;        // the item expected next (a hint: the one after the last match)
  int learn = -1;         // a replay of this segment was just abandoned: if the invoke that did it conflicts, it is a terminator
  size_t learn_n = 0;     // ... provided the group still has this many invokes
  Segment rec;            // the group being recorded (full bookkeeping path)
  bool rec_open = false;
  uint64_t use_clock = 0;
  unsigned backoff = 0, backoff_next = 2; // groups to collect without consulting the cache / after the next mismatch
  int kind = 0;               // KIND_GEMM / KIND_UNARY / KIND_BINARY of the queued invokes
  const void *desc = nullptr; // their (single) descriptor
  bool vec_ok = true, out_ok = true, pair_ok = true;
  int n = 0;
  Footprint reads, writes;
  // Work lists live in host-pinned (device-mapped) memory and every workgroup reads its 40-byte item over PCIe, once,
  // at its head. Moving the list to HBM with one hipMemcpyAsync in front of each grouped launch was built and
  // measured (profiles/r02_tile_queue_device_lists.txt): the copy costs 15-20 us of host time per flush on this
  // runtime - the reference's headline pattern (3 flushes per iteration) went from 47 to 103 us - while the PCIe
  // read is ~1 us of latency that all workgroups pay in parallel.
  // A slot is reused SLOTS flushes later, once the launch that read it has finished. One event per flush cost ~2 us of host
  // time each (two of the 25 us of the headline bf16 pattern): the slots are used in groups of GROUP, ONE event is recorded
  // behind the last launch of a group, and it is waited for when the group is entered again - 24 launches later.
  WorkItem *pinned[SLOTS] = {};
  hipEvent_t done[SLOTS / GROUP] = {};
  bool used[SLOTS / GROUP] = {};
  hipStream_t gstream[SLOTS / GROUP] = {}; // the stream the group's launches went to
  int slot = 0;
  hipStream_t stream = nullptr;
  DirectWindow *dw = nullptr; // the inline queue's window (the scheduler thread's queue has none: its callers hand over through rings)
  uint64_t dw_gen = 0;
  bool window_open = false;
  int64_t direct_groups = 0;  // groups closed with lock-free arrivals in them
  // a grouped launch that has been decided but not issued: the whole recorded group, from its segment's list. Issued after the
  // NEXT group's window has been opened, so the other callers enter that group while this thread is inside hipLaunchKernel.
  struct Pending {
    bool armed = false;
    int kind = 0;
    const void *desc = nullptr;
    int seg = -1, n = 0;
    bool vec_ok = true, out_ok = true, pair_ok = true;
    hipStream_t stream = nullptr;
  } pending;
  TileQueue() { segs.reserve(NSEG); } // callers inside a direct window hold pointers into segs: it never reallocates

  // no caller is inside the window any more on return; the lock-free arrivals are added to n
  void close_window() {
    if (!window_open) return;
    window_open = false;
    const uint64_t c = dw->cur.load(std::memory_order_relaxed);
    dw->cur.store(0, std::memory_order_seq_cst);
    const int nc = dw->ncallers.load(std::memory_order_seq_cst);
    int arrived = 0;
    for (int i = 0; i < nc; ++i) {
      DirectWindow::Caller &k = dw->callers[i];
      while (k.busy.load(std::memory_order_seq_cst) != 0) cpu_relax();
      if (k.tag == c) arrived += (int)k.count;
    }
    if (arrived) {
      n += arrived;
      ++direct_groups;
      g_q_replayed.store(g_q_replayed.load(std::memory_order_relaxed) + arrived, std::memory_order_relaxed);
    }
  }
  void open_window(int seg) {
    if (!dw) return;
    ++dw_gen;
    window_open = true;
    dw->cur.store((dw_gen << 7) | (uint64_t)(seg + 1), std::memory_order_seq_cst);
  }
  // the members of the replayed group that have arrived, as a dense work list in pinned[slot] (window closed): a replay that ends
  // before every member has joined, or is abandoned
  void materialize() {
    const Segment &S = segs[replay];
    int k = 0;
    for (size_t i = 0; i < S.items.size(); ++i)
      if (__atomic_load_n(&S.seen[i], __ATOMIC_RELAXED) == S.round) pinned[slot][k++] = S.items[i].w;
    if (k != n) die("tpp-xsmm-hip: internal error: %d members marked, %d counted in a replayed group", k, n);
  }
  void issue_pending() {
    if (!pending.armed) return;
    pending.armed = false;
    Segment &S = segs[pending.seg];
    if (pending.kind == KIND_GEMM && S.grid_state == 0) detect_grid(S);
    if (pending.kind == KIND_GEMM && S.grid_state == 1) {
      HIP_OK(launch_gemm(*S.grid_desc, S.grid_w.A, S.grid_w.B, S.grid_w.C, S.grid_w.D, S.grid_w.br, pending.stream));
      g_last_merged.store(S.grid_desc->trace, std::memory_order_relaxed);
      return;
    }
    g_last_merged.store(nullptr, std::memory_order_relaxed);
    if (pending.kind == KIND_GEMM) HIP_OK(launch_gemm_grouped(*(const GemmDesc *)pending.desc, S.list_dev, pending.n, pending.vec_ok, pending.out_ok, pending.pair_ok, S.items[0].w.br, pending.stream));
    else if (pending.kind == KIND_UNARY) HIP_OK(launch_unary_grouped(*(const UnaryDesc *)pending.desc, S.list_dev, pending.n, pending.stream));
    else HIP_OK(launch_binary_grouped(*(const BinaryDesc *)pending.desc, S.list_dev, pending.n, pending.stream));
    S.list_used = true;
    S.list_stream = pending.stream;
  }

  void ensure_slot() {
    if (n != 0) return;
    const int g = slot / GROUP;
    if (slot % GROUP == 0 && used[g]) {
      HIP_OK(hipEventSynchronize(done[g])); // every launch that read a slot of this group has finished
      used[g] = false;
    }
    if (!pinned[slot]) HIP_OK(hipHostMalloc((void **)&pinned[slot], sizeof(WorkItem) * CAP, hipHostMallocDefault));
  }
  void launched() { // a grouped launch on `stream` has been issued from pinned[slot]
    const int g = slot / GROUP;
    if (slot % GROUP != 0 && gstream[g] != stream) HIP_OK(hipStreamSynchronize(gstream[g])); // (the caller changed streams inside a group)
    gstream[g] = stream;
    if (slot % GROUP == GROUP - 1) {
      if (!done[g]) HIP_OK(hipEventCreateWithFlags(&done[g], hipEventDisableTiming));
      HIP_OK(hipEventRecord(done[g], stream));
      used[g] = true;
    }
    slot = (slot + 1) % SLOTS;
  }
  void store_recording(const TraceItem *next) {
    if (learn >= 0 && next && rec_open && rec.items.size() == learn_n) {
      // the group is exactly what was replayed from segs[learn] and `next` conflicts with it: one more way that group ends
      Segment &S = segs[learn];
      if (S.terminators.size() < 64 && !S.is_terminator(next->desc, next->w, next->stream)) S.terminators.push_back(*next);
    } else if (rec_open && rec.items.size() >= MIN_SEG) {
      rec.terminators.clear();
      if (next) rec.terminators.push_back(*next);
      rec.vec_ok = vec_ok;
      rec.out_ok = out_ok;
      rec.pair_ok = pair_ok;
      rec.last_use = ++use_clock;
      rec.build();
      size_t at = segs.size();
      for (size_t i = 0; i < segs.size(); ++i)
        if (segs[i].index_of(rec.items[0].desc, rec.items[0].w, rec.items[0].stream) >= 0) at = i; // overlapping group: the newer one wins
      if (at == segs.size() && segs.size() >= NSEG) {
        at = 0;
        for (size_t i = 1; i < segs.size(); ++i)
          if (segs[i].last_use < segs[at].last_use) at = i;
      }
      if (at == segs.size()) segs.emplace_back();
      std::swap(segs[at], rec);
    }
    learn = -1;
    rec.items.clear();
    rec_open = false;
  }
  // the most recently used recorded group that contains the invoke; its index in *item
  int find_segment(const void *d, const WorkItem &w, hipStream_t s, int *item) {
    int best = -1;
    for (size_t i = 0; i < segs.size(); ++i) {
      const int idx = segs[i].index_of(d, w, s);
      if (idx >= 0 && (best < 0 || segs[i].last_use > segs[best].last_use)) {
        best = (int)i;
        *item = idx;
      }
    }
    return best;
  }
  // next: the invoke whose conflict ends this group (nullptr: an external flush point). defer: the launch may be left pending
  // (the caller opens the next group first and then calls issue_pending()).
  void flush(const TraceItem *next = nullptr, bool defer = false) {
    issue_pending();
    close_window();
    const int rp = replay;
    // the recorded group, complete, and its device-resident list exists (not while capturing): launch from its own list
    const bool whole = rp >= 0 && n > 0 && (size_t)n == segs[rp].items.size() && segs[rp].list_valid;
    if (rp >= 0 && n > 0 && !whole) materialize();
    // (counts are exact: an arrival is counted by whoever's atomic exchange on the item's mark saw it unmarked - once per round)
    store_recording(next); // (never touches segs[rp] during a replay: nothing is being recorded)
    replay = -1;
    if (n == 0) return;
    bump(g_q_launches);
    if (whole) {
      pending = Pending{true, kind, desc, rp, n, vec_ok, out_ok, pair_ok, stream};
      if (!defer) issue_pending();
    } else {
      if (kind == KIND_GEMM) g_last_merged.store(nullptr, std::memory_order_relaxed);
      if (kind == KIND_GEMM) HIP_OK(launch_gemm_grouped(*(const GemmDesc *)desc, pinned[slot], n, vec_ok, out_ok, pair_ok, pinned[slot][0].br, stream));
      else if (kind == KIND_UNARY) HIP_OK(launch_unary_grouped(*(const UnaryDesc *)desc, pinned[slot], n, stream));
      else HIP_OK(launch_binary_grouped(*(const BinaryDesc *)desc, pinned[slot], n, stream));
      launched();
    }
    n = 0;
    desc = nullptr;
    vec_ok = out_ok = pair_ok = true;
    reads.clear();
    writes.clear();
  }
};

// What a caller hands over to the scheduler: descriptor + pointers of one invoke - or a fence (desc == nullptr,
// w.C = the flag to raise). 56 bytes: with the slot's sequence word ONE cache line crosses from the caller's core
// to the scheduler's per invoke (a 300-byte entry with the footprints resolved on the caller's side cost five, and
// made two callers 3x slower than one); the scheduler derives the footprints itself (queued_operands).
struct QEntry {
  const void *desc = nullptr;
  WorkItem w{};
  hipStream_t stream = nullptr;
};

// does the invoke conflict with the group being collected (another handle / stream, capacity, a data dependence)?
inline bool conflicts_with_group(const TileQueue &q, int kind, const void *desc, const Operand &out, uintptr_t anchor_out,
                                 const Operand *const *in, const uintptr_t *anchor_in, int n_in, hipStream_t stream) {
  if (q.n == 0) return false;
  if (q.kind != kind || q.desc != desc || q.stream != stream || q.n >= TileQueue::CAP) return true;
  if (q.writes.overlaps(out, anchor_out) || q.reads.overlaps(out, anchor_out)) return true;
  for (int i = 0; i < n_in; ++i)
    if (q.writes.overlaps(*in[i], anchor_in[i])) return true;
  return false;
}
// appends one invoke to the group being collected (full bookkeeping; the group is recorded for the trace cache)
inline void append_to_group(TileQueue &q, int kind, const void *desc, const WorkItem &w, const Operand &out, uintptr_t anchor_out,
                            const Operand *const *in, const uintptr_t *anchor_in, int n_in, bool vec_ok, bool out_ok,
                            bool pair_ok, hipStream_t stream) {
  q.ensure_slot();
  q.kind = kind;
  q.desc = desc;
  q.stream = stream;
  q.vec_ok = q.vec_ok && vec_ok;
  q.out_ok = q.out_ok && out_ok;
  q.pair_ok = q.pair_ok && pair_ok;
  if (q.n == 0) { // a new group: record it
    q.rec.items.clear();
    q.rec_open = true;
  }
  bump(g_q_checked);
  if (q.rec_open) q.rec.items.push_back(TraceItem{desc, w, stream});
  if (q.learn >= 0) { // the invoke that ended a replay joined the group: the caller has left the recorded pattern
    q.learn = -1;
    q.backoff = q.backoff_next;
    if (q.backoff_next < 64) q.backoff_next *= 2;
  }
  q.pinned[q.slot][q.n++] = w;
  for (int i = 0; i < n_in; ++i) q.reads.insert(*in[i], anchor_in[i]);
  q.writes.insert(out, anchor_out);
}
// the first invoke of a group on an empty queue: replay the recorded group it belongs to, if there is one
inline bool try_start_replay(TileQueue &q, DeviceRanges &devmem, const void *desc, const WorkItem &w, hipStream_t stream) {
  if (q.backoff > 0) {
    --q.backoff;
    return false;
  }
  int item = 0;
  const int idx = q.find_segment(desc, w, stream, &item);
  if (idx < 0) return false;
  Segment &S = q.segs[idx];
  if (++S.round == 0) { // (wrapped: forget the marks)
    std::fill(S.seen.begin(), S.seen.end(), 0u);
    S.round = 1;
  }
  S.seen[item] = S.round;
  q.ensure_slot();
  (void)S.ensure_list(stream, (size_t)TileQueue::CAP); // (false while the stream is being captured: the flush gathers the members instead)
  q.kind = *(const int *)desc;
  q.desc = desc;
  q.stream = stream;
  q.vec_ok = S.vec_ok;
  q.out_ok = S.out_ok;
  q.pair_ok = S.pair_ok;
  q.n = 1; // members are marked and counted, not copied: the work list is S.list (all of them) or is gathered at the flush
  q.replay = idx;
  q.rpos = (size_t)item + 1;
  S.last_use = ++q.use_clock;
  bump(g_q_replayed);
  if (q.dw) (void)S.prove(devmem, devmem.epoch); // (devmem belongs to the thread that runs this and is of the current epoch)
  q.open_window(idx); // from here on the other members may arrive without the lock
  return true;
}
// bookkeeping of one queued invoke: footprints from the descriptor, allocation bases ("anchors" of the 2-D planes) from
// `devmem` - the allocation cache of the thread that runs this (every operand was seen to be device memory by the caller)
__attribute__((always_inline)) inline void process_item(TileQueue &q, DeviceRanges &devmem, const void *desc, const WorkItem &w, hipStream_t stream) {
  QueuedOps o;
  queued_operands(desc, w, o);
  const Operand *in[3] = {&o.op[0], &o.op[1], &o.op[2]};
  uintptr_t anchor_in[3] = {0, 0, 0};
  auto anchor = [&](const Operand &x) -> uintptr_t {
    if (!x.rows || !x.ptr) return 0;
    if (uintptr_t b = devmem.base_of(x.ptr)) return b;
    (void)devmem.is_device(x.ptr); // first sight of this allocation on this thread in this epoch
    return devmem.base_of(x.ptr);
  };
  for (int i = 0; i < o.n_in; ++i) anchor_in[i] = anchor(o.op[i]);
  const Operand &out = o.op[o.out];
  const uintptr_t anchor_out = anchor(out);
  const int kind = *(const int *)desc;
  // strict mode: a group holds invokes of ONE alignment class and ONE batch count - the grouped launch takes its operand path from
  // the AND of the members' alignment flags and its chunk count from the first member, so a mixed group would make a member's kernel
  // depend on its neighbours
  const bool strict_break = q.n > 0 && kind == KIND_GEMM && q.kind == KIND_GEMM && cfg().strict.load(std::memory_order_relaxed) &&
                            (o.vec_ok != q.vec_ok || o.out_ok != q.out_ok || o.pair_ok != q.pair_ok || w.br != q.pinned[q.slot][0].br);
  if (strict_break || conflicts_with_group(q, kind, desc, out, anchor_out, in, anchor_in, o.n_in, stream)) {
    const TraceItem term{desc, w, stream};
    q.flush(&term);
    if (try_start_replay(q, devmem, desc, w, stream)) return; // the group this invoke starts has been collected before
  }
  append_to_group(q, kind, desc, w, out, anchor_out, in, anchor_in, o.n_in, o.vec_ok, o.out_ok, o.pair_ok, stream);
}

// footprints of the queued invokes into the (empty) read / write sets: a replay is being abandoned
inline void rebuild_footprints(TileQueue &q, DeviceRanges &devmem) {
  for (int i = 0; i < q.n; ++i) {
    QueuedOps o;
    queued_operands(q.desc, q.pinned[q.slot][i], o);
    for (int j = 0; j <= o.out; ++j) {
      const Operand &x = o.op[j];
      uintptr_t a = 0;
      if (x.rows && x.ptr && !(a = devmem.base_of(x.ptr))) {
        (void)devmem.is_device(x.ptr);
        a = devmem.base_of(x.ptr);
      }
      if (j == o.out) q.writes.insert(x, a);
      else q.reads.insert(x, a);
    }
  }
}
// One queued invoke: replayed from the trace cache if it belongs to the recorded group being replayed, else the full bookkeeping.
inline void submit_item(TileQueue &q, DeviceRanges &devmem, const void *desc, const WorkItem &w, hipStream_t stream) {
  if (q.replay >= 0) {
    Segment &S = q.segs[q.replay];
    int idx = -1;
    if (q.rpos < S.items.size() && S.items[q.rpos].same(desc, w, stream)) idx = (int)q.rpos;
    else idx = S.index_of(desc, w, stream);
    if (idx >= 0 && S.mark(idx)) {
      ++q.n;
      q.rpos = (size_t)idx + 1;
      bump(g_q_replayed);
      return;
    }
    if (idx < 0 && S.is_terminator(desc, w, stream)) {
      bump(g_q_terminated);
      q.flush(nullptr, true); // as seen before: this invoke conflicts with the group (replay ends, the queue is empty; the launch is
      q.backoff_next = 2;     // issued once the next group is open). A whole group replayed: the caller is repeating itself
    } else if (idx < 0 && (q.close_window(), (size_t)q.n == S.items.size()) && q.find_segment(desc, w, stream, &idx) >= 0) {
      // Every member of the recorded group has arrived and this invoke belongs to ANOTHER recorded group: the group is over
      // (nothing the cache knows could still join it) - launch it and replay the invoke's own group. Flushing early is always
      // safe; what this saves is learning one terminator per distinct first arriver of the next group: with several OpenMP
      // callers the first invoke of the next layer is a different tile every iteration, and every unknown one used to cost an
      // abandoned replay plus a growing back-off (2 of 10 runs of the 8-caller benchmark spent their timed iterations learning).
      bump(g_q_terminated);
      q.flush(nullptr, true);
      q.backoff_next = 2;
    } else if (idx >= 0 && (q.close_window(), (size_t)q.n == S.items.size())) {
      // Every member has arrived and this invoke is a member AGAIN: the caller runs the same group once more - the timing loop of a
      // single-layer benchmark (benchmarks/config/matmul/*.json, fc/*.json: tpp-run calls the one-layer kernel N times; round 5:
      // every iteration used to abandon its replay here, 2017 of 2020 groups, and the rebuilt bookkeeping made the run host-bound -
      // 7.4 us per iteration of 48 invokes against 5.3 for the GPU side). Flushing early is always safe; the invoke then starts the
      // replay of its group afresh below.
      bump(g_q_terminated);
      q.flush(nullptr, true);
      q.backoff_next = 2;
    } else { // neither a member nor a known terminator: make the bookkeeping catch up with what has been queued
      bump(g_q_abandoned);
      q.close_window();
      q.materialize();
      q.learn = q.replay;
      q.learn_n = (size_t)q.n;
      q.replay = -1;
      rebuild_footprints(q, devmem);
      q.rec.items.clear();
      for (int i = 0; i < q.n; ++i) q.rec.items.push_back(TraceItem{q.desc, q.pinned[q.slot][i], q.stream});
      q.rec_open = true;
    }
  }
  if (q.n == 0 && try_start_replay(q, devmem, desc, w, stream)) {
    q.issue_pending();
    return;
  }
  q.issue_pending();
  process_item(q, devmem, desc, w, stream);
}

// The scheduler. The reference calls invoke from OpenMP workers (scf.parallel over the tile grid): with one
// lock around the dependence bookkeeping eight callers took 290 us for what one caller did in 45 (lock
// hand-offs, and interleaved callers defeat the interval merging). Callers therefore only HAND OVER their
// invokes; a single scheduler thread does the dependence bookkeeping without any lock and launches a group
// whenever the next invoke conflicts with it. Callers never touch HIP on this path; launches and slot waits
// happen on the scheduler thread, overlapped with the callers.
//
// Hand-over = one private single-producer ring per calling thread, merged by TIME STAMP. (Round 1 used one
// multi-producer ring with a ticket counter: on the 256-core host of the GPU box the counter's cache line
// hopping between the callers cost 130-230 ns per invoke - two callers took 180 us for what one did in 48.)
//   * An entry is ONE cache line: stamp, descriptor, the four operand pointers, batch count, stream; the
//     scheduler derives the footprints itself (queued_operands). A push writes that line and nothing shared.
//   * The stamp is the invariant TSC (`lfence; rdtsc`) when the kernel trusts it as its clock source, else a shared
//     counter. Either way  a happens-before b  =>  stamp(a) < stamp(b), and a is visible to whoever sees b.
//   * The scheduler keeps the non-empty rings in a min-heap on the stamp of their oldest entry and always takes
//     the smallest. A ring it finds empty stays WARM for a while: its next slot (a line in the scheduler's cache
//     until the producer writes it) is polled before every pop. A ring that stays empty for some thousand polls is
//     PARKED (flag in the ring, Dekker-style re-check); the producer's next push sees the flag and announces
//     the ring on a small wake list - the only shared write on the producer side, once per burst.
//   * Before every pop the warm rings and the wake list are polled until a whole pass finds nothing new. So when an
//     entry b is taken, every entry that happened before b is already consumed, or in the heap with a smaller stamp,
//     or behind such an entry in its own ring: the processing order respects every caller's program order and
//     every happens-before between callers (an OpenMP barrier, a join). Entries without such a relation are
//     concurrent invokes of the caller's program, and those do not conflict in a race-free program.
struct alignas(64) PSlot {
  std::atomic<uint32_t> seq; // (uint32_t)(index + 1) once the entry at `index` is complete
  int32_t br;
  uint64_t stamp;
  const void *desc; // nullptr: a fence, C = the std::atomic<int> to raise once everything before it is launched
  const void *A, *B;
  void *C;
  const void *D;
  hipStream_t stream;
};
static_assert(sizeof(PSlot) == 64, "one cache line per queued invoke");

struct PQueue {
  static constexpr uint64_t CAP = 2048, MASK = CAP - 1;
  PSlot *ring = nullptr;
  // producer side
  alignas(64) uint64_t tail = 0;
  uint64_t head_seen = 0;              // last value read from head_pub
  std::atomic<uint64_t> tail_pub{0};   // = tail, for drain()'s "anything pending?" test
  // consumer side
  alignas(64) uint64_t head = 0;       // next index to consume (owned by the live scheduler thread)
  std::atomic<uint64_t> head_pub{0};   // published every 16 entries and when the ring is parked: the producer reads it only when the ring looks full
  std::atomic<uint64_t> clean_head{0}; // every entry below this has been LAUNCHED
  // rarely written by either side
  alignas(64) std::atomic<int> parked{1}; // 1: the scheduler is not watching this ring - the next push must announce it
  std::atomic<int> owned{0};               // a caller thread holds this ring
};

struct Scheduler {
  static constexpr int MAXQ = 1024;
  std::atomic<PQueue *> queues[MAXQ];
  std::atomic<int> nq{0}; // high-water mark of allocated rings
  std::mutex alloc_mu;
  PQueue overflow; // more than MAXQ simultaneous caller threads: they share this ring under a mutex
  std::mutex overflow_mu;
  // wake list: ring indices + 1 (0 = empty cell); a ring is on it at most once, so MAXQ + 1 cells cannot overflow
  static constexpr uint32_t WCAP = 2048;
  alignas(64) std::atomic<uint32_t> wake_tail{0};
  alignas(64) std::atomic<uint32_t> wake_cell[WCAP];
  uint32_t wake_head = 0; // scheduler thread only

  const bool use_tsc;
  // Parking a ring is a Dekker pair (producer: publish entry, read `parked`; scheduler: set `parked`, re-read the slot). The
  // producer's side runs once per invoke and a full fence there stalls it on the slot line's ownership request (the line is in the
  // scheduler's cache from the previous lap: ~150 ns across cores, measured as 260 ns per invoke with two callers), so the
  // fence is moved to the side that runs once per burst: the scheduler issues membarrier(PRIVATE_EXPEDITED) - a full barrier on
  // every thread of the process - between its two steps, and the producers use plain release stores / loads. Without that
  // system call (old kernels, seccomp) the producers fall back to sequentially consistent stores.
  const bool asym_fence;
  alignas(64) std::atomic<uint64_t> stamp_ctr{1};

  std::atomic<bool> stop{false};
  std::thread worker;
  int device = 0;
  TileQueue q;
  DeviceRanges devmem; // the worker's allocation cache (per epoch, like the callers' own)
  std::vector<std::pair<uint64_t, int>> heap; // (stamp of the ring's oldest entry, ring index), min on top
  struct Warm {
    int qi;
    unsigned polls;
  };
  std::vector<Warm> warm; // rings found empty a moment ago
  static constexpr unsigned PARK_AFTER = 4096; // polls without an entry before a warm ring is parked

  // The worker exists only while there is traffic: after ~2 s without an entry it leaves (a library that was used
  // once must not keep a thread napping for the rest of the process), and the next push starts a new one. The
  // hand-over is a Dekker pair on (running, wake list): the worker clears `running` BEFORE it re-reads the wake list
  // (every ring is parked while the worker idles, so every push goes through that list), a producer announces its
  // ring BEFORE it reads `running` - at least one of them sees the other.
  alignas(64) std::atomic<bool> running{false}; // read by every producer on every push: its own cache line, written twice in a worker's life
  alignas(64) std::mutex life_mu;

  static bool kernel_trusts_tsc() {
    char buf[32] = {0};
    if (FILE *f = fopen("/sys/devices/system/clocksource/clocksource0/current_clocksource", "r")) {
      if (!fgets(buf, sizeof(buf), f)) buf[0] = 0;
      fclose(f);
    }
    return strncmp(buf, "tsc", 3) == 0;
  }
  static bool register_membarrier() {
    if (getenv("TPP_HIP_NO_MEMBARRIER")) return false;
    return syscall(__NR_membarrier, MEMBARRIER_CMD_REGISTER_PRIVATE_EXPEDITED, 0) == 0;
  }
  Scheduler() : use_tsc(kernel_trusts_tsc() && !getenv("TPP_HIP_NO_TSC")), asym_fence(register_membarrier()) {
    for (auto &c : queues) c.store(nullptr, std::memory_order_relaxed);
    for (auto &c : wake_cell) c.store(0, std::memory_order_relaxed);
    init_ring(overflow);
    if (hipGetDevice(&device) != hipSuccess) device = 0;
  }
  ~Scheduler() {
    stop.store(true);
    std::thread w;
    {
      std::lock_guard<std::mutex> lk(life_mu);
      w = std::move(worker);
    }
    if (!w.joinable()) return;
    // a fatal error on the scheduler thread itself exits the process from that thread: never join yourself
    if (w.get_id() == std::this_thread::get_id()) w.detach();
    else w.join(); // outside life_mu: a worker on its way out takes that lock
  }
  static void init_ring(PQueue &Q) {
    Q.ring = static_cast<PSlot *>(aligned_alloc(64, sizeof(PSlot) * PQueue::CAP));
    if (!Q.ring) die("tpp-xsmm-hip: out of memory for a caller's invoke ring");
    for (uint64_t i = 0; i < PQueue::CAP; ++i) new (&Q.ring[i].seq) std::atomic<uint32_t>(0);
  }
  uint64_t stamp() {
#if defined(__x86_64__)
    if (use_tsc) {
      unsigned lo, hi;
      asm volatile("lfence\n\trdtsc" : "=a"(lo), "=d"(hi)::"memory"); // after every earlier load (the caller's synchronisation) has completed
      return ((uint64_t)hi << 32) | lo;
    }
#endif
    return stamp_ctr.fetch_add(1, std::memory_order_seq_cst);
  }
  PQueue *ring_at(int i) { return i == MAXQ ? &overflow : queues[i].load(std::memory_order_acquire); }

  // ---- caller side -------------------------------------------------------------------------------------------
  // the calling thread's ring: claimed on first use, handed back when the thread ends (entries still in it stay
  // valid; the next owner continues at its tail)
  struct Lease {
    Scheduler *s = nullptr;
    int idx = -1;
    ~Lease() {
      if (s && idx >= 0 && idx < MAXQ) s->queues[idx].load(std::memory_order_relaxed)->owned.store(0, std::memory_order_release);
    }
  };
  int claim() {
    const int n = nq.load(std::memory_order_acquire);
    for (int i = 0; i < n; ++i) {
      PQueue *Q = queues[i].load(std::memory_order_acquire);
      int expect = 0;
      if (Q && Q->owned.load(std::memory_order_relaxed) == 0 && Q->owned.compare_exchange_strong(expect, 1, std::memory_order_acq_rel)) return i;
    }
    std::lock_guard<std::mutex> lk(alloc_mu);
    const int m = nq.load(std::memory_order_relaxed);
    if (m >= MAXQ) return MAXQ; // the shared overflow ring
    PQueue *Q = new PQueue;
    init_ring(*Q);
    Q->owned.store(1, std::memory_order_relaxed);
    queues[m].store(Q, std::memory_order_release);
    nq.store(m + 1, std::memory_order_release);
    return m;
  }
  int my_ring() {
    thread_local Lease lease;
    if (lease.s != this) {
      lease.s = this;
      lease.idx = claim();
    }
    return lease.idx;
  }
  void ensure_worker() {
    if (running.load(std::memory_order_seq_cst)) return;
    std::lock_guard<std::mutex> lk(life_mu);
    if (running.load(std::memory_order_relaxed) || stop.load()) return;
    if (worker.joinable()) worker.join(); // the previous worker has left (it cleared `running` on its way out)
    running.store(true, std::memory_order_seq_cst);
    worker = std::thread([this] { run(); });
  }
  void push_to(int qi, PQueue &Q, const QEntry &e) {
    const uint64_t h = Q.tail;
    if (h - Q.head_seen >= PQueue::CAP) {
      // Ring full: this caller outruns the scheduler. Wait until HALF of it is free again, not for one slot: the scheduler
      // then streams through a backlog of finished (prefetched) lines while the producer refills in a burst, instead of
      // the two moving in lockstep with every line crossing cores just in time.
      for (unsigned spins = 0; h - (Q.head_seen = Q.head_pub.load(std::memory_order_acquire)) > PQueue::CAP / 2; ++spins) {
        if (stop.load(std::memory_order_relaxed)) return; // the process is exiting (static destruction): nobody will consume the ring
        if (spins < 2000) cpu_relax();
        else {
          ensure_worker();
          sched_yield();
        }
      }
    }
    PSlot &s = Q.ring[h & PQueue::MASK];
    s.br = (int32_t)e.w.br;
    s.desc = e.desc;
    s.A = e.w.A;
    s.B = e.w.B;
    s.C = e.w.C;
    s.D = e.w.D;
    s.stream = e.stream;
    s.stamp = stamp();
    s.seq.store((uint32_t)(h + 1), asym_fence ? std::memory_order_release : std::memory_order_seq_cst); // Dekker with `parked`, see asym_fence
    Q.tail = h + 1;
    Q.tail_pub.store(h + 1, std::memory_order_relaxed);
    if (Q.parked.load(std::memory_order_seq_cst) && Q.parked.exchange(0, std::memory_order_seq_cst)) {
      const uint32_t pos = wake_tail.fetch_add(1, std::memory_order_seq_cst);
      std::atomic<uint32_t> &cell = wake_cell[pos % WCAP];
      while (cell.load(std::memory_order_acquire) != 0) cpu_relax(); // (a lap behind: cannot happen with <= MAXQ + 1 rings)
      cell.store((uint32_t)qi + 1, std::memory_order_seq_cst);
    }
    ensure_worker();
  }
  void push(const QEntry &e) {
    if (e.w.br > 0x7fffffff) die("tpp-xsmm-hip: batch count %ld is too large for the tile queue", (long)e.w.br);
    const int qi = my_ring();
    if (qi == MAXQ) {
      std::lock_guard<std::mutex> lk(overflow_mu);
      push_to(qi, overflow, e);
    } else {
      push_to(qi, *queues[qi].load(std::memory_order_relaxed), e);
    }
  }
  // everything pushed before this call (by this thread, or by another with a happens-before to this call) has been
  // launched on return
  void drain() {
    bool pending = false;
    const int n = nq.load(std::memory_order_acquire);
    for (int i = 0; i <= n && !pending; ++i) {
      PQueue *Q = i == n ? &overflow : queues[i].load(std::memory_order_acquire);
      pending = Q && Q->clean_head.load(std::memory_order_acquire) < Q->tail_pub.load(std::memory_order_acquire);
    }
    if (!pending || stop.load(std::memory_order_relaxed)) return;
    std::atomic<int> flag{0};
    QEntry f;
    f.w.C = &flag; // desc == nullptr: a fence
    push(f);
    for (unsigned spins = 0; !flag.load(std::memory_order_acquire); ++spins) {
      // the scheduler is being destroyed (exit() on another thread while this one flushes): its worker will not start again
      // (ensure_worker) and the fence would never be raised - give up instead of spinning through process teardown
      if (stop.load(std::memory_order_relaxed) && !running.load(std::memory_order_seq_cst)) return;
      if (spins < 4000) cpu_relax();
      else sched_yield();
    }
  }

  // ---- scheduler thread -----------------------------------------------------------------------------------------
  static bool later(const std::pair<uint64_t, int> &a, const std::pair<uint64_t, int> &b) { return a.first > b.first; }
  bool take_if_ready(int qi, PQueue &Q) { // the ring's next slot: into the heap with it if it is complete
    PSlot &s = Q.ring[Q.head & PQueue::MASK];
    if (s.seq.load(std::memory_order_acquire) != (uint32_t)(Q.head + 1)) return false;
    heap.emplace_back(s.stamp, qi);
    std::push_heap(heap.begin(), heap.end(), later);
    return true;
  }
  void examine(int qi, PQueue &Q) { // after a pop / a wake-up: heap or warm list
    if (!take_if_ready(qi, Q)) warm.push_back(Warm{qi, 0});
  }
  bool park(int qi, PQueue &Q) { // true: an entry slipped in and is in the heap now
    PSlot &s = Q.ring[Q.head & PQueue::MASK];
    Q.head_pub.store(Q.head, std::memory_order_release);
    Q.parked.store(1, std::memory_order_seq_cst);
    if (asym_fence && syscall(__NR_membarrier, MEMBARRIER_CMD_PRIVATE_EXPEDITED, 0) != 0) die("tpp-xsmm-hip: membarrier failed");
    if (s.seq.load(std::memory_order_seq_cst) == (uint32_t)(Q.head + 1) && Q.parked.exchange(0, std::memory_order_seq_cst)) {
      // an entry arrived while the ring was being parked and its producer has not taken the flag: it is ours again
      // (if the producer took the flag, the ring comes back through the wake list)
      heap.emplace_back(s.stamp, qi);
      std::push_heap(heap.begin(), heap.end(), later);
      return true;
    }
    return false;
  }
  bool drain_wake_list() {
    bool any = false;
    for (;;) {
      std::atomic<uint32_t> &cell = wake_cell[wake_head % WCAP];
      uint32_t v = cell.load(std::memory_order_seq_cst);
      if (!v) {
        // Producers RESERVE a cell (fetch_add on wake_tail) and fill it afterwards: an empty cell below the reserved tail is a
        // producer between its two steps. A later cell - or a warm ring - may already hold an entry that happened AFTER that
        // producer's push (it saw the push through a barrier), so stopping here would let that entry overtake it. Wait for the
        // laggard: the window is a few instructions unless the producer was preempted inside it.
        if (wake_tail.load(std::memory_order_seq_cst) == wake_head) return any;
        while (!(v = cell.load(std::memory_order_acquire))) cpu_relax();
      }
      cell.store(0, std::memory_order_release);
      ++wake_head;
      any = true;
      examine((int)v - 1, *ring_at((int)v - 1));
    }
  }
  // one pass over the warm rings; true if an entry turned up. count: this pass counts towards parking
  bool poll_warm(bool count) {
    bool any = false;
    for (size_t i = 0; i < warm.size();) {
      PQueue &Q = *ring_at(warm[i].qi);
      if (take_if_ready(warm[i].qi, Q)) {
        any = true;
      } else if (count && ++warm[i].polls > PARK_AFTER) {
        any = park(warm[i].qi, Q) || any;
      } else {
        ++i;
        continue;
      }
      warm[i] = warm.back();
      warm.pop_back();
    }
    return any;
  }
  // everything that happened before any entry now in the heap is consumed, in the heap, or behind a heap entry of its ring
  void collect() {
    bool any = drain_wake_list();
    any = poll_warm(true) || any;
    while (any) { // an entry turned up: whatever happened before IT was published earlier - look again
      any = drain_wake_list();
      any = poll_warm(false) || any;
    }
  }
  void mark_clean() { // everything consumed so far has been launched
    const int n = nq.load(std::memory_order_acquire);
    for (int i = 0; i <= n; ++i) {
      PQueue *Q = i == n ? &overflow : queues[i].load(std::memory_order_acquire);
      if (Q) Q->clean_head.store(Q->head, std::memory_order_release);
    }
  }
  void run() {
    // the creating thread may be pinned (OMP_PROC_BIND pins each worker to one core): inheriting that mask would
    // put the scheduler on the caller's own core. Use the mask the PROCESS had when the library was loaded
    // (taskset / numactl / cgroup limits are respected; only later per-thread pinning is undone).
    if (g_have_process_mask) (void)sched_setaffinity(0, sizeof(g_process_mask), &g_process_mask);
    (void)hipSetDevice(device);
    unsigned idle = 0;
    for (;;) {
      collect();
      if (!heap.empty()) {
        std::pop_heap(heap.begin(), heap.end(), later);
        const int qi = heap.back().second;
        heap.pop_back();
        PQueue &Q = *ring_at(qi);
        const PSlot &s = Q.ring[Q.head & PQueue::MASK];
        QEntry e;
        e.desc = s.desc;
        e.w = WorkItem{s.A, s.B, s.C, s.D, s.br};
        e.stream = s.stream;
        __builtin_prefetch(&Q.ring[(Q.head + 4) & PQueue::MASK]);
        __builtin_prefetch(&Q.ring[(Q.head + 8) & PQueue::MASK]);
        ++Q.head;
        if ((Q.head & 15) == 0) Q.head_pub.store(Q.head, std::memory_order_release);
        examine(qi, Q);
        idle = 0;
        if (!e.desc) {
          q.flush();
          mark_clean();
          ((std::atomic<int> *)e.w.C)->store(1, std::memory_order_release);
        } else {
          devmem.refresh();
          submit_item(q, devmem, e.desc, e.w, e.stream);
        }
        continue;
      }
      if (stop.load(std::memory_order_relaxed)) break;
      if (++idle < 4000) cpu_relax();
      else if (idle < 20000) sched_yield();
      else { // nothing for a long while: stop burning a core (a caller that arrives now waits one nap)
        timespec ts{0, idle < 40000 ? 50000 : 1000000}; // 50 us naps, then 1 ms naps
        nanosleep(&ts, nullptr);
        if (idle > 42000) { // ~2 s of 1 ms naps: leave, unless a producer has announced a ring meanwhile
          std::lock_guard<std::mutex> lk(life_mu); // ensure_worker() joins this thread under the same lock: decide inside it
          if (!warm.empty()) continue; // (every ring must be parked before the worker may leave)
          running.store(false, std::memory_order_seq_cst);
          if (!drain_wake_list()) return;
          running.store(true, std::memory_order_seq_cst);
          idle = 0;
        }
      }
    }
  }
};
std::atomic<Scheduler *> g_sched{nullptr};
std::mutex g_sched_mu;
Scheduler &sched() {
  Scheduler *p = g_sched.load(std::memory_order_acquire);
  if (!p) {
    std::lock_guard<std::mutex> lk(g_sched_mu);
    p = g_sched.load(std::memory_order_relaxed);
    if (!p) {
      static Scheduler the_scheduler; // destroyed (worker joined) at process exit
      p = &the_scheduler;
      g_sched.store(p, std::memory_order_release);
    }
  }
  return *p;
}
// Three ways into the queue state. DIRECT: a member of the recorded group that is being replayed is marked by its caller without
// any lock (DirectWindow) - the steady state of compiled code that repeats itself, from one thread or from the reference's OpenMP
// team alike. INLINE: everything else takes a spin lock and does the bookkeeping itself (45 ns per invoke for one caller).
// SCHEDULED: if several threads keep arriving on the locked path - a program the trace cache does not help, where one lock
// around the bookkeeping serialises the callers - the process switches, once and for good, to the rings + scheduler thread above
// (xsmm_hip_set_tile_queue(2) / TPP_HIP_TILE_QUEUE=2: as soon as a second thread shows up, the round-2 behaviour).
struct SpinLock {
  std::atomic<int> f{0};
  void lock() {
    for (unsigned spins = 0;; ++spins) {
      if (f.load(std::memory_order_relaxed) == 0 && f.exchange(1, std::memory_order_acquire) == 0) return; // (waiters spin on a shared line)
      if (spins < 4000) cpu_relax();
      else sched_yield();
    }
  }
  void unlock() { f.store(0, std::memory_order_release); }
};
struct InlineQueue {
  SpinLock mu;
  TileQueue q;
  DirectWindow dw;
  std::atomic<bool> scheduled{false}; // one-way switch, flipped under mu after q has been flushed
  uint64_t owner = 0;                 // thread that queued last (under mu)
  int foreign = 0;                    // arrivals of other threads since the last flush point (tile-queue mode 2)
  bool multi = false;                 // more than one thread has queued
  int64_t slow = 0, groups_at = 0;    // locked arrivals since a group was last replayed through the window / q.direct_groups then
  InlineQueue() { q.dw = &dw; }
};
InlineQueue &inl() {
  static InlineQueue i;
  return i;
}
std::atomic<int> g_dt_pending{0}; // number of remembered transposes (see "deferred transposes" below)
void dt_materialize();
void flush_tile_queue() {
  if (g_dt_pending.load(std::memory_order_acquire)) dt_materialize();
  if (!cfg().tile_queue.load(std::memory_order_relaxed)) return;
  InlineQueue &iq = inl();
  if (!iq.scheduled.load(std::memory_order_acquire)) {
    iq.dw.touch(thread_token());
    std::lock_guard<SpinLock> lk(iq.mu);
    if (!iq.scheduled.load(std::memory_order_relaxed)) {
      iq.q.flush();
      iq.foreign = 0;
      return;
    }
  }
  if (Scheduler *p = g_sched.load(std::memory_order_acquire)) p->drain();
}

// Queues one invoke of `desc`; true if queued (nothing launched yet), false if an operand is host memory (the
// caller flushes and takes the mirrored path). `ptrs` are the item's non-null operand pointers.
// The tile queue serves ONE device per process: the scheduler thread binds to the device of the first caller, work lists are
// plain pinned allocations and the tile heuristics cache that device's CU count. A caller on another device would get its
// grouped launches issued on the wrong GPU - refuse loudly instead (checked once per thread and synchronisation epoch, not per
// invoke). Non-queued invokes launch from the calling thread and follow its current device as usual.
std::atomic<int> g_queue_device{-1};
void check_queue_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return; // no device: the launch itself will fail loudly
  }
  int expect = -1;
  if (!g_queue_device.compare_exchange_strong(expect, dev) && expect != dev)
    die("tpp-xsmm-hip: the tile queue serves one device per process (first used on device %d, this thread's current device is %d); "
        "turn the queue off (xsmm_hip_set_tile_queue(0)) for multi-device processes", expect, dev);
}

// everything a calling thread keeps for the enqueue path, behind ONE thread-local lookup per invoke (in a shared library every
// thread_local access is a call into the dynamic TLS resolver)
struct CallerState;
// One pointer in the static TLS block (initial-exec: a %fs-relative load; the general-dynamic model of a shared library calls
// __tls_get_addr on every access - 10-15 cycles of an invoke), the state itself behind the usual thread_local so that it is
// destroyed with its thread. 8 bytes of the loader's static-TLS reserve: dlopen-safe.
// -DTPP_TLS_DEFAULT_MODEL: the compiler's default model for a shared object instead - for a process whose static-TLS
// surplus is already spent by other initial-exec libraries when this one is dlopen'ed ("cannot allocate memory in static TLS block").
#ifdef TPP_TLS_DEFAULT_MODEL
static __thread CallerState *tl_fast = nullptr;
#else
static __thread CallerState *tl_fast __attribute__((tls_model("initial-exec"))) = nullptr;
#endif
void dt_release_slot(int slot);
struct CallerState {
  DeviceRanges devmem; // per caller: no sharing, no lock
  DirectWindow::Caller *me = nullptr;
  bool claimed = false;
  int dt_slot = -1; // this thread's slot of remembered transposes ("deferred transposes" below)
  ~CallerState() {
    tl_fast = nullptr;
    if (me) me->owned.store(0, std::memory_order_release);
    if (dt_slot >= 0) dt_release_slot(dt_slot);
  }
};

static __attribute__((noinline)) CallerState &caller_state_slow() {
  thread_local CallerState tl;
  tl_fast = &tl;
  return tl;
}
static inline CallerState &caller_state() {
  CallerState *p = tl_fast;
  return p ? *p : caller_state_slow();
}

bool enqueue_item(const void *desc, const WorkItem &item, const void *const *ptrs, int n_ptrs, hipStream_t s) {
  CallerState &tl = caller_state();
  DeviceRanges &devmem = tl.devmem;
  static InlineQueue &iq = inl();
  // DIRECT: the invoke is a member of the recorded group being replayed. proven: only if the group's pointers have been proven
  // device memory in this epoch (Segment::prove) - the caller has not looked at its operands yet
  auto join_window = [&](bool proven, uint64_t epoch) __attribute__((always_inline)) -> bool {
    const uint64_t c = iq.dw.cur.load(std::memory_order_acquire);
    if (!c) return false;
    if (!tl.claimed) {
      tl.claimed = true;
      tl.me = iq.dw.claim();
    }
    DirectWindow::Caller *me = tl.me;
    if (!me) return false;
    // BRACKET FIRST: seq goes odd BEFORE `multi` is read, with a compiler barrier in between. The switching thread
    // sets multi, issues membarrier (an IPI = a full barrier at a precise point of this thread's instruction stream) and then waits
    // for an even seq. Interrupts are precise: either the seq store had retired when the IPI landed - then it is visible behind the
    // barrier and the switcher waits for this section to end -, or it had not - then the load of `multi` below had not retired
    // either, is re-executed behind the barrier and sees multi == true. (Round 4 read `multi` first: an IPI between the two
    // instructions let the switcher see an even seq while this thread went on into a solo section.)
    const uint64_t seq0 = me->seq.load(std::memory_order_relaxed);
    me->seq.store(seq0 + 1, std::memory_order_relaxed);
    std::atomic_signal_fence(std::memory_order_seq_cst);
    const bool solo = !iq.dw.multi.load(std::memory_order_relaxed); // (a thread that holds a slot and sees solo IS the one thread)
    if (solo) {
      me->busy.store(c, std::memory_order_relaxed);
      std::atomic_signal_fence(std::memory_order_seq_cst); // the compiler keeps busy-store, cur-load in this order (the hardware needs no fence: one thread)
    } else {
      me->seq.store(seq0 + 2, std::memory_order_release); // not solo after all: the bracket closes, the two-sided protocol from here
      me->busy.store(c, std::memory_order_seq_cst);
    }
    bool joined = false;
    if (iq.dw.cur.load(solo ? std::memory_order_relaxed : std::memory_order_seq_cst) == c) {
      Segment &S = iq.q.segs[(c & 127) - 1];
      if (!proven || S.dev_epoch == epoch) {
        if (me->tag != c) {
          me->tag = c;
          me->count = 0;
        }
        int idx = -1;
        if (me->hint < S.items.size() && S.items[me->hint].same(desc, item, s)) idx = (int)me->hint;
        else idx = S.index_of(desc, item, s);
        if (idx >= 0 && (solo ? S.mark_solo(idx) : S.mark(idx))) {
          ++me->count;
          me->hint = (uint32_t)idx + 1;
          joined = true;
        }
      }
    }
    me->busy.store(0, std::memory_order_release);
    if (solo) {
      std::atomic_signal_fence(std::memory_order_seq_cst);
      me->seq.store(seq0 + 2, std::memory_order_release); // even again: the solo section is over
    }
    return joined;
  };
  const uint64_t epoch = g_devmem_epoch.load(std::memory_order_relaxed);
  if (devmem.epoch == epoch && join_window(true, epoch)) return true; // (this thread has been through the checks below in this epoch)
  if (devmem.refresh()) check_queue_device();
  for (int i = 0; i < n_ptrs; ++i)
    if (!devmem.is_device(ptrs[i], i)) return false;
  if (join_window(false, 0)) return true;
  if (!iq.scheduled.load(std::memory_order_acquire)) {
    iq.dw.touch(thread_token());
    std::lock_guard<SpinLock> lk(iq.mu);
    if (!iq.scheduled.load(std::memory_order_relaxed)) {
      const uint64_t me = (uint64_t)(uintptr_t)&devmem; // the address of this thread's cache identifies the thread (one TLS lookup per invoke, not two)
      if (iq.owner != me) {
        if (iq.owner != 0) iq.multi = true;
        if (iq.owner != 0 && cfg().tile_queue.load(std::memory_order_relaxed) == 2 && ++iq.foreign > 4) iq.slow = 1 << 30;
        iq.owner = me;
      }
      if (iq.q.direct_groups != iq.groups_at) { // a group went through the window since the last look: the cache is working
        iq.groups_at = iq.q.direct_groups;
        iq.slow = 0;
      }
      if (iq.multi && ++iq.slow > 8192) { // several threads, and the locked path is where they meet: hand over to the scheduler
        iq.q.flush();
        (void)sched(); // create it (its worker thread starts with the first entry)
        iq.scheduled.store(true, std::memory_order_release);
      } else {
        submit_item(iq.q, devmem, desc, item, s);
        return true;
      }
    }
  }
  QEntry e;
  e.desc = desc;
  e.w = item;
  e.stream = s;
  sched().push(e);
  return true;
}

bool queue_active() {
  return cfg().tile_queue.load(std::memory_order_relaxed) && cfg().async.load(std::memory_order_relaxed);
}

bool try_enqueue(const GemmDesc *d, void *a, void *b, void *c, void *dp, int64_t br, hipStream_t s) {
  if (d->m > 64 || d->n > 64) return false; // big descriptors fill the chip on their own
  const void *ptrs[4] = {a, b, c, dp};
  return enqueue_item(d, WorkItem{a, b, c, dp, br}, ptrs, 4, s);
}

// ---- deferred transposes (round 5) ------------------------------------------------------------------------------------
// A contraction whose B operand is transposed in memory reaches the runtime as TWO invokes per tile: xsmm.unary transpose into a
// small temporary, then xsmm.gemm reading it (ConvertLinalgToXsmm; test/Conversion/LinalgToXsmm/linalg-to-gemm.mlir:46-62 has the
// query-times-key benchmark lowered exactly so: transpose [32,64,512,32] + gemm [32,32,64,512,32,32] per (batch, head), ONE
// temporary for every tile of a caller). Through the tile queue that is a chain of true and anti dependences on the temporary:
// every invoke its own launch (1024 launches for benchmarks/mlir/fp32-query-times-key.mlir, 3.6 ms; the queue cannot help).
// So a transpose of a small tile into a DENSE destination (ldo = m) is not launched when it is invoked but REMEMBERED - one record
// per calling thread (the reference's OpenMP callers own a temporary each) - and
//   * a gemm of the same thread whose B operand is exactly that destination (k = the transpose's n, n = its m, ldb = ldo, one batch
//     element, f32, no operand of it overlapping the destination, C not overlapping the transpose's source) runs on a SIBLING
//     descriptor that reads B transposed straight from the transpose's source (GemmDesc::b_trans - the generic kernel). All such
//     gemms of a loop, of every thread, share that sibling: the queue groups them into one launch;
//   * a second transpose of the same thread, the same descriptor and the same destination REPLACES the record: the remembered one is
//     dead - fully overwritten, and its only readers were served from its source;
//   * any other invoke of the owning thread launches the remembered transpose first, the ordinary way (dt_launch); an invoke of
//     ANOTHER thread does so if one of its operands overlaps the record's destination, or if it writes into the record's source (a
//     race-free program orders such an invoke behind the transpose's invoke: it then sees the record); a flush and every
//     synchronisation point launch every record - the destination holds what the program wrote whenever anything can look at it.
// Between a transpose's invoke and its launch only folded gemms of its own thread and invokes that touch neither its destination
// nor (writing) its source run: the deferred launch reads what the immediate one would have read.
// Summation order of a folded gemm = the generic kernel's (what a single invoke of the same gemm on the generic kernel adds).
struct DeferredTranspose {
  const UnaryDesc *d = nullptr;
  void *src = nullptr, *dst = nullptr;
  hipStream_t stream = nullptr;
  const GemmDesc *sib_of = nullptr, *sib = nullptr; // the last gemm descriptor folded and its sibling
};
struct alignas(64) DtSlot {
  // line 0 - what EVERY thread reads per invoke while records exist; written when a record appears or goes, not per tile:
  std::atomic<uintptr_t> owner{0};   // thread_token() of the thread that owns the slot (0: free)
  std::atomic<int> live{0};          // a record is remembered
  // the record's destination, and the hull of the sources it has had (the source changes with every tile of a loop - the next
  // transpose replaces the record -, the hull stops growing after one pass: eight callers that each rewrote a line the seven others
  // read per invoke took 2.5 us per tile). For the other threads' overlap test: written under mu before live = 1 (release), read after live
  // (acquire). A reader that races with a replacement may see either record's source range - both belong to invokes it is not ordered with.
  std::atomic<uintptr_t> d_lo{0}, d_hi{0}, s_lo{0}, s_hi{0};
  // line 1 - the owner's (and, rarely, of a thread that launches the record):
  alignas(64) SpinLock mu;           // the record and its hand-over
  DeferredTranspose r;               // under mu
  std::atomic<int64_t> folded{0}, dropped{0}; // statistics (the owner's relaxed adds)
};
constexpr int DT_SLOTS = 64;
DtSlot g_dt_slots[DT_SLOTS];
std::atomic<int> g_dt_top{0}; // slots [0, top) have been claimed at some time
std::atomic<int64_t> g_dt_launched{0}; // statistics (xsmm_hip_fold_transpose_stats; folded / dropped: per slot)
static __thread bool tl_dt_busy = false; // this thread is inside dt_launch's hand-over (its own flush_tile_queue calls must not re-enter)
void unary_invoke_core(const UnaryDesc *d, void *pi, float scalar, bool use_scalar, void *po, bool may_defer);
// Launches the slot's remembered transpose, if there is one (any thread). The record stays live until the transpose HAS BEEN handed to
// the queue / launched, and the lock is held across that: a thread that then sees live = 0 (and goes on to queue an invoke that reads
// the destination) is ordered behind the transpose.
void dt_launch(DtSlot &sl) {
  if (tl_dt_busy) return;
  std::lock_guard<SpinLock> lk(sl.mu);
  if (!sl.live.load(std::memory_order_relaxed)) return;
  const DeferredTranspose r = sl.r;
  g_dt_launched.fetch_add(1, std::memory_order_relaxed);
  if (cfg().stream.load(std::memory_order_relaxed) != r.stream) die("tpp-xsmm-hip: a deferred transpose outlived its stream"); // (xsmm_hip_set_stream flushes first)
  tl_dt_busy = true;
  unary_invoke_core(r.d, r.src, 0.0f, false, r.dst, false);
  tl_dt_busy = false;
  sl.live.store(0, std::memory_order_release);
  g_dt_pending.fetch_sub(1, std::memory_order_release);
}
void dt_materialize() { // every record (flush, synchronisation points)
  if (tl_dt_busy) return;
  const int top = g_dt_top.load(std::memory_order_acquire);
  for (int i = 0; i < top; ++i)
    if (g_dt_slots[i].live.load(std::memory_order_acquire)) dt_launch(g_dt_slots[i]);
}
struct DtRange {
  uintptr_t lo, hi;
};
inline DtRange dt_range(const void *p, size_t bytes) { return DtRange{(uintptr_t)p, p ? (uintptr_t)p + bytes : 0}; }
inline bool dt_overlap(const void *a, size_t na, const void *b, size_t nb) {
  return a && b && na && nb && (uintptr_t)a < (uintptr_t)b + nb && (uintptr_t)b < (uintptr_t)a + na;
}
// the records of OTHER threads that this invoke (reads rd[0..nr), writes wr[0..nw)) must see launched
void dt_scan_foreign(const DtSlot *mine, const DtRange *rd, int nr, const DtRange *wr, int nw) {
  const int top = g_dt_top.load(std::memory_order_acquire);
  for (int i = 0; i < top; ++i) {
    DtSlot &sl = g_dt_slots[i];
    if (&sl == mine || !sl.live.load(std::memory_order_acquire)) continue;
    const uintptr_t dl = sl.d_lo.load(std::memory_order_relaxed), dh = sl.d_hi.load(std::memory_order_relaxed);
    const uintptr_t slo = sl.s_lo.load(std::memory_order_relaxed), shi = sl.s_hi.load(std::memory_order_relaxed);
    bool hit = false;
    for (int k = 0; k < nr && !hit; ++k) hit = rd[k].lo < dh && dl < rd[k].hi;
    for (int k = 0; k < nw && !hit; ++k) hit = (wr[k].lo < dh && dl < wr[k].hi) || (wr[k].lo < shi && slo < wr[k].hi);
    if (hit) dt_launch(sl);
  }
}
// the owning thread ends: the slot is free for another thread once its record (if any) has been launched by a flush
void dt_release_slot(int slot) { g_dt_slots[slot].owner.store(0, std::memory_order_release); }
DtSlot *dt_my_slot(bool claim) {
  CallerState &tl = caller_state();
  if (tl.dt_slot >= 0) return &g_dt_slots[tl.dt_slot];
  if (!claim) return nullptr;
  const uintptr_t me = thread_token();
  for (int i = 0; i < DT_SLOTS; ++i) {
    DtSlot &sl = g_dt_slots[i];
    uintptr_t none = 0;
    if (sl.owner.load(std::memory_order_relaxed) == 0 && !sl.live.load(std::memory_order_acquire) && sl.owner.compare_exchange_strong(none, me)) {
      int top = g_dt_top.load(std::memory_order_relaxed);
      while (top < i + 1 && !g_dt_top.compare_exchange_weak(top, i + 1, std::memory_order_release)) {
      }
      tl.dt_slot = i;
      return &sl;
    }
  }
  return nullptr; // more transposing threads than slots: this one's transposes are launched as they come
}
const GemmDesc *dt_sibling(const GemmDesc *d, int64_t ld_src) {
  std::vector<int64_t> key = {KIND_GEMM, -29, (int64_t)(uintptr_t)d, ld_src};
  return (const GemmDesc *)intern(key, [&]() {
    GemmDesc *e = new GemmDesc(*d);
    e->b_trans = 1;
    e->ldb = ld_src;
    e->variant = GEMM_VARIANT_GENERIC;
    e->generic_forced = 1;
    snprintf(e->name, sizeof(e->name), "brgemm_grouped(generic), B read transposed");
    snprintf(e->trace, sizeof(e->trace), "gemm[%ld,%ld,%ld,%ld,(%ld)^T,%ld] dt%ld flags%ld %s (transpose folded)", (long)d->m, (long)d->n, (long)d->k,
             (long)d->lda, (long)ld_src, (long)d->ldc, (long)d->dtype, (long)d->wire_flags, e->name);
    return (void *)e;
  });
}
// a gemm invoke while transposes are remembered: the sibling descriptor + the transpose's source if it folds into this thread's record
// (which stays), else nullptr - this thread's record, and every other thread's record the gemm's operands touch, launched first
const GemmDesc *dt_gemm(const GemmDesc *d, void *pa, void *pb, void *pc, void *pd, int64_t br, hipStream_t s, void **src) {
  const size_t es = esize(d->dtype);
  DtSlot *mine = dt_my_slot(false);
  const GemmDesc *sib = nullptr;
  if (mine && mine->live.load(std::memory_order_acquire)) {
    {
      std::lock_guard<SpinLock> lk(mine->mu);
      if (mine->live.load(std::memory_order_relaxed)) {
        DeferredTranspose &r = mine->r;
        const UnaryDesc *t = r.d;
        const size_t dst_bytes = (size_t)t->n * t->m * 4, src_bytes = span(t->m, t->ldi, t->n) * 4;
        if (pb == r.dst && br == 1 && d->dtype == DT_F32 && !d->vnni_b && !d->vnni_c && !d->b_trans && d->k == t->n && d->n == t->m && d->ldb == t->ldo &&
            s == r.stream && d->m <= 64 && d->n <= 64 && queue_active() && !dt_overlap(pa, span(d->m, d->lda, d->k) * 4, r.dst, dst_bytes) &&
            !dt_overlap(pc, span(d->m, d->ldc, d->n) * 4, r.dst, dst_bytes) && !dt_overlap(pd, d->bias ? (size_t)d->n * 4 : 0, r.dst, dst_bytes) &&
            !dt_overlap(pc, span(d->m, d->ldc, d->n) * 4, r.src, src_bytes)) {
          if (r.sib_of != d) {
            r.sib = dt_sibling(d, t->ldi);
            r.sib_of = d;
          }
          *src = r.src;
          sib = r.sib;
          mine->folded.store(mine->folded.load(std::memory_order_relaxed) + 1, std::memory_order_relaxed);
        }
      }
    }
    if (!sib) dt_launch(*mine);
  }
  if (g_dt_pending.load(std::memory_order_relaxed) > (sib ? 1 : 0)) { // other threads' records
    const GemmDesc *e = sib ? sib : d;
    const void *b = sib ? *src : pb;
    const int64_t vf = e->vnni_factor ? e->vnni_factor : 2;
    const size_t bspan = e->vnni_b ? span((e->k + vf - 1) / vf, vf * e->ldb, vf * e->n) : e->b_trans ? span(e->n, e->ldb, e->k) : span(e->k, e->ldb, e->n);
    const size_t nb = br > 0 ? (size_t)(br - 1) : 0;
    const DtRange rd[4] = {dt_range(pa, (nb * e->stride_a + span(e->m, e->lda, e->k)) * es), dt_range(b, (nb * e->stride_b + bspan) * es),
                           dt_range(pd, e->bias ? (size_t)e->n * es : 0), dt_range(pc, span(e->m, e->ldc, e->n) * es * (e->vnni_c ? 2 : 1))};
    dt_scan_foreign(mine, rd, 4, rd + 3, 1);
  }
  return sib;
}
// any other invoke while transposes are remembered: this thread's record first, then the other threads' records it touches
void dt_other(const void *const *reads, const size_t *read_bytes, int nr, const void *out, size_t out_bytes) {
  if (DtSlot *mine = dt_my_slot(false)) {
    if (mine->live.load(std::memory_order_acquire)) dt_launch(*mine);
  }
  if (g_dt_pending.load(std::memory_order_relaxed) == 0) return;
  DtRange rd[3], wr[1] = {dt_range(out, out_bytes)};
  for (int i = 0; i < nr && i < 3; ++i) rd[i] = dt_range(reads[i], read_bytes[i]);
  dt_scan_foreign(nullptr, rd, nr < 3 ? nr : 3, wr, 1);
}
// a transpose invoke: true = remembered (nothing launched)
bool dt_defer(const UnaryDesc *d, void *pi, void *po, hipStream_t s) {
  if (d->dtype != DT_F32 || d->m > 64 || d->n > 64 || d->ldo != d->m || !cfg().fold_transpose.load(std::memory_order_relaxed) || cfg().strict.load(std::memory_order_relaxed) || !queue_active()) return false;
  DeviceRanges &devmem = caller_state().devmem;
  if (devmem.refresh()) check_queue_device();
  if (!devmem.is_device(pi, 0) || !devmem.is_device(po, 1)) return false;
  const size_t dst_bytes = (size_t)d->n * d->m * 4, src_bytes = span(d->m, d->ldi, d->n) * 4;
  if (dt_overlap(pi, src_bytes, po, dst_bytes)) return false;
  DtSlot *mine = dt_my_slot(true);
  if (!mine) return false;
  const uintptr_t s_lo = (uintptr_t)pi, s_hi = (uintptr_t)pi + src_bytes;
  bool replaced = false, launch_old = false;
  if (mine->live.load(std::memory_order_acquire)) {
    std::lock_guard<SpinLock> lk(mine->mu);
    if (mine->live.load(std::memory_order_relaxed)) {
      DeferredTranspose &r = mine->r;
      if (r.d == d && r.dst == po && r.stream == s) {
        r.src = pi; // the remembered transpose is dead: fully overwritten, its readers were served from its source
        // (the published source range only GROWS while the record lives: the hull of the sources of the loop's transposes - after one
        // pass over the source tensor the line the other threads read is not written any more)
        if (s_lo < mine->s_lo.load(std::memory_order_relaxed)) mine->s_lo.store(s_lo, std::memory_order_relaxed);
        if (s_hi > mine->s_hi.load(std::memory_order_relaxed)) mine->s_hi.store(s_hi, std::memory_order_relaxed);
        mine->dropped.store(mine->dropped.load(std::memory_order_relaxed) + 1, std::memory_order_relaxed);
        replaced = true;
      } else {
        launch_old = true;
      }
    }
  }
  if (launch_old) dt_launch(*mine);
  // the other threads' records this transpose touches (it will read its source and write its destination when it is launched)
  if (g_dt_pending.load(std::memory_order_relaxed) > (replaced ? 1 : 0)) {
    const DtRange rd[1] = {dt_range(pi, src_bytes)}, wr[1] = {dt_range(po, dst_bytes)};
    dt_scan_foreign(mine, rd, 1, wr, 1);
  }
  if (replaced) return true;
  std::lock_guard<SpinLock> lk(mine->mu);
  if (mine->live.load(std::memory_order_relaxed)) return false; // (cannot happen: only the owner makes a record live)
  mine->r = DeferredTranspose{d, pi, po, s, nullptr, nullptr};
  mine->d_lo.store((uintptr_t)po, std::memory_order_relaxed);
  mine->d_hi.store((uintptr_t)po + dst_bytes, std::memory_order_relaxed);
  mine->s_lo.store(s_lo, std::memory_order_relaxed);
  mine->s_hi.store(s_hi, std::memory_order_relaxed);
  g_dt_pending.fetch_add(1, std::memory_order_relaxed);
  mine->live.store(1, std::memory_order_release);
  return true;
}

// ---- strict mode: single invokes of queue-sized tiles run on the grouped launcher with a work list of ONE item. The kernels read
// their item from device-visible memory: a per-thread ring of pinned (device-mapped) items, like the tile queue's lists; the stream
// is drained once per lap of the ring, so a slot is never rewritten while a launch may still read it.
struct StrictRing {
  static constexpr int N = 1024;
  WorkItem *items = nullptr;
  int next = 0;
  ~StrictRing() {
    if (items) (void)hipHostFree(items);
  }
};
WorkItem *strict_item_slot(hipStream_t s) {
  thread_local StrictRing r;
  if (!r.items) HIP_OK(hipHostMalloc((void **)&r.items, sizeof(WorkItem) * StrictRing::N, hipHostMallocDefault));
  if (r.next == StrictRing::N) {
    HIP_OK(hipStreamSynchronize(s));
    r.next = 0;
  }
  return &r.items[r.next++];
}
void strict_item_done(hipStream_t) {}

// ---- host cache (round 6, host_cache.h): host operands translated to device mirrors that outlive the invoke ---------------------
// One scope per ABI invoke: the constructor translates the host operands (their pointers are REPLACED by mirror addresses, so the
// tile queue, the deferred transposes and the launch paths below see device memory), the destructor - behind the launch and, in
// synchronous mode, behind finish()'s stream synchronisation - copies what was written back (synchronous mode) or remembers it for
// the next synchronisation point (asynchronous mode) and ends the reader section.
void hc_flush_hook() { flush_tile_queue(); }
bool hc_is_device_hook(const void *p, int pos) {
  DeviceRanges &dm = caller_state().devmem;
  if (cfg().async.load(std::memory_order_relaxed)) dm.refresh();
  else dm.known.clear(); // synchronous mode: every invoke is a point after which the caller may free buffers (see stage_in)
  return dm.is_device(p, pos);
}
bool hc_setup() {
  hc::set_hooks(hc::Hooks{&hc_flush_hook, &hc_is_device_hook});
  if (const char *e = getenv("TPP_HIP_HOST_CACHE"))
    if (atoi(e) != 0) (void)hc::set_enabled(1);
  return true;
}
inline bool hc_on() {
  static const bool once = hc_setup();
  (void)once;
  return hc::enabled();
}
struct HcScope {
  hc::OpRef ops[4];
  int n = 0, hits = 0;
  bool async = false;
  hipStream_t s = nullptr;
  void *memo = nullptr; // asynchronous mode: the invoke was answered from the thread's whole-invoke memo (hc::memo_hit)
  void add(void **pp, const Operand &o, bool read, bool written) {
    ops[n++] = hc::OpRef{pp, o.bytes, o.rows, o.row_bytes, o.pitch, read, written, nullptr, 0};
  }
  uint64_t epoch = 0;
  void go(hipStream_t stream) {
    s = stream;
    async = cfg().async.load(std::memory_order_relaxed) != 0;
    epoch = g_devmem_epoch.load(std::memory_order_relaxed);
    hits = hc::translate(ops, n, async, epoch, s);
  }
  ~HcScope() {
    if (memo) {
      hc::memo_done(memo, epoch);
      return;
    }
    if (!hits) return;
    hc::complete(ops, n, async, epoch, s);
    hc::leave();
  }
};

void gemm_invoke_common(const char *who, bool want_fused, int64_t dtype, int64_t handle, void *a, int64_t off_a,
                        void *b, int64_t off_b, void *c, int64_t off_c, void *dptr, int64_t off_d, int64_t br) {
  const GemmDesc *d = as_desc<GemmDesc>(handle, KIND_GEMM, who);
  if (d->dtype != dtype) die("%s: invoke dtype %ld != dispatch dtype %ld", who, (long)dtype, (long)d->dtype);
  if (want_fused != (d->fused != 0)) die("%s: handle dispatched for a different gemm flavour", who);
  if (br < 0) die("%s: negative batch count %ld", who, (long)br);
  if (d->m == 0 || d->n == 0) return;
  TraceRange trace_range(who, d->trace);
  const size_t es = esize(dtype);
  void *pa = (char *)a + off_a * es, *pb = (char *)b + off_b * es, *pc = (char *)c + off_c * es;
  void *pd = dptr ? (char *)dptr + off_d * es : nullptr;
  if (d->bias && !dptr) die("%s: fused bias operand is null", who);
  hipStream_t s = invoke_stream();
  HcScope hcs;
  if (hc_on()) {
    const bool async = cfg().async.load(std::memory_order_relaxed) != 0;
    hcs.epoch = g_devmem_epoch.load(std::memory_order_relaxed);
    if (async) hcs.memo = hc::memo_hit(d, &pa, &pb, &pc, &pd, br, hcs.epoch, s);
    if (!hcs.memo) {
      Operand A, B, C, D;
      gemm_operands(d, pa, pb, pc, pd, br, A, B, C, D);
      hcs.add(&pa, A, true, false);
      hcs.add(&pb, B, true, false);
      hcs.add(&pc, C, !d->beta0, true);
      hcs.add(&pd, D, true, false);
      hcs.go(s);
      if (async && hcs.hits) hc::memo_store(d, br, hcs.ops, 4, hcs.epoch, s);
    }
  }
  if (g_dt_pending.load(std::memory_order_acquire)) { // a remembered transpose: this gemm reads its source instead, or it is launched now
    void *src = nullptr;
    if (const GemmDesc *sib = dt_gemm(d, pa, pb, pc, pd, br, s, &src)) {
      d = sib;
      pb = src;
    }
  }
  if (cfg().tile_queue.load(std::memory_order_relaxed)) {
    if (cfg().async.load(std::memory_order_relaxed) && try_enqueue(d, pa, pb, pc, pd, br, s)) return;
    flush_tile_queue();
  }
  Operand A, B, C, D;
  gemm_operands(d, pa, pb, pc, pd, br, A, B, C, D);
  C.read = !d->beta0; // pure output under BETA_0: never uploaded
  std::vector<Operand *> ops = {&A, &B, &C, &D};
  stage_in(ops, s);
  if (cfg().strict.load(std::memory_order_relaxed) && d->m <= 64 && d->n <= 64) {
    // strict mode: a tile the queue would take runs on the kernel its group runs on - the grouped launcher with a work list of one
    // (launch_gemm_grouped decides as if every list held one item: xsmm_desc.h strict_kernels)
    const WorkItem one{A.dev, B.dev, C.dev, D.dev, br};
    WorkItem *slot = strict_item_slot(s);
    *slot = one;
    HIP_OK(launch_gemm_grouped(*d, slot, 1, ((((uintptr_t)A.dev) | ((uintptr_t)B.dev)) & 15) == 0,
                               (((uintptr_t)C.dev) & 15) == 0 && (((uintptr_t)D.dev) & 7) == 0 && br >= 1, !(br & 1), br, s));
    strict_item_done(s);
  } else {
    HIP_OK(launch_gemm(*d, A.dev, B.dev, C.dev, D.dev, br, s));
  }
  finish(ops, s);
}

// ---- chains of whole-layer fused BRGEMMs in one launch (xsmm_hip_fused_brgemm_chain_invoke) -----------------------------
// Hand-off state of the chain kernel (brgemm_bf16_lw.hip, chain mode): arrival counters that only grow - a launch adds
// tiles_n to each, its target is epoch * tiles_n - so a block of counters is tied to ONE (stream, tile grid, layer count):
// launches of one block are ordered by their stream and issued under the mutex (epoch order = stream order). The err word
// lives in pinned host memory: a workgroup whose wait timed out writes it over PCIe, the host reads it at its sync points.
struct ChainBlock {
  hipStream_t stream;
  int tiles_m, tiles_n, nlayers;
  unsigned *cnt; // device: (CH_MAXL - 1) * tiles_m counters, CHAIN_CNT_STRIDE words apart
  unsigned *err; // pinned host
  unsigned epoch;
  int verified;  // launches of this block that were checked synchronously and had every hand-off succeed (probation, see below)
};
// A STARVED chain launch (another process, or another stream's LDS-heavy kernel, held compute units while it ran: not every
// workgroup became resident, a consumer's bounded wait ran out, the error word is set) used to end the process. Round 5 (VERDICT r4
// item 5): the library degrades instead - the reference never aborts on a valid invoke (XsmmRunnerUtils.cpp:363-383).
//  * PROBATION: the first launch of every block (stream, tile grid, layer count) is followed by a stream synchronisation and a look
//    at the error word. A device that is shared when the harness starts is found out here, before anything could consume the
//    launch's outputs: the chain call then runs call by call at once (its inputs are intact: beta 0, outputs overlap no operand),
//    and the process remembers that the device is shared - every later chain invoke runs call by call (TPP_HIP_CHAIN=0 behaviour).
//  * LATER launches stay asynchronous; the calls of every launch since the last check are kept in a journal (the last launch per
//    set of output pointers). If the check at a synchronisation point finds the error word set, the journal is re-run call by call
//    in launch order before the synchronisation returns: what the caller then reads is what the calls compute from the operands as
//    they are now. (Work that OTHERS enqueued between a starved launch and the synchronisation has read invalid outputs - the
//    stderr line says so; the chain contract of include/tpp_xsmm_abi.h asks for the device to oneself for this reason.)
// Round 6: every journaled launch has its OWN error word (a pool of pinned words), so the check knows WHICH launch
// starved: only that launch and the later ones of the same stream are re-run (a healthy earlier launch whose inputs have since
// been overwritten is left alone); the journal is looked at per stream - the one that has just been drained - and the entries of
// other streams stay; when the pool runs dry the launching thread synchronises and checks instead of dropping entries; the re-run
// goes to the launch's stream through a thread-local override (the process-wide stream setting is not touched);
// xsmm_hip_chain_status() counts the repairs, TPP_HIP_CHAIN_STRICT=1 keeps fail-stop.
struct ChainCall {
  int n;
  int64_t dtype;
  int64_t handle[CH_MAXL];
  void *a[CH_MAXL], *b[CH_MAXL], *c[CH_MAXL], *d[CH_MAXL];
  int64_t br[CH_MAXL];
  hipStream_t stream;
  unsigned *err; // this launch's own error word (pinned host memory, from g_chain_err_free)
};
std::vector<ChainCall> g_chain_journal; // under g_chain_mu, in launch order
std::vector<unsigned *> g_chain_err_free; // under g_chain_mu
constexpr int CHAIN_ERR_POOL = 512;
std::atomic<int> g_chain_journaled{0};  // entries in the journal (read without the lock: "is the pool about to run dry")
std::atomic<int64_t> g_chain_repairs{0}; // starved launches found and re-run since process start (xsmm_hip_chain_status)
std::atomic<bool> g_chain_shared{false}; // a chain launch was starved once: no more chain launches in this process
std::mutex g_chain_mu;

std::vector<ChainBlock> g_chain_blocks;
std::atomic<int> g_chain_launched{0}; // chain launches since the last check of the err words

ChainBlock &chain_block(hipStream_t s, int tiles_m, int tiles_n, int nlayers) { // under g_chain_mu
  for (ChainBlock &b : g_chain_blocks)
    if (b.stream == s && b.tiles_m == tiles_m && b.tiles_n == tiles_n && b.nlayers == nlayers) return b;
  ChainBlock b{s, tiles_m, tiles_n, nlayers, nullptr, nullptr, 0, 0};
  const size_t bytes = sizeof(unsigned) * (size_t)(CH_MAXL - 1) * (size_t)tiles_m * CHAIN_CNT_STRIDE;
  HIP_OK(hipMalloc((void **)&b.cnt, bytes));
  HIP_OK(hipMemset(b.cnt, 0, bytes));
  HIP_OK(hipHostMalloc((void **)&b.err, sizeof(unsigned), hipHostMallocDefault));
  *b.err = 0;
  g_chain_blocks.push_back(b);
  return g_chain_blocks.back();
}
// after stream `s` has been drained: did a hand-off of a chain launch on it time out?
void dump_chain_stamps();
void chain_rerun_call_by_call(const ChainCall &c);
void check_chain_errors(hipStream_t s) {
  if (!g_chain_launched.load(std::memory_order_acquire)) return;
  std::vector<ChainCall> redo;
  unsigned layer = 0;
  {
    std::lock_guard<std::mutex> lk(g_chain_mu);
    dump_chain_stamps();
    std::vector<ChainCall> keep;
    for (const ChainCall &c : g_chain_journal) {
      if (c.stream != s) { // another stream's launch: not drained by this synchronisation, stays
        keep.push_back(c);
        continue;
      }
      const unsigned e = *(volatile unsigned *)c.err;
      if (e && !layer) layer = e;
      if (layer) redo.push_back(c); // the first starved launch of this stream and every later one (they may have consumed its outputs)
      *(volatile unsigned *)c.err = 0;
      g_chain_err_free.push_back(c.err);
    }
    g_chain_journal.swap(keep);
    g_chain_journaled.store((int)g_chain_journal.size(), std::memory_order_relaxed);
    if (g_chain_journal.empty()) g_chain_launched.store(0, std::memory_order_release);
  }
  if (!layer) return;
  static const bool strict = [] { const char *e = getenv("TPP_HIP_CHAIN_STRICT"); return e && atoi(e) != 0; }();
  if (strict)
    die("tpp-xsmm-hip: a fused-brgemm chain launch was starved (a hand-off for layer %u's input timed out: not every workgroup was resident - "
        "the device is shared) and TPP_HIP_CHAIN_STRICT=1 asks for fail-stop", layer - 1);
  // starved: the device is shared. The starved launch and the later ones of its stream run again, call by call, in launch order;
  // chains are off from now on.
  g_chain_shared.store(true, std::memory_order_release);
  g_chain_repairs.fetch_add((int64_t)redo.size(), std::memory_order_relaxed);
  fprintf(stderr, "[tpp-xsmm-hip] a fused-brgemm chain launch was starved (a hand-off for layer %u's input timed out: not every workgroup "
                  "was resident - the device is shared); that launch and the %zu later one(s) of its stream are re-run call by call now "
                  "(earlier launches completed and are left alone), and chain invokes run call by call from here on "
                  "(xsmm_hip_chain_status() counts; TPP_HIP_CHAIN_STRICT=1 ends the process instead). Work that others enqueued behind a "
                  "starved launch has read invalid data.\n",
          layer - 1, redo.size() - 1);
  for (const ChainCall &c : redo) chain_rerun_call_by_call(c);
  HIP_OK(hipStreamSynchronize(s));
}
void check_chain_errors() { check_chain_errors(cfg().stream.load()); }

int chip_cus() { // compute units of the current device (0: unknown)
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
// Compute units a launch on `s` can actually use: the device's, restricted by the stream's CU mask (hipExtStreamCreateWithCUMask,
// or the process-wide ROC_GLOBAL_CU_MASK / HSA_CU_MASK the runtime folds into every stream's mask). The persistent chain kernel
// needs all of its workgroups resident at once - one per CU - so its grid is checked against THIS number. What no query
// can see is another PROCESS (or another stream's LDS-heavy kernel) holding CUs at launch time: the chain launch needs the device
// to itself (include/tpp_xsmm_abi.h says so); every spin in the kernel is bounded and a starved launch is reported, not hung.
int stream_cus(hipStream_t s) {
  const int all = chip_cus();
  uint32_t mask[16] = {};
  if (all <= 0 || hipExtStreamGetCUMask(s, 16, mask) != hipSuccess) {
    (void)hipGetLastError();
    return all;
  }
  int bits = 0;
  for (uint32_t w : mask) bits += __builtin_popcount(w);
  return bits > 0 && bits < all ? bits : all;
}

// profiling (-DTPP_HIP_ABLATION side builds only, build.py --ablation): TPP_HIP_CHAIN_STAMPS=<file> makes every chain launch record s_memrealtime stamps (100 MHz) per workgroup and layer
// (see blw_stamp in brgemm_bf16_lw.hip) into pinned host memory; the LAST launch's stamps are written to the file at every sync point.
unsigned long long *g_stamps = nullptr;
size_t g_stamps_wgs = 0;
unsigned long long *chain_stamps(size_t wgs) { // under g_chain_mu
#ifdef TPP_HIP_ABLATION
  static const char *path = getenv("TPP_HIP_CHAIN_STAMPS");
  if (!path) return nullptr;
  if (!g_stamps) HIP_OK(hipHostMalloc((void **)&g_stamps, sizeof(unsigned long long) * 8 * CH_MAXL * 1024, hipHostMallocDefault));
  if (wgs > 1024) return nullptr;
  g_stamps_wgs = wgs;
  return g_stamps;
#else
  (void)wgs;
  return nullptr; // the shipped kernels carry no stamp code (brgemm_bf16_lw.hip: blw_stamp)
#endif
}
void dump_chain_stamps() {
#ifdef TPP_HIP_ABLATION
  const char *path = getenv("TPP_HIP_CHAIN_STAMPS");
#else
  const char *path = nullptr;
#endif
  if (!path || !g_stamps || !g_stamps_wgs) return;
  if (FILE *f = fopen(path, "w")) {
    for (size_t w = 0; w < g_stamps_wgs; ++w)
      for (int l = 0; l < CH_MAXL; ++l) {
        const unsigned long long *s = g_stamps + (w * CH_MAXL + l) * 8;
        if (!s[0] && !s[5]) continue;
        fprintf(f, "%zu %d", w, l);
        for (int i = 0; i < 8; ++i) fprintf(f, " %llu", s[i]);
        fputc('\n', f);
      }
    fclose(f);
  }
  // the loaders' per-chunk records of the first 16 workgroups (TPP_HIP_CHAIN_DBG & 1024; brgemm_bf16_lw.hip BlwChunkStamps)
  if (chain_ablation_bits() & 1024) {
    const std::string p2 = std::string(path) + ".chunks";
    if (FILE *f = fopen(p2.c_str(), "w")) {
      const unsigned long long *base = g_stamps + g_stamps_wgs * CH_MAXL * 8;
      for (int w = 0; w < 16 && (size_t)w < g_stamps_wgs; ++w)
        for (int which = 0; which < 2; ++which) {
          const unsigned long long *r = base + ((size_t)w * 2 + which) * (64 * 3 + 1);
          const int n = (int)(r[0] > 64 ? 64 : r[0]);
          for (int i = 0; i < n; ++i) fprintf(f, "%d %d %d %llu %llu %llu\n", w, which, i, r[1 + 3 * i], r[2 + 3 * i], r[3 + 3 * i]);
        }
      fclose(f);
    }
  }
}

bool ranges_overlap(const void *a, size_t na, const void *b, size_t nb) {
  return (const char *)a < (const char *)b + nb && (const char *)b < (const char *)a + na;
}

// true: the chain was launched as ONE kernel. false: the caller runs the invokes one by one (same result).
bool try_chain_launch(int n, const GemmDesc *const *d, void *const *pa, void *const *pb, void *const *pc, void *const *pd, const int64_t *br,
                      hipStream_t s) {
  // with TPP_HIP_TRACE >= 1 the reason for running call by call goes to stderr
#define NOCHAIN(why)                                                                             \
  do {                                                                                           \
    if (cfg().trace) fprintf(stderr, "[tpp-xsmm-hip] fused_brgemm_chain: call by call (%s)\n", why); \
    return false;                                                                                \
  } while (0)
  if (n < 2 || n > CH_MAXL) NOCHAIN("fewer than 2 or more than 8 calls");
  if (!cfg().async.load(std::memory_order_relaxed)) NOCHAIN("synchronous mode");
  {
    // a launch's hand-off target (epoch x tiles per row block) is baked into its arguments: replayed from a graph it would be
    // stale - the consumers would not wait. Captured streams get the separate launches.
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(cfg().stream.load(std::memory_order_relaxed), &cs) != hipSuccess) (void)hipGetLastError();
    else if (cs != hipStreamCaptureStatusNone) NOCHAIN("the stream is being captured into a graph");
  }
  static const int enabled = [] {
    const char *e = getenv("TPP_HIP_CHAIN");
    return e ? atoi(e) : 1;
  }();
  if (!enabled) NOCHAIN("TPP_HIP_CHAIN=0");
  if (g_chain_shared.load(std::memory_order_acquire)) NOCHAIN("an earlier chain launch was starved: the device is shared");
  const int64_t m = d[0]->m, nn = d[0]->n;
  thread_local DeviceRanges devmem;
  devmem.refresh();
  // f32 chains (round 4): every call planned on the SAME K-split loader-wave tile (the launch is then bit-identical to the calls)
  const bool f32 = d[0]->dtype == DT_F32;
  const int f32_tile = f32 ? f32_chain_tile(*d[0]) : -1;
  for (int i = 0; i < n; ++i) {
    const GemmDesc &g = *d[i];
    if (f32) {
      if (g.dtype != DT_F32 || !g.beta0 || f32_chain_tile(g) < 0 || f32_chain_tile(g) != f32_tile)
        NOCHAIN("an f32 call is not beta 0 / not planned on the K-split loader-wave tile of the first call");
      if (g.bias && ((uintptr_t)pd[i] & 15)) NOCHAIN("an f32 bias operand is not 16-byte aligned");
      if (g.ldc & 3) NOCHAIN("an f32 output's leading dimension is not a multiple of 4");
    } else
    // every layer the same kind of B operand (VNNI-2, flat or VNNI-4: the B image is a template parameter of the launch)
    if (g.dtype != DT_BF16 || g.vnni_c || !g.beta0 || bf16_lw_b_kind(g) < 0 || bf16_lw_b_kind(g) != bf16_lw_b_kind(*d[0]))
      NOCHAIN("a call is not bf16 / beta 0 / aligned for the LDS-DMA tiles, or the calls' B operands differ in kind (VNNI-2 / flat / VNNI-4)");
    if (g.m != m || g.n != nn || br[i] < 1) NOCHAIN("the calls differ in m or n, or a batch is empty");
    if (g.variant == GEMM_VARIANT_GENERIC) NOCHAIN("a call was dispatched to the generic kernel"); // (a forced generic kernel stays generic)
    if (((uintptr_t)pa[i] | (uintptr_t)pb[i] | (uintptr_t)pc[i]) & 15) NOCHAIN("an operand is not 16-byte aligned");
    if (g.bias && (!pd[i] || ((uintptr_t)pd[i] & 7))) NOCHAIN("a bias operand is not 8-byte aligned");
    if (i > 0 && (pa[i] != pc[i - 1] || g.lda != d[i - 1]->ldc)) NOCHAIN("not a chain: a call does not read its predecessor's output");
    // The kernel hands layer i-1's output over row block by row block (a consumer waits for the producers of ITS rows only): every
    // batch element of layer i must stay inside its own rows, i.e. the batch strides walk along k within one leading dimension.
    // (A row-striding stride_a would read rows that other workgroups may not have stored yet.)
    if (i > 0 && (br[i] - 1) * g.stride_a + g.k > g.lda) NOCHAIN("a later call's batch elements leave the rows of its predecessor's output");
    if (!devmem.is_device(pa[i], 0) || !devmem.is_device(pb[i], 1) || !devmem.is_device(pc[i], 2) || (g.bias && !devmem.is_device(pd[i], 3)))
      NOCHAIN("a host operand");
  }
  // The tile: all workgroups must be co-resident (one per CU by LDS), so the grid may not exceed the CUs. If every layer was planned
  // with the same loader-wave tile and that tile fits, use it - the launch is then bit-identical to the separate launches; else
  // the smallest tile that fits (most CUs busy).
  int tile = -1, bm = 0, bn = 0;
  const int64_t cus = stream_cus(s); // (the stream the launch goes to: ADVICE r4)
  auto fits = [&](int t) {
    if (f32) (void)f32_chain_tile_dims(t, &bm, &bn);
    else blw_tile_dims(t, &bm, &bn);
    return m % bm == 0 && nn % bn == 0 && (m / bm) * (nn / bn) <= cus;
  };
  if (f32 && !fits(f32_tile)) NOCHAIN("more tiles than compute units");
  const int b_kind = f32 ? 0 : bf16_lw_b_kind(*d[0]);
  // (variants 20 .. 23 VNNI-2, 24 .. 27 flat B, 28 .. 31 VNNI-4: the same four tiles)
  const int planned = d[0]->variant - (b_kind == 2 ? GEMM_VARIANT_BF16_LW0 + 4 : b_kind == 4 ? GEMM_VARIANT_BF16_LW4_0 : GEMM_VARIANT_BF16_LW0);
  bool same = !f32 && planned >= 0 && planned < 4;
  for (int i = 1; i < n && same; ++i) same = d[i]->variant == d[0]->variant;
  if (f32) tile = f32_tile;
  if (same && fits(planned)) tile = planned;
  if (tile < 0 && cfg().strict.load(std::memory_order_relaxed)) NOCHAIN("strict mode: one launch only on the tile the layers were planned on");
  for (int t = 0; t < 4 && tile < 0; ++t)
    if (fits(t)) tile = t;
  if (tile < 0) NOCHAIN("more tiles than compute units");
  (void)fits(tile); // bm, bn of the chosen tile
  // no operand of the launch may overlap an output (a layer's input rows are read by other workgroups while later layers store)
  Operand A, B, C, D;
  struct Span { const void *p; size_t n; };
  Span outs[CH_MAXL], ins[2 * CH_MAXL + 1], a_in[CH_MAXL];
  int n_ins = 0;
  for (int i = 0; i < n; ++i) {
    gemm_operands(d[i], pa[i], pb[i], pc[i], pd[i], br[i], A, B, C, D);
    outs[i] = Span{C.ptr, C.bytes};
    a_in[i] = Span{A.ptr, A.bytes};
    ins[n_ins++] = Span{B.ptr, B.bytes};
    if (d[i]->bias) ins[n_ins++] = Span{D.ptr, D.bytes};
    if (i == 0) ins[n_ins++] = Span{A.ptr, A.bytes};
  }
  for (int i = 0; i < n; ++i) {
    for (int j = i + 1; j < n; ++j)
      if (ranges_overlap(outs[i].p, outs[i].n, outs[j].p, outs[j].n)) NOCHAIN("two outputs overlap");
    for (int j = 0; j < n_ins; ++j)
      if (ranges_overlap(outs[i].p, outs[i].n, ins[j].p, ins[j].n)) NOCHAIN("an output overlaps an input");
    // the A operand of a later layer is its predecessor's output by construction; what it reads (k may be wider than the
    // predecessor's n: the gap columns of the rows) may overlap no OTHER output of the launch
    for (int j = 1; j < n; ++j)
      if (j != i + 1 && ranges_overlap(outs[i].p, outs[i].n, a_in[j].p, a_in[j].n)) NOCHAIN("an output overlaps a later call's input");
  }
#undef NOCHAIN
  ChainArgs c;
  memset(&c, 0, sizeof(c));
  c.A = pa[0];
  c.lda = d[0]->lda;
  c.m = (int)m;
  c.n = (int)nn;
  c.nlayers = n;
  c.dbg = chain_ablation_bits();
  for (int i = 0; i < n; ++i)
    c.L[i] = ChainLayer{pb[i], pd[i], pc[i], d[i]->ldb, d[i]->ldc, d[i]->stride_a, d[i]->stride_b, (int)d[i]->k, (int)br[i],
                        EP_BETA0 | (d[i]->bias ? EP_BIAS : 0) | (d[i]->relu ? EP_RELU : 0), 0};
  // the pool of error words is about to run dry (hundreds of launches without a synchronisation): synchronise and check here
  // instead of ever dropping a journal entry
  if (g_chain_journaled.load(std::memory_order_relaxed) >= CHAIN_ERR_POOL - 8) {
    std::vector<hipStream_t> streams;
    {
      std::lock_guard<std::mutex> lk0(g_chain_mu);
      for (const ChainCall &j : g_chain_journal)
        if (std::find(streams.begin(), streams.end(), j.stream) == streams.end()) streams.push_back(j.stream);
    }
    for (hipStream_t st : streams) {
      HIP_OK(hipStreamSynchronize(st));
      check_chain_errors(st);
    }
    if (g_chain_shared.load(std::memory_order_acquire)) return false; // (found a starved launch: call by call from here on)
  }
  std::lock_guard<std::mutex> lk(g_chain_mu);
  ChainBlock &blk = chain_block(s, (int)(m / bm), (int)(nn / bn), n);
  c.cnt = blk.cnt;
  c.err = blk.err; // probation launches: the block's word (checked right behind the launch)
  if (blk.verified >= 1) {
    if (g_chain_err_free.empty() && g_chain_journal.empty()) { // first use: the pool
      unsigned *pool = nullptr;
      HIP_OK(hipHostMalloc((void **)&pool, sizeof(unsigned) * CHAIN_ERR_POOL, hipHostMallocDefault));
      for (int i = 0; i < CHAIN_ERR_POOL; ++i) {
        pool[i] = 0;
        g_chain_err_free.push_back(pool + i);
      }
    }
    if (g_chain_err_free.empty()) return false; // (cannot happen: the check above keeps 8 words spare; call by call is always right)
    c.err = g_chain_err_free.back();
    g_chain_err_free.pop_back();
  }
  c.target = ++blk.epoch * (unsigned)blk.tiles_n;
  c.stamps = chain_stamps((size_t)blk.tiles_m * (size_t)blk.tiles_n);
  if (f32) HIP_OK(launch_f32_chain(tile, c, s));
  else HIP_OK(launch_bf16_chain(tile, b_kind, c, s));
  if (blk.verified < 1) {
    // probation (comment at ChainBlock): wait for this launch and look at its error word before anyone can consume its outputs
    HIP_OK(hipStreamSynchronize(s));
    const unsigned e = *(volatile unsigned *)blk.err;
    if (e) {
      *(volatile unsigned *)blk.err = 0;
      g_chain_shared.store(true, std::memory_order_release);
      fprintf(stderr, "[tpp-xsmm-hip] the first fused-brgemm chain launch on this stream was starved (a hand-off for layer %u's input timed "
                      "out: not every workgroup was resident - the device is shared): this call and every later chain invoke run call "
                      "by call.\n", e - 1);
      return false; // the caller runs the calls one by one (inputs intact: beta 0, outputs overlap no operand)
    }
    ++blk.verified;
    return true;
  }
  // journal: the calls of this launch with its own error word, for a re-run should the check at the next synchronisation of this
  // stream find it starved
  {
    ChainCall j;
    j.n = n;
    j.dtype = d[0]->dtype;
    j.stream = s;
    j.err = c.err;
    for (int i = 0; i < n; ++i) {
      j.handle[i] = reinterpret_cast<int64_t>(d[i]);
      j.a[i] = pa[i]; j.b[i] = pb[i]; j.c[i] = pc[i]; j.d[i] = pd[i]; j.br[i] = br[i];
    }
    g_chain_journal.push_back(j);
    g_chain_journaled.store((int)g_chain_journal.size(), std::memory_order_relaxed);
  }
  g_chain_launched.store(1, std::memory_order_release);
  return true;
}

// the calls of one journaled chain launch, one by one (operands are pointers with offsets applied: offsets 0)
void chain_rerun_call_by_call(const ChainCall &c) {
  // on the stream the launch went to - through this thread's override: the process-wide setting is not touched (another thread may
  // invoke, or call xsmm_hip_set_stream, meanwhile: ADVICE r5)
  tl_stream_override = c.stream;
  tl_has_stream_override = true;
  for (int i = 0; i < c.n; ++i)
    xsmm_fused_brgemm_invoke(c.dtype, c.handle[i], c.a[i], 0, c.b[i], 0, c.c[i], 0, c.d[i], 0, c.br[i]);
  flush_tile_queue();
  tl_has_stream_override = false;
}

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

static void unary_invoke_common(const char *who, int64_t dtype, int64_t handle, void *in, int64_t off_in,
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
