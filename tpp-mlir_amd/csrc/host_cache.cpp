// host_cache.cpp - see host_cache.h. Host side only (no kernels): userfaultfd (async write-protect) + PAGEMAP_SCAN + device mirrors.
#include "host_cache.h"

#include <fcntl.h>
#include <linux/userfaultfd.h>
#include <sys/ioctl.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_set>
#include <vector>

// feature bits / ioctl of kernels newer than the container's headers (linux/userfaultfd.h, linux/fs.h of 6.7+), restated
#ifndef UFFD_FEATURE_WP_UNPOPULATED
#define UFFD_FEATURE_WP_UNPOPULATED (1 << 13)
#endif
#ifndef UFFD_FEATURE_WP_ASYNC
#define UFFD_FEATURE_WP_ASYNC (1 << 15)
#endif
#ifndef UFFD_USER_MODE_ONLY
#define UFFD_USER_MODE_ONLY 1
#endif

namespace tpp {
namespace hc {
std::atomic<int> g_on{0}; // (external linkage: host_cache.h enabled() reads it inline - once per invoke of every entry point)
namespace {

struct PmRegion {
  uint64_t start, end, categories;
};
struct PmScanArg {
  uint64_t size, flags, start, end, walk_end, vec, vec_len, max_pages, category_inverted, category_mask, category_anyof_mask, return_mask;
};
#define TPP_PAGEMAP_SCAN _IOWR('f', 16, struct PmScanArg)
enum : uint64_t { PG_WPALLOWED = 1, PG_WRITTEN = 2 };
enum : uint64_t { SCAN_WP_MATCHING = 1, SCAN_CHECK_WPASYNC = 2 };

constexpr uintptr_t PG = 4096;
inline uintptr_t pg_floor(uintptr_t x) { return x & ~(PG - 1); }
inline uintptr_t pg_ceil(uintptr_t x) { return (x + PG - 1) & ~(PG - 1); }

[[noreturn]] void die(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
  fflush(stderr);
  exit(-1); // XsmmRunnerUtils.cpp:132-137 convention, like runtime.cpp
}
#define HC_HIP_OK(expr)                                                                                        \
  do {                                                                                                         \
    hipError_t e_ = (expr);                                                                                    \
    if (e_ != hipSuccess) die("tpp-xsmm-hip (host cache): %s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

inline void cpu_relax() {
#if defined(__x86_64__)
  __builtin_ia32_pause();
#else
  asm volatile("" ::: "memory");
#endif
}

// One mirrored stretch of host address space. [lo, hi): the page-aligned hull of the operands seen here, registered with the
// userfaultfd; [cap_lo, cap_hi) >= [lo, hi): what the mirror allocation can hold (an extent grows upward inside its capacity without
// moving). valid bit of a page: the mirror page equals the host page as of the last scan (plus whatever kernels wrote since).
struct Extent {
  uintptr_t lo = 0, hi = 0, cap_lo = 0, cap_hi = 0;
  char *raw = nullptr, *mirror = nullptr; // mirror + (x - cap_lo) is the device address of host byte x
  std::vector<uint64_t> valid;
  std::vector<uint64_t> armed; // (under g_mu) the page is write-protected as far as the runtime knows: a page the kernel reports written
                               // although it is armed was written by the HOST; one that was never armed just reads as written
  std::atomic<uint64_t> polled_epoch{0};
  std::atomic<uint64_t> inval_gen{0}; // grows whenever a valid bit is cleared: what a thread's translation cache is checked against
  std::atomic<hipStream_t> stream{nullptr};
  char *dev(uintptr_t x) const { return mirror + (x - cap_lo); }
  bool get(uintptr_t page_addr) const {
    const size_t i = (page_addr - cap_lo) / PG;
    return (__atomic_load_n(&valid[i >> 6], __ATOMIC_RELAXED) >> (i & 63)) & 1;
  }
  void set(uintptr_t page_addr, bool v) {
    const size_t i = (page_addr - cap_lo) / PG;
    if (v) __atomic_fetch_or(&valid[i >> 6], uint64_t(1) << (i & 63), __ATOMIC_RELAXED);
    else {
      __atomic_fetch_and(&valid[i >> 6], ~(uint64_t(1) << (i & 63)), __ATOMIC_RELAXED);
      inval_gen.fetch_add(1, std::memory_order_release);
    }
  }
  bool is_armed(uintptr_t page_addr) const {
    const size_t i = (page_addr - cap_lo) / PG;
    return (armed[i >> 6] >> (i & 63)) & 1;
  }
  void arm(uintptr_t a, uintptr_t b, bool v) { // pages of [a, b)
    for (uintptr_t x = a; x < b; x += PG) {
      const size_t i = (x - cap_lo) / PG;
      if (v) armed[i >> 6] |= uint64_t(1) << (i & 63);
      else armed[i >> 6] &= ~(uint64_t(1) << (i & 63));
    }
  }
  bool all_valid(uintptr_t a, uintptr_t b) const { // pages of [a, b), a and b page-aligned
    size_t i = (a - cap_lo) / PG;
    const size_t e = (b - cap_lo) / PG;
    while (i < e) {
      const uint64_t w = __atomic_load_n(&valid[i >> 6], __ATOMIC_RELAXED);
      const size_t bit = i & 63, take = std::min<size_t>(64 - bit, e - i);
      const uint64_t mask = (take == 64 ? ~uint64_t(0) : ((uint64_t(1) << take) - 1)) << bit;
      if ((w & mask) != mask) return false;
      i += take;
    }
    return true;
  }
};

struct Pending {
  uintptr_t host;
  size_t bytes, rows, row_bytes, pitch;
};

struct SpinLock {
  std::atomic<bool> f{false};
  void lock() {
    while (f.exchange(true, std::memory_order_acquire))
      while (f.load(std::memory_order_relaxed)) cpu_relax();
  }
  void unlock() { f.store(false, std::memory_order_release); }
};

// what a thread remembers of its last translations (asynchronous mode: a timing loop passes the same tile pointers every iteration)
struct Cached {
  uintptr_t host = 0;
  size_t bytes = 0;
  char *dev = nullptr;
  Extent *e = nullptr;
  uint64_t gen = 0, epoch = 0, inval = 0, pend_epoch = 0; // structure generation, synchronisation epoch, e->inval_gen; epoch in which the footprint was noted as written
  hipStream_t stream = nullptr;
};
struct Memo { // one whole invoke (asynchronous mode)
  const void *desc = nullptr;
  void *p[4] = {nullptr, nullptr, nullptr, nullptr};
  int64_t br = 0;
  char *dev[4] = {nullptr, nullptr, nullptr, nullptr};
  Extent *e[4] = {nullptr, nullptr, nullptr, nullptr};
  uint64_t inval[4] = {0, 0, 0, 0};
  uint64_t gen = 0, epoch = 0, pend_epoch = 0;
  hipStream_t stream = nullptr;
  Pending wr{0, 0, 0, 0, 0}; // the written footprint (bytes == 0: none)
  bool last = false; // of its set's two ways, the one stored last
};
struct alignas(64) ThreadState {
  std::atomic<int> active{0}; // inside a reader section: holds mirror addresses, extents must not move
  ThreadState *next = nullptr;
  SpinLock pmu;
  std::vector<Pending> pending; // asynchronous mode: footprints written since the last synchronisation point
  std::unordered_set<uint64_t> seen;
  Extent *mru[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  uint64_t mru_gen = 0;
  bool holds_sync = false; // synchronous mode: this thread's invoke holds g_sync_mu (until leave())
  int depth = 0; // translate() .. leave() of this thread (an invoke issued from inside the flush hook - a remembered transpose - is not translated again)
  int64_t n_fast = 0, n_slow = 0; // (per thread: a shared counter would be the one line every caller writes)
  Cached cache[256];
  Cached *slot(uintptr_t p) { return &cache[(p * 0x9E3779B97F4A7C15ull) >> 56]; } // (tile pointers are a power of two apart: multiplicative hash)
  Memo memo[4096]; // 2048 sets of two ways (768 tile invokes per iteration of the reference's MLP: a direct-mapped table of this size
                   // still has a third of them share a slot with another one and miss every time)
  Memo *memo_set(const void *a, const void *c) { // (A, C) of a tile invoke: a layer's tiles differ in C, the layers in A
    const uint64_t h = (uint64_t)(uintptr_t)c * 0x9E3779B97F4A7C15ull ^ (uint64_t)(uintptr_t)a * 0xC2B2AE3D27D4EB4Full;
    return &memo[(h >> 53) * 2];
  }
};

std::mutex g_mu;                    // every slow path (polls, uploads, write-backs, structure changes)
// SYNCHRONOUS mode: ONE host-cached invoke at a time, from its translate() to the end of its write-back. Tiles of one row-major matrix
// share pages: thread 1's write-back marks a shared page for re-upload (it holds bytes thread 1 did not write) while thread 2's kernel
// has just put its own rows of that page into the mirror - thread 3's upload of the page would wipe them out before thread 2 copies them
// back. (The plain path gives every invoke a private mirror; asynchronous mode never invalidates a page between two synchronisation
// points.) Synchronous invokes serialise on the stream and its drain anyway.
std::mutex g_sync_mu;
std::atomic<int> g_writer{0};       // a structure change is waiting for / excluding the readers (set and cleared under g_mu)
std::atomic<ThreadState *> g_threads{nullptr};
std::vector<Extent *> g_ext;        // sorted by lo, disjoint; changed only with every other reader outside its section
std::atomic<uint64_t> g_struct_gen{1};
std::vector<Pending> g_orphans;     // pendings of threads that ended (under g_mu)
int g_uffd = -1, g_pm = -1;
Hooks g_hooks{nullptr, nullptr};
std::atomic<int64_t> st_upload_bytes{0}, st_scans{0}, st_wb_bytes{0}, st_wb_skipped{0}, st_grows{0}, st_drops{0};
struct Rejected { // host ranges that cannot be registered (file-backed, shared, huge-TLB mappings): the plain mirror path serves them
  uintptr_t lo, hi;
};
std::vector<Rejected> g_rejected;
const int g_trace = [] { const char *e = getenv("TPP_HIP_HOST_CACHE_TRACE"); return e ? atoi(e) : 0; }(); // 1: one stderr line per slow-path event
#define HC_TRACE(...) do { if (g_trace) { fprintf(stderr, "[tpp-xsmm-hip host cache] " __VA_ARGS__); fputc('\n', stderr); } } while (0)

// (one pointer in the static TLS block, like runtime.cpp's CallerState: a %fs-relative load instead of __tls_get_addr per invoke)
#ifdef TPP_TLS_DEFAULT_MODEL
static __thread ThreadState *tl_state = nullptr;
#else
static __thread ThreadState *tl_state __attribute__((tls_model("initial-exec"))) = nullptr;
#endif
struct ThreadHolder {
  ThreadState *t = nullptr;
  ~ThreadHolder() {
    if (!t) return;
    std::lock_guard<std::mutex> lk(g_mu);
    t->pmu.lock();
    for (const Pending &p : t->pending) g_orphans.push_back(p);
    t->pending.clear();
    t->seen.clear();
    t->pmu.unlock();
    t->active.store(0, std::memory_order_release); // the node stays in the list (it is small, and the writer walks a stable list)
    tl_state = nullptr;
  }
};
ThreadState &tstate_slow() {
  thread_local ThreadHolder h;
  if (!h.t) {
    h.t = new ThreadState();
    ThreadState *head = g_threads.load(std::memory_order_relaxed);
    do h.t->next = head;
    while (!g_threads.compare_exchange_weak(head, h.t, std::memory_order_release, std::memory_order_relaxed));
  }
  tl_state = h.t;
  return *h.t;
}
inline ThreadState &tstate() {
  ThreadState *t = tl_state;
  return t ? *t : tstate_slow();
}

// ---- kernel interface ---------------------------------------------------------------------------------------------------------
bool open_kernel_interface() {
  if (g_uffd >= 0 && g_pm >= 0) return true;
  if (sysconf(_SC_PAGESIZE) != (long)PG) return false;
  int fd = (int)syscall(SYS_userfaultfd, O_CLOEXEC | O_NONBLOCK | UFFD_USER_MODE_ONLY);
  if (fd < 0) return false;
  uffdio_api api;
  memset(&api, 0, sizeof api);
  api.api = UFFD_API;
  api.features = UFFD_FEATURE_WP_ASYNC | UFFD_FEATURE_WP_UNPOPULATED;
  if (ioctl(fd, UFFDIO_API, &api) != 0 || (api.features & (UFFD_FEATURE_WP_ASYNC | UFFD_FEATURE_WP_UNPOPULATED)) != (UFFD_FEATURE_WP_ASYNC | UFFD_FEATURE_WP_UNPOPULATED)) {
    close(fd);
    return false;
  }
  int pm = open("/proc/self/pagemap", O_RDONLY | O_CLOEXEC);
  if (pm < 0) {
    close(fd);
    return false;
  }
  // does this kernel know PAGEMAP_SCAN? (an empty scan of one page of our own stack)
  char probe_page;
  PmScanArg a;
  memset(&a, 0, sizeof a);
  PmRegion r;
  a.size = sizeof a;
  a.start = pg_floor((uintptr_t)&probe_page);
  a.end = a.start + PG;
  a.vec = (uint64_t)(uintptr_t)&r;
  a.vec_len = 1;
  a.return_mask = PG_WPALLOWED;
  if (ioctl(pm, TPP_PAGEMAP_SCAN, &a) < 0) {
    close(pm);
    close(fd);
    return false;
  }
  g_uffd = fd;
  g_pm = pm;
  return true;
}

int uffd_register(uintptr_t lo, uintptr_t hi) {
  uffdio_register r;
  memset(&r, 0, sizeof r);
  r.range.start = lo;
  r.range.len = hi - lo;
  r.mode = UFFDIO_REGISTER_MODE_WP;
  return ioctl(g_uffd, UFFDIO_REGISTER, &r) ? -errno : 0;
}
void uffd_unregister(uintptr_t lo, uintptr_t hi) {
  uffdio_range r;
  r.start = lo;
  r.len = hi - lo;
  (void)ioctl(g_uffd, UFFDIO_UNREGISTER, &r);
}

void uffd_unprotect(uintptr_t lo, uintptr_t hi) { // before the runtime itself writes the range (a write-back): no fault per page
  uffdio_writeprotect w;
  memset(&w, 0, sizeof w);
  w.range.start = lo;
  w.range.len = hi - lo;
  w.mode = 0;
  (void)ioctl(g_uffd, UFFDIO_WRITEPROTECT, &w);
}

// PAGEMAP_SCAN over [lo, hi). reprotect: the pages that match are write-protected again in the same call and the call fails (-EPERM)
// if any part of the range is not registered with OUR userfaultfd in async mode. Appends to `out`; returns 0 or -errno.
int scan(uintptr_t lo, uintptr_t hi, bool reprotect, uint64_t mask, uint64_t inverted, std::vector<PmRegion> &out) {
  PmRegion buf[64];
  st_scans.fetch_add(1, std::memory_order_relaxed);
  while (lo < hi) {
    PmScanArg a;
    memset(&a, 0, sizeof a);
    a.size = sizeof a;
    a.flags = reprotect ? (SCAN_WP_MATCHING | SCAN_CHECK_WPASYNC) : 0;
    a.start = lo;
    a.end = hi;
    a.vec = (uint64_t)(uintptr_t)buf;
    a.vec_len = 64;
    a.category_mask = mask;
    a.category_inverted = inverted;
    a.return_mask = PG_WPALLOWED | PG_WRITTEN;
    const long n = ioctl(g_pm, TPP_PAGEMAP_SCAN, &a);
    if (n < 0) return -errno;
    for (long i = 0; i < n; ++i) out.push_back(buf[i]);
    if (a.walk_end <= lo || n < 64) break; // the walk ended (vector not full): done
    lo = a.walk_end;
  }
  return 0;
}

// ---- readers / the one writer --------------------------------------------------------------------------------------------------
// A reader section spans one invoke: translate() .. leave(). Readers announce themselves with one sequentially consistent store and
// look at g_writer (Dekker); the writer - a structure change: an extent created, grown beyond its capacity, merged or dropped - holds
// g_mu, raises g_writer and waits until no OTHER thread is inside a section. A reader never waits for g_mu while it is announced.
inline void announce(ThreadState &t) {
  for (;;) {
    t.active.store(1, std::memory_order_seq_cst);
    if (!g_writer.load(std::memory_order_seq_cst)) return;
    t.active.store(0, std::memory_order_release);
    std::lock_guard<std::mutex> lk(g_mu); // the writer holds g_mu for as long as g_writer is up
  }
}
void exclude_readers(ThreadState &me) { // g_mu held, me.active == 0
  g_writer.store(1, std::memory_order_seq_cst);
  for (ThreadState *t = g_threads.load(std::memory_order_acquire); t; t = t->next)
    if (t != &me)
      while (t->active.load(std::memory_order_seq_cst)) cpu_relax();
}
void readmit_readers() { g_writer.store(0, std::memory_order_release); }

Extent *find(uintptr_t lo, uintptr_t hi) { // the extent whose observed range holds [lo, hi)
  for (Extent *e : g_ext) {
    if (lo >= e->lo && hi <= e->hi) return e;
    if (e->lo > lo) break;
  }
  return nullptr;
}

bool rejected(uintptr_t lo, uintptr_t hi) {
  for (const Rejected &r : g_rejected)
    if (lo < r.hi && r.lo < hi) return true;
  return false;
}

// structure change (writer): make one extent hold [lo, hi), merging every extent the range overlaps. Returns it, or nullptr if the
// range cannot be registered (then it is remembered as rejected).
Extent *grow(uintptr_t lo, uintptr_t hi, hipStream_t s) {
  std::vector<Extent *> over;
  uintptr_t nlo = lo, nhi = hi;
  for (Extent *e : g_ext)
    if (lo < e->hi && e->lo < hi) {
      over.push_back(e);
      nlo = std::min(nlo, e->lo);
      nhi = std::max(nhi, e->hi);
    }
  // (the new hull may reach further extents: repeat until it is stable)
  for (bool again = true; again;) {
    again = false;
    for (Extent *e : g_ext)
      if (nlo < e->hi && e->lo < nhi && std::find(over.begin(), over.end(), e) == over.end()) {
        over.push_back(e);
        nlo = std::min(nlo, e->lo);
        nhi = std::max(nhi, e->hi);
        again = true;
      }
  }
  if (over.size() == 1 && nlo == over[0]->lo && nhi <= over[0]->cap_hi) { // upward inside the capacity: nothing moves
    Extent *e = over[0];
    if (int err = uffd_register(e->hi, nhi)) {
      (void)err;
      g_rejected.push_back(Rejected{lo, hi});
      return nullptr;
    }
    e->hi = nhi;
    return e;
  }
  // register what is new first: a range that cannot be registered leaves everything as it was
  {
    std::vector<std::pair<uintptr_t, uintptr_t>> fresh{{nlo, nhi}};
    for (Extent *e : over) { // subtract the registered stretches
      std::vector<std::pair<uintptr_t, uintptr_t>> nx;
      for (auto &f : fresh) {
        if (e->hi <= f.first || e->lo >= f.second) { nx.push_back(f); continue; }
        if (f.first < e->lo) nx.push_back({f.first, e->lo});
        if (e->hi < f.second) nx.push_back({e->hi, f.second});
      }
      fresh.swap(nx);
    }
    for (size_t i = 0; i < fresh.size(); ++i)
      if (uffd_register(fresh[i].first, fresh[i].second)) {
        for (size_t j = 0; j < i; ++j) uffd_unregister(fresh[j].first, fresh[j].second);
        g_rejected.push_back(Rejected{lo, hi});
        return nullptr;
      }
  }
  Extent *n = new Extent();
  n->lo = nlo;
  n->hi = nhi;
  n->cap_lo = nlo;
  const uintptr_t len = nhi - nlo;
  n->cap_hi = nlo + std::max<uintptr_t>(over.empty() ? 2 * len : 4 * len, 256 * 1024);
  HC_HIP_OK(hipMalloc((void **)&n->raw, (n->cap_hi - n->cap_lo) + PG));
  n->mirror = (char *)pg_ceil((uintptr_t)n->raw);
  n->valid.assign(((n->cap_hi - n->cap_lo) / PG + 63) / 64, 0);
  n->armed.assign(n->valid.size(), 0);
  n->stream.store(s, std::memory_order_relaxed);
  uint64_t ep = over.empty() ? 0 : ~uint64_t(0);
  for (Extent *e : over) {
    if (e->stream.load(std::memory_order_relaxed) != s) HC_HIP_OK(hipStreamSynchronize(e->stream.load(std::memory_order_relaxed)));
    HC_HIP_OK(hipMemcpyAsync(n->dev(e->lo), e->dev(e->lo), e->hi - e->lo, hipMemcpyDeviceToDevice, s));
    for (uintptr_t x = e->lo; x < e->hi; x += PG) {
      if (e->get(x)) n->set(x, true);
      if (e->is_armed(x)) n->arm(x, x + PG, true);
    }
    ep = std::min(ep, e->polled_epoch.load(std::memory_order_relaxed));
  }
  if (!over.empty()) HC_HIP_OK(hipStreamSynchronize(s)); // the copies are done (and everything that used the old mirrors: the caller flushed the tile queue)
  // (pages that join here were never polled: they are invalid, and upload() protects before it reads - the parts' oldest poll epoch stands)
  n->polled_epoch.store(ep, std::memory_order_relaxed);
  for (Extent *e : over) {
    g_ext.erase(std::find(g_ext.begin(), g_ext.end(), e));
    HC_HIP_OK(hipFree(e->raw));
    delete e;
  }
  g_ext.insert(std::upper_bound(g_ext.begin(), g_ext.end(), n, [](const Extent *a, const Extent *b) { return a->lo < b->lo; }), n);
  g_struct_gen.fetch_add(1, std::memory_order_release);
  st_grows.fetch_add(1, std::memory_order_relaxed);
  HC_TRACE("extent [%#lx, %#lx) capacity %lu KiB (merged %zu)", (unsigned long)n->lo, (unsigned long)n->hi, (unsigned long)((n->cap_hi - n->cap_lo) >> 10), over.size());
  return n;
}

void drop(Extent *e, hipStream_t s) { // writer; the tile queue has been flushed
  HC_HIP_OK(hipStreamSynchronize(s));
  if (e->stream.load(std::memory_order_relaxed) != s) HC_HIP_OK(hipStreamSynchronize(e->stream.load(std::memory_order_relaxed)));
  uffd_unregister(e->lo, e->hi);
  g_ext.erase(std::find(g_ext.begin(), g_ext.end(), e));
  HC_HIP_OK(hipFree(e->raw));
  delete e;
  g_struct_gen.fetch_add(1, std::memory_order_release);
  st_drops.fetch_add(1, std::memory_order_relaxed);
}

// look at [lo, hi) of e (page-aligned, inside [e->lo, e->hi)): pages the host wrote lose their valid bit and are protected again.
// false: the range is no longer (entirely) ours - unmapped and mapped again, or never registered.
bool poll(Extent *e, uintptr_t lo, uintptr_t hi) {
  std::vector<PmRegion> w;
  int err = scan(lo, hi, true, PG_WRITTEN, 0, w);
  if (err == -EPERM) {
    // part of the range lost its registration (munmap + mmap behind our back, brk shrink + growth): whatever lives there now is
    // new - register it again (new pages then read as written) or give the extent up
    if (uffd_register(e->lo, e->hi) != 0) return false;
    w.clear();
    err = scan(lo, hi, true, PG_WRITTEN, 0, w);
  }
  if (err != 0) return false;
  e->arm(lo, hi, true); // (written pages were protected again by the scan, the others were protected before or are now)
  for (const PmRegion &r : w) {
    HC_TRACE("poll [%#lx, %#lx): written [%#lx, %#lx)", (unsigned long)lo, (unsigned long)hi, (unsigned long)r.start, (unsigned long)r.end);
    for (uintptr_t x = std::max<uintptr_t>(r.start, lo); x < std::min<uintptr_t>(r.end, hi); x += PG) e->set(x, false);
  }
  return true;
}

// Uploads go through the runtime's OWN pinned staging buffer, filled by a CPU copy. hipMemcpy from pageable memory makes the driver pin
// the caller's pages for writing (get_user_pages with write intent): every protected page takes a write-protect fault, reads as
// "written" at the next scan and would be uploaded again for ever (measured on the first GPU run of this file: 12 MiB per invoke, 93 ms
// each). A CPU read leaves the protection alone.
struct Staging {
  static constexpr size_t SLOT = 4u << 20;
  char *buf = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  bool used[2] = {false, false};
  int next = 0;
  char *acquire(int *slot) {
    if (!buf) {
      HC_HIP_OK(hipHostMalloc((void **)&buf, 2 * SLOT, hipHostMallocDefault));
      for (hipEvent_t &e : ev) HC_HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const int i = next;
    next ^= 1;
    if (used[i]) HC_HIP_OK(hipEventSynchronize(ev[i])); // the copy that last read this slot is done
    *slot = i;
    return buf + (size_t)i * SLOT;
  }
  void release(int slot, hipStream_t s) {
    HC_HIP_OK(hipEventRecord(ev[slot], s));
    used[slot] = true;
  }
};
Staging g_staging; // under g_mu
char *g_scratch = nullptr; // scratch block of synchronous pure outputs (under g_sync_mu + g_mu)
size_t g_scratch_cap = 0;

void upload(Extent *e, uintptr_t lo, uintptr_t hi, hipStream_t s, bool &flushed) {
  if (e->stream.load(std::memory_order_relaxed) != s) { // the mirror's last user was another stream: order behind it
    HC_HIP_OK(hipStreamSynchronize(e->stream.load(std::memory_order_relaxed)));
    e->stream.store(s, std::memory_order_relaxed);
  }
  for (uintptr_t x = lo; x < hi;) {
    if (e->get(x)) { x += PG; continue; }
    uintptr_t y = x + PG;
    while (y < hi && !e->get(y)) y += PG;
    if (!flushed) { // queued invokes were promised the mirror as it is NOW: they launch before it changes
      flushed = true;
      if (g_hooks.flush_queue) g_hooks.flush_queue();
    }
    // INVARIANT: a valid page is a write-protected page. Pages that joined the extent since its last poll (growth) are registered but
    // not protected yet - protect first, read second: a host write that lands after the protection is seen by the next poll, one that
    // landed before it is in the bytes uploaded now.
    {
      std::vector<PmRegion> ign;
      if (scan(x, y, true, PG_WRITTEN, 0, ign) == 0) e->arm(x, y, true);
    }
    for (uintptr_t c = x; c < y;) {
      const size_t len = std::min<size_t>(y - c, Staging::SLOT);
      int slot;
      char *st = g_staging.acquire(&slot);
      memcpy(st, (const void *)c, len);
      HC_HIP_OK(hipMemcpyAsync(e->dev(c), st, len, hipMemcpyHostToDevice, s));
      g_staging.release(slot, s);
      c += len;
    }
    HC_TRACE("upload [%#lx, %#lx) of extent [%#lx, %#lx)", (unsigned long)x, (unsigned long)y, (unsigned long)e->lo, (unsigned long)e->hi);
    st_upload_bytes.fetch_add((int64_t)(y - x), std::memory_order_relaxed);
    for (uintptr_t z = x; z < y; z += PG) e->set(z, true);
    x = y;
  }
}

// A PURE, DENSE output (written, not read, every byte of [p, p + bytes) written by the kernel - C under BETA_0 with ldc == n, the output
// of a unary with ldo == n): the pages that lie entirely inside it need no upload - the kernel overwrites them whole. They are marked
// valid as they are (the mirror page will be the truth once the kernel ran; whoever reads it later - the next layer - must not upload
// over it); they get their write protection when the footprint is written back (rearm). Only the two edge pages, which hold other
// bytes as well, are uploaded like any input.
inline bool pure_dense(const OpRef &o) { return o.written && !o.read && !(o.rows && o.row_bytes < o.pitch); }
void make_ready(Extent *e, const OpRef &o, uintptr_t p, hipStream_t s, bool &flushed) {
  const uintptr_t lo = pg_floor(p), hi = pg_ceil(p + o.bytes);
  if (pure_dense(o)) {
    const uintptr_t in_lo = pg_ceil(p), in_hi = pg_floor(p + o.bytes);
    if (in_lo < in_hi) {
      if (lo < in_lo) upload(e, lo, in_lo, s, flushed);
      for (uintptr_t x = in_lo; x < in_hi; x += PG)
        if (!e->get(x)) e->set(x, true);
      if (in_hi < hi) upload(e, in_hi, hi, s, flushed);
      return;
    }
  }
  upload(e, lo, hi, s, flushed);
}

// after a write-back of [a, b) (host bytes just overwritten with the mirror's): protect the pages again. Pages every byte of which
// was rewritten stay valid; a page only partly covered may also hold somebody else's write of the meantime: it is looked at.
void rearm(Extent *e, uintptr_t a, uintptr_t b, bool dense) {
  const uintptr_t lo = pg_floor(a), hi = pg_ceil(b);
  std::vector<PmRegion> w;
  if (scan(lo, hi, true, PG_WRITTEN, 0, w) != 0) { // lost its registration meanwhile: the next poll deals with it
    for (uintptr_t x = lo; x < hi; x += PG) e->set(x, false);
    e->arm(lo, hi, false);
    return;
  }
  e->arm(lo, hi, true);
  const uintptr_t full_lo = dense ? pg_ceil(a) : hi, full_hi = dense ? pg_floor(b) : hi; // pages entirely inside [a, b)
  HC_TRACE("written back [%#lx, %#lx)", (unsigned long)a, (unsigned long)b);
  for (const PmRegion &r : w)
    for (uintptr_t x = std::max<uintptr_t>(r.start, lo); x < std::min<uintptr_t>(r.end, hi); x += PG)
      if (!(x >= full_lo && x < full_hi)) e->set(x, false);
}

} // namespace

void set_hooks(const Hooks &h) { g_hooks = h; }

int translate(OpRef *ops, int n, bool async, uint64_t epoch, hipStream_t s) {
  for (int i = 0; i < n; ++i) ops[i].host = nullptr, ops[i].kind = 0;
  if (!g_on.load(std::memory_order_relaxed)) return 0;
  ThreadState &t = tstate();
  if (t.depth) return 0; // issued from inside the flush hook: its operands are mirror addresses already
  if (!async) {
    g_sync_mu.lock();
    t.holds_sync = true;
  }
  announce(t);
  // fast path (asynchronous mode): every host operand lies in an extent polled in this epoch, all its pages valid. The last answers
  // are remembered per thread: a timing loop passes the same pointers every iteration, and an answer stays good while no extent moved
  // (structure generation), no page of its extent lost its valid bit (inval_gen) and the epoch and stream are the same.
  const uint64_t gen = g_struct_gen.load(std::memory_order_acquire);
  if (t.mru_gen != gen) {
    for (Extent *&m : t.mru) m = nullptr;
    t.mru_gen = gen;
  }
  bool fast = true;
  int hits = 0;
  char *dev[8];
  for (int i = 0; i < n && fast; ++i) {
    dev[i] = nullptr;
    const uintptr_t p = (uintptr_t)*ops[i].ptr;
    if (!p || !ops[i].bytes) continue;
    Cached *c = t.slot(p);
    if (async && c->host == p && c->bytes == ops[i].bytes && c->gen == gen && c->epoch == epoch && c->stream == s &&
        c->inval == c->e->inval_gen.load(std::memory_order_acquire)) {
      dev[i] = c->dev;
      ++hits;
      continue;
    }
    Extent *e = t.mru[i & 7];
    if (!e || p < e->lo || p + ops[i].bytes > e->hi) {
      e = nullptr;
      for (Extent *x : g_ext)
        if (p >= x->lo && p + ops[i].bytes <= x->hi) { e = x; break; }
      if (e) t.mru[i & 7] = e;
    }
    if (!e) {
      if (g_hooks.is_device && g_hooks.is_device((const void *)p, i)) continue; // device memory: used in place
      fast = false;
      break;
    }
    const uint64_t inval = e->inval_gen.load(std::memory_order_acquire);
    if (!async || e->polled_epoch.load(std::memory_order_relaxed) != epoch || e->stream.load(std::memory_order_relaxed) != s ||
        !e->all_valid(pg_floor(p), pg_ceil(p + ops[i].bytes))) {
      fast = false;
      break;
    }
    dev[i] = e->dev(p);
    ++hits;
    const uint64_t pend = (c->host == p && c->bytes == ops[i].bytes) ? c->pend_epoch : 0;
    *c = Cached{p, ops[i].bytes, dev[i], e, gen, epoch, inval, pend, s};
  }
  if (fast) {
    if (!hits) {
      t.active.store(0, std::memory_order_release);
      if (t.holds_sync) {
        t.holds_sync = false;
        g_sync_mu.unlock();
      }
      return 0;
    }
    for (int i = 0; i < n; ++i)
      if (dev[i]) {
        ops[i].host = *ops[i].ptr;
        *ops[i].ptr = dev[i];
        ops[i].kind = 1;
      }
    ++t.n_fast;
    t.depth = 1;
    return hits;
  }
  // slow path: no mirror address is held while waiting for the lock
  t.active.store(0, std::memory_order_release);
  ++t.n_slow;
  t.depth = 1; // (the flush hook may issue invokes of its own on this thread)
  std::unique_lock<std::mutex> lk(g_mu);
  Extent *ext[8];
  bool scratch[8];
  bool need_writer = false;
  // SYNCHRONOUS mode, a pure dense output whose inner pages no extent holds: nothing of it is worth keeping on the device - the host gets
  // it at once anyway. It is computed into a scratch block and copied back by a DMA straight into the caller's pages, which is only
  // fast while those pages are NOT registered with the userfaultfd (the driver pins registered pages one by one: 60 us a page
  // measured); registering them for a mirror nobody will read would make every later copy-back crawl.
  auto scratch_output = [&](int i) {
    if (async || !pure_dense(ops[i])) return false;
    const uintptr_t p = (uintptr_t)*ops[i].ptr, in_lo = pg_ceil(p), in_hi = pg_floor(p + ops[i].bytes);
    if (in_lo >= in_hi) return false;
    for (Extent *c : g_ext)
      if (in_lo < c->hi && c->lo < in_hi) return false;
    return true;
  };
  for (int i = 0; i < n; ++i) {
    ext[i] = nullptr;
    dev[i] = nullptr;
    scratch[i] = false;
    const uintptr_t p = (uintptr_t)*ops[i].ptr;
    if (!p || !ops[i].bytes) continue;
    const uintptr_t lo = pg_floor(p), hi = pg_ceil(p + ops[i].bytes);
    ext[i] = find(lo, hi);
    if (ext[i]) continue;
    bool overlaps = false;
    for (Extent *c : g_ext) overlaps = overlaps || (lo < c->hi && c->lo < hi);
    if (!overlaps && g_hooks.is_device && g_hooks.is_device((const void *)p, i)) continue;
    if (rejected(lo, hi)) continue;
    if (scratch_output(i)) { scratch[i] = true; continue; }
    need_writer = true;
  }
  bool flushed = false;
  if (need_writer) {
    // every other invoke has left its section (whatever it translated is launched or sits in the tile queue) - THEN the queue is
    // flushed and the stream drained: nothing holds an address of a mirror that is about to move
    exclude_readers(t);
    if (g_hooks.flush_queue) g_hooks.flush_queue();
    flushed = true;
    HC_HIP_OK(hipStreamSynchronize(s));
    for (int i = 0; i < n; ++i) {
      const uintptr_t p = (uintptr_t)*ops[i].ptr;
      if (!p || !ops[i].bytes) continue;
      const uintptr_t lo = pg_floor(p), hi = pg_ceil(p + ops[i].bytes);
      // (an earlier operand of this invoke may have merged the extent this one was found in: look again)
      ext[i] = find(lo, hi);
      if (ext[i]) continue;
      bool overlaps = false;
      for (Extent *c : g_ext) overlaps = overlaps || (lo < c->hi && c->lo < hi);
      if (!overlaps && g_hooks.is_device && g_hooks.is_device((const void *)p, i)) continue;
      if (rejected(lo, hi)) continue;
      if (scratch[i]) continue;
      ext[i] = grow(lo, hi, s);
    }
    // (a later operand's grow may have merged - deleted - the extent an earlier one was found in: look all of them up again)
    for (int i = 0; i < n; ++i) {
      const uintptr_t p = (uintptr_t)*ops[i].ptr;
      ext[i] = (p && ops[i].bytes && !scratch[i]) ? find(pg_floor(p), pg_ceil(p + ops[i].bytes)) : nullptr;
    }
    readmit_readers();
  }
  // polls + uploads (no structure change: other readers go on)
  for (int pass = 0; pass < 2; ++pass) {
    bool lost = false;
    for (int i = 0; i < n; ++i) {
      Extent *e = ext[i];
      if (!e) continue;
      const uintptr_t p = (uintptr_t)*ops[i].ptr;
      const uintptr_t lo = pg_floor(p), hi = pg_ceil(p + ops[i].bytes);
      bool ok = true;
      if (!async) ok = poll(e, lo, hi); // synchronous mode: the host may have written between any two invokes
      else if (e->polled_epoch.load(std::memory_order_relaxed) != epoch) {
        ok = poll(e, e->lo, e->hi); // asynchronous mode: once per synchronisation epoch, the whole extent
        if (ok) e->polled_epoch.store(epoch, std::memory_order_relaxed);
      }
      if (!ok) { // the range is not ours any more: give the extent up (writer), this operand takes the plain mirror path
        exclude_readers(t);
        if (g_hooks.flush_queue) g_hooks.flush_queue();
        flushed = true;
        for (int j = 0; j < n; ++j)
          if (ext[j] == e && j != i) ext[j] = nullptr;
        drop(e, s);
        ext[i] = nullptr;
        readmit_readers();
        lost = true;
        continue;
      }
      make_ready(e, ops[i], p, s, flushed);
    }
    if (!lost) break;
  }
  t.active.store(1, std::memory_order_seq_cst); // (under g_mu: no writer can be waiting)
  hits = 0;
  for (int i = 0; i < n; ++i)
    if (ext[i]) {
      ops[i].host = *ops[i].ptr;
      *ops[i].ptr = ext[i]->dev((uintptr_t)ops[i].host);
      ops[i].kind = 1;
      t.mru[i & 7] = ext[i];
      ++hits;
    } else if (scratch[i]) { // (one written operand per invoke, one synchronous host-cached invoke at a time: g_sync_mu)
      const uintptr_t p = (uintptr_t)*ops[i].ptr;
      if (g_scratch_cap < ops[i].bytes + 512) {
        if (g_scratch) {
          HC_HIP_OK(hipStreamSynchronize(s));
          HC_HIP_OK(hipFree(g_scratch));
        }
        g_scratch_cap = std::max(ops[i].bytes + 512, 2 * g_scratch_cap);
        HC_HIP_OK(hipMalloc((void **)&g_scratch, g_scratch_cap));
      }
      ops[i].host = (void *)p;
      *ops[i].ptr = (char *)(((uintptr_t)g_scratch + 255) & ~(uintptr_t)255) + (p & 255); // (the caller's alignment class: kernel choices depend on it)
      ops[i].kind = 2;
      ++hits;
    }
  t.mru_gen = g_struct_gen.load(std::memory_order_relaxed);
  lk.unlock();
  if (!hits) {
    t.active.store(0, std::memory_order_release);
    t.depth = 0;
    if (t.holds_sync) {
      t.holds_sync = false;
      g_sync_mu.unlock();
    }
  }
  return hits;
}

void leave() {
  ThreadState &t = tstate();
  t.active.store(0, std::memory_order_release);
  t.depth = 0;
  if (t.holds_sync) {
    t.holds_sync = false;
    g_sync_mu.unlock();
  }
}

namespace {
// Write-backs go through the pinned staging buffer too, the host bytes are written by a CPU copy: a hipMemcpy INTO pageable memory of a
// userfaultfd-registered mapping makes the driver pin those pages one by one through the slow get_user_pages path (measured: 60 ms for
// 4 MiB, against 80 us into an unregistered buffer). The caller has unprotected the pages (no write-protect fault per page) and has
// drained the stream behind the kernels; this function returns with the host bytes in place.
void copy_back(Extent *e, uintptr_t host, size_t bytes, size_t rows, size_t row_bytes, size_t pitch, hipStream_t s) {
  const bool strided = rows && row_bytes < pitch;
  if (strided && row_bytes * 4 < pitch) { // narrow rows (a tile of a wide matrix): gather the rows on the device side of the copy
    for (size_t r0 = 0; r0 < rows;) {
      const size_t nr = std::min(rows - r0, Staging::SLOT / row_bytes);
      int slot;
      char *st = g_staging.acquire(&slot);
      HC_HIP_OK(hipMemcpy2DAsync(st, row_bytes, e->dev(host + r0 * pitch), pitch, row_bytes, nr, hipMemcpyDeviceToHost, s));
      HC_HIP_OK(hipStreamSynchronize(s));
      for (size_t r = 0; r < nr; ++r) memcpy((void *)(host + (r0 + r) * pitch), st + r * row_bytes, row_bytes);
      g_staging.release(slot, s);
      r0 += nr;
    }
  } else { // dense, or rows that fill most of their pitch: whole spans through the staging buffer, only the written bytes to the host
    for (size_t off = 0; off < bytes;) {
      size_t len = std::min(bytes - off, Staging::SLOT);
      if (strided && len < bytes - off) len = std::max<size_t>(len / pitch, 1) * pitch; // (whole rows per piece)
      len = std::min(len, bytes - off);
      int slot;
      const auto t0 = std::chrono::steady_clock::now();
      char *st = g_staging.acquire(&slot);
      const auto t1 = std::chrono::steady_clock::now();
      HC_HIP_OK(hipMemcpyAsync(st, e->dev(host + off), len, hipMemcpyDeviceToHost, s));
      HC_HIP_OK(hipStreamSynchronize(s));
      const auto t2 = std::chrono::steady_clock::now();
      if (!strided) {
        memcpy((void *)(host + off), st, len);
      } else {
        for (size_t r = 0; r * pitch < len; ++r) memcpy((void *)(host + off + r * pitch), st + r * pitch, std::min(row_bytes, len - r * pitch));
      }
      if (g_trace >= 2)
        fprintf(stderr, "[tpp-xsmm-hip host cache] copy_back %zu B: staging slot %.0f us, D2H %.0f us, memcpy %.0f us\n", len,
                std::chrono::duration<double, std::micro>(t1 - t0).count(), std::chrono::duration<double, std::micro>(t2 - t1).count(),
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t2).count());
      g_staging.release(slot, s);
      off += len;
    }
  }
  e->arm(pg_floor(host), pg_ceil(host + bytes), false);
  st_wb_bytes.fetch_add((int64_t)(strided ? rows * row_bytes : bytes), std::memory_order_relaxed);
}
} // namespace

void complete(OpRef *ops, int n, bool async, uint64_t epoch, hipStream_t s) {
  bool any = false;
  for (int i = 0; i < n; ++i) any = any || (ops[i].host && ops[i].written);
  if (!any) return;
  if (async) {
    ThreadState &t = tstate();
    for (int i = 0; i < n; ++i) {
      if (!ops[i].host || !ops[i].written) continue;
      Cached *c = t.slot((uintptr_t)ops[i].host);
      if (c->host == (uintptr_t)ops[i].host && c->bytes == ops[i].bytes) {
        if (c->pend_epoch == epoch) continue; // this footprint is on the list of this epoch already
        c->pend_epoch = epoch;
      }
      const uint64_t key = (uint64_t)(uintptr_t)ops[i].host * 0x9E3779B97F4A7C15ull ^ (uint64_t)ops[i].bytes * 0xC2B2AE3D27D4EB4Full ^ (uint64_t)ops[i].rows;
      std::lock_guard<SpinLock> lk(t.pmu);
      if (t.seen.insert(key).second) t.pending.push_back(Pending{(uintptr_t)ops[i].host, ops[i].bytes, ops[i].rows, ops[i].row_bytes, ops[i].pitch});
    }
    return;
  }
  // synchronous mode: the caller has drained the stream behind its kernel, so nothing in flight holds a mirror address of this
  // invoke any more: the section ends HERE (a reader never waits for g_mu while it is announced - a writer holding g_mu would wait
  // for it forever); the extents are looked up again by host address under the lock (a grow meanwhile has carried the data along).
  tstate().active.store(0, std::memory_order_release);
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < n; ++i) {
    if (!ops[i].host || !ops[i].written) continue;
    const uintptr_t h = (uintptr_t)ops[i].host;
    if (ops[i].kind == 2) {
      // scratch output: the inner pages by DMA straight into the caller's (unregistered) pages; an edge page that some extent holds
      // through the staging buffer + a CPU copy (the kernel's write tracking sees that write like any other host write)
      const char *src = (const char *)*ops[i].ptr;
      const uintptr_t end = h + ops[i].bytes, in_lo = pg_ceil(h), in_hi = pg_floor(end);
      auto piece = [&](uintptr_t a, uintptr_t b) {
        if (a >= b) return;
        bool registered = false;
        for (Extent *c : g_ext) registered = registered || (pg_floor(a) < c->hi && c->lo < pg_ceil(b));
        if (!registered) {
          HC_HIP_OK(hipMemcpyAsync((void *)a, src + (a - h), b - a, hipMemcpyDeviceToHost, s));
        } else {
          int slot;
          char *st = g_staging.acquire(&slot);
          HC_HIP_OK(hipMemcpyAsync(st, src + (a - h), b - a, hipMemcpyDeviceToHost, s));
          HC_HIP_OK(hipStreamSynchronize(s));
          memcpy((void *)a, st, b - a);
          g_staging.release(slot, s);
        }
        st_wb_bytes.fetch_add((int64_t)(b - a), std::memory_order_relaxed);
      };
      piece(h, in_lo);
      piece(in_lo, in_hi);
      piece(in_hi, end);
      continue;
    }
    Extent *e = find(pg_floor(h), pg_ceil(h + ops[i].bytes));
    if (!e) continue; // (given up meanwhile: its range changed hands)
    uffd_unprotect(pg_floor(h), pg_ceil(h + ops[i].bytes)); // the copy below writes these pages: no write-protect fault per page
    copy_back(e, h, ops[i].bytes, ops[i].rows, ops[i].row_bytes, ops[i].pitch, s);
  }
  HC_HIP_OK(hipStreamSynchronize(s));
  for (int i = 0; i < n; ++i) {
    if (!ops[i].host || !ops[i].written || ops[i].kind == 2) continue;
    const uintptr_t h = (uintptr_t)ops[i].host;
    Extent *e = find(pg_floor(h), pg_ceil(h + ops[i].bytes));
    if (!e) continue;
    if (pure_dense(ops[i])) {
      // a pure output: its pages stay unprotected and are not trusted - the next kernel that writes them whole needs nothing of them,
      // one that reads them (or their edge pages' other bytes) uploads them again. No scan, no protection round per invoke.
      for (uintptr_t x = pg_floor(h); x < pg_ceil(h + ops[i].bytes); x += PG) e->set(x, false);
    } else {
      rearm(e, h, h + ops[i].bytes, !(ops[i].rows && ops[i].row_bytes < ops[i].pitch));
    }
  }
}

void *memo_hit(const void *desc, void **p0, void **p1, void **p2, void **p3, int64_t br, uint64_t epoch, hipStream_t s) {
  if (!g_on.load(std::memory_order_relaxed)) return nullptr;
  ThreadState &t = tstate();
  if (t.depth) return nullptr;
  Memo *m = t.memo_set(*p0, *p2);
  if (m->p[2] != *p2 || m->p[0] != *p0) ++m; // the other way of the set
  if (g_trace >= 2) {
    static std::atomic<long> why[9];
    static const bool reg = (atexit([] { fprintf(stderr, "[memo] key-miss desc %ld p0 %ld p1 %ld p2 %ld p3 %ld br %ld epoch %ld stream %ld total %ld\n", why[0].load(), why[1].load(), why[2].load(), why[3].load(), why[4].load(), why[5].load(), why[6].load(), why[7].load(), why[8].load()); }), true);
    (void)reg;
    ++why[8];
    if (m->desc != desc) ++why[0];
    else if (m->p[0] != *p0) ++why[1];
    else if (m->p[1] != *p1) ++why[2];
    else if (m->p[2] != *p2) ++why[3];
    else if (m->p[3] != *p3) ++why[4];
    else if (m->br != br) ++why[5];
    else if (m->epoch != epoch) ++why[6];
    else if (m->stream != s) ++why[7];
  }
  if (m->desc != desc || m->p[0] != *p0 || m->p[1] != *p1 || m->p[2] != *p2 || m->p[3] != *p3 || m->br != br || m->epoch != epoch || m->stream != s) return nullptr;
  announce(t);
  bool good = m->gen == g_struct_gen.load(std::memory_order_acquire);
  for (int i = 0; good && i < 4; ++i)
    if (m->e[i]) good = m->inval[i] == m->e[i]->inval_gen.load(std::memory_order_acquire);
  if (!good) {
    t.active.store(0, std::memory_order_release);
    return nullptr;
  }
  if (m->dev[0]) *p0 = m->dev[0];
  if (m->dev[1]) *p1 = m->dev[1];
  if (m->dev[2]) *p2 = m->dev[2];
  if (m->dev[3]) *p3 = m->dev[3];
  t.depth = 1;
  ++t.n_fast;
  return m;
}

void memo_done(void *token, uint64_t epoch) {
  Memo *m = (Memo *)token;
  ThreadState &t = tstate();
  if (m->wr.bytes && m->pend_epoch != epoch) { // the written footprint goes on this epoch's list once
    m->pend_epoch = epoch;
    const uint64_t key = (uint64_t)m->wr.host * 0x9E3779B97F4A7C15ull ^ (uint64_t)m->wr.bytes * 0xC2B2AE3D27D4EB4Full ^ (uint64_t)m->wr.rows;
    std::lock_guard<SpinLock> lk(t.pmu);
    if (t.seen.insert(key).second) t.pending.push_back(m->wr);
  }
  t.active.store(0, std::memory_order_release);
  t.depth = 0;
}

// after a translate() in asynchronous mode (inside its section: the extents cannot move)
void memo_store(const void *desc, int64_t br, const OpRef *ops, int n, uint64_t epoch, hipStream_t s) {
  if (n != 4) return;
  ThreadState &t = tstate();
  void *orig[4];
  for (int i = 0; i < 4; ++i) orig[i] = ops[i].host ? ops[i].host : *ops[i].ptr;
  Memo *m = t.memo_set(orig[0], orig[2]);
  if (!(m->p[2] == orig[2] && m->p[0] == orig[0])) { // not in way 0: way 1 if it is there or free, else the way that was not stored last
    Memo *w1 = m + 1;
    if ((w1->p[2] == orig[2] && w1->p[0] == orig[0]) || !w1->desc) m = w1;
    else if (m->desc && m->last) m = w1;
  }
  (m == t.memo_set(orig[0], orig[2]) ? m + 1 : m - 1)->last = false;
  Memo fresh;
  fresh.last = true;
  fresh.desc = desc;
  fresh.br = br;
  fresh.gen = g_struct_gen.load(std::memory_order_acquire);
  fresh.epoch = epoch;
  fresh.stream = s;
  for (int i = 0; i < 4; ++i) {
    fresh.p[i] = orig[i];
    if (!ops[i].host) continue; // device memory or null: stays as it is
    if (ops[i].kind != 1) return; // (scratch outputs are a synchronous-mode thing)
    const uintptr_t h = (uintptr_t)ops[i].host;
    Extent *e = nullptr;
    for (Extent *c : g_ext)
      if (h >= c->lo && h + ops[i].bytes <= c->hi) { e = c; break; }
    if (!e || e->polled_epoch.load(std::memory_order_relaxed) != epoch) return;
    fresh.dev[i] = (char *)*ops[i].ptr;
    fresh.e[i] = e;
    fresh.inval[i] = e->inval_gen.load(std::memory_order_acquire);
    if (!e->all_valid(pg_floor(h), pg_ceil(h + ops[i].bytes))) return; // (cannot happen right behind translate())
    if (ops[i].written) fresh.wr = Pending{h, ops[i].bytes, ops[i].rows, ops[i].row_bytes, ops[i].pitch};
  }
  if (m->desc == desc && m->p[2] == orig[2] && m->wr.host == fresh.wr.host) fresh.pend_epoch = m->pend_epoch;
  *m = fresh;
}

void on_sync_point(hipStream_t s) {
  if (!g_on.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lk(g_mu);
  std::vector<Pending> all;
  all.swap(g_orphans);
  for (ThreadState *t = g_threads.load(std::memory_order_acquire); t; t = t->next) {
    std::lock_guard<SpinLock> pl(t->pmu);
    all.insert(all.end(), t->pending.begin(), t->pending.end());
    t->pending.clear();
    t->seen.clear();
  }
  if (all.empty()) return;
  const auto t_begin = std::chrono::steady_clock::now();
  // the union of the written footprints as byte intervals (tiles of one buffer merge into a few long runs)
  std::vector<std::pair<uintptr_t, uintptr_t>> iv;
  for (const Pending &p : all) {
    if (p.rows && p.row_bytes < p.pitch)
      for (size_t r = 0; r < p.rows; ++r) iv.push_back({p.host + r * p.pitch, p.host + r * p.pitch + p.row_bytes});
    else
      iv.push_back({p.host, p.host + p.bytes});
  }
  std::sort(iv.begin(), iv.end());
  std::vector<std::pair<uintptr_t, uintptr_t>> runs;
  for (const auto &v : iv) {
    if (!runs.empty() && v.first <= runs.back().second) runs.back().second = std::max(runs.back().second, v.second);
    else runs.push_back(v);
  }
  struct Done {
    Extent *e;
    uintptr_t a, b;
  };
  std::vector<Done> done;
  std::vector<PmRegion> reg;
  // (adjacent outputs of different extents merge into one run: cut the runs at the extents' - page-aligned - ends again; a stretch
  // that no extent holds any more was given up since, its range changed hands: nothing may be written there)
  {
    std::vector<std::pair<uintptr_t, uintptr_t>> cut;
    for (const auto &r : runs) {
      uintptr_t at = r.first;
      for (Extent *e : g_ext) {
        if (e->hi <= at) continue;
        if (e->lo >= r.second) break;
        if (e->lo > at) st_wb_skipped.fetch_add((int64_t)((pg_ceil(e->lo) - pg_floor(at)) / PG), std::memory_order_relaxed);
        const uintptr_t a = std::max(at, e->lo), b = std::min(r.second, e->hi);
        cut.push_back({a, b});
        at = b;
      }
      if (at < r.second) st_wb_skipped.fetch_add((int64_t)((pg_ceil(r.second) - pg_floor(at)) / PG), std::memory_order_relaxed);
    }
    runs.swap(cut);
  }
  for (const auto &r : runs) {
    const uintptr_t lo = pg_floor(r.first), hi = pg_ceil(r.second);
    Extent *e = find(lo, hi);
    if (!e) continue; // (cannot happen: the run was cut to an extent)
    // what the kernel says about these pages now: every one must still be mapped and ours; a page inside the run that the host has
    // written since the poll is a contract violation (or free()'s list pointers in a chunk freed too early): the host's bytes stay
    reg.clear();
    if (scan(lo, hi, false, 0, 0, reg) != 0) { st_wb_skipped.fetch_add((int64_t)((hi - lo) / PG), std::memory_order_relaxed); continue; }
    uintptr_t covered = 0;
    bool ours = true;
    for (const PmRegion &g : reg) {
      covered += std::min<uintptr_t>(g.end, hi) - std::max<uintptr_t>(g.start, lo);
      ours = ours && (g.categories & PG_WPALLOWED);
    }
    if (!ours || covered != hi - lo) { st_wb_skipped.fetch_add((int64_t)((hi - lo) / PG), std::memory_order_relaxed); continue; }
    // (Round 6, first version: a page inside the run that read as written although it was armed was taken for a host write - a
    // contract violation, or free()'s list pointers in a chunk freed too early - and left alone. WRONG: pages of a buffer that was
    // ever the source / destination of a plain hipMemcpy are write-faulted again by the driver behind everybody's back (the ROCm
    // runtime's pin cache: profiles/r06_wp_vs_hipmemcpy.txt) and read as written, and a skipped write-back is a stale output on the host -
    // tests/test_host_cache_gpu.py caught it inside the whole suite. The device's bytes always go back to pages that are still
    // mapped and still ours; what protects a range that changed hands is the check above, nothing finer.)
    uintptr_t a = r.first;
    uffd_unprotect(lo, hi); // the copies below write these pages: no write-protect fault per page (rearm protects them again)
    if (a < r.second) { copy_back(e, a, r.second - a, 0, 0, 0, s); done.push_back(Done{e, a, r.second}); }
  }
  HC_HIP_OK(hipStreamSynchronize(s));
  const auto t_copied = std::chrono::steady_clock::now();
  for (const Done &d : done) rearm(d.e, d.a, d.b, true);
  HC_TRACE("synchronisation point: %zu footprints in %zu runs written back in %.0f us, protected again in %.0f us", all.size(), runs.size(),
           std::chrono::duration<double, std::micro>(t_copied - t_begin).count(),
           std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_copied).count());
}

int set_enabled(int on) {
  const int prev = g_on.load(std::memory_order_relaxed);
  if (on) {
    if (!prev) {
      std::lock_guard<std::mutex> lk(g_mu);
      if (!open_kernel_interface()) {
        fprintf(stderr, "[tpp-xsmm-hip] host cache unavailable: needs Linux >= 6.7 with userfaultfd(UFFD_USER_MODE_ONLY), UFFD_FEATURE_WP_ASYNC and "
                        "PAGEMAP_SCAN; host operands keep the plain mirror path\n");
        return -1;
      }
      g_on.store(1, std::memory_order_release);
    }
    return prev;
  }
  if (prev) { // off: the caller has drained the stream and written everything back (on_sync_point); forget every extent
    ThreadState &t = tstate();
    std::lock_guard<std::mutex> lk(g_mu);
    exclude_readers(t);
    g_on.store(0, std::memory_order_release);
    while (!g_ext.empty()) {
      Extent *e = g_ext.back();
      uffd_unregister(e->lo, e->hi);
      (void)hipStreamSynchronize(e->stream.load(std::memory_order_relaxed));
      (void)hipFree(e->raw);
      delete e;
      g_ext.pop_back();
    }
    g_rejected.clear();
    g_struct_gen.fetch_add(1, std::memory_order_release);
    readmit_readers();
  }
  return prev;
}

void stats(int64_t out[10]) {
  std::lock_guard<std::mutex> lk(g_mu);
  int64_t bytes = 0;
  for (const Extent *e : g_ext) bytes += (int64_t)(e->cap_hi - e->cap_lo);
  out[0] = (int64_t)g_ext.size();
  out[1] = bytes;
  out[2] = st_upload_bytes.load();
  out[3] = st_scans.load();
  out[4] = st_wb_bytes.load();
  out[5] = st_wb_skipped.load();
  out[6] = st_grows.load();
  out[7] = out[8] = 0;
  for (ThreadState *t = g_threads.load(std::memory_order_acquire); t; t = t->next) {
    out[7] += t->n_fast; // (read while their owners may be counting: approximate by a few while invokes are in flight)
    out[8] += t->n_slow;
  }
  out[9] = st_drops.load();
}

} // namespace hc
} // namespace tpp
