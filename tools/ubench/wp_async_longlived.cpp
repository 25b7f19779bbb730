// wp_async_probe - the kernel interface the host cache (csrc/host_cache.cpp) stands on, measured by itself (no HIP):
// userfaultfd in USER_MODE_ONLY with UFFD_FEATURE_WP_ASYNC | WP_UNPOPULATED (no fault ever reaches user space: the kernel resolves a
// write to a write-protected page itself and remembers it) + the PAGEMAP_SCAN ioctl of /proc/self/pagemap ("which pages of this range
// were written since I last asked, and protect them again" in one call; Linux >= 6.7).
//   1. cost of a scan over 64 KiB .. 64 MiB: nothing written / one page written / everything written
//   2. what a scan says about a range that was unmapped and mapped again behind our back, and about a free()d + re-malloc()ed chunk
//   3. a write through a system call (read(2) into the range) - with sync write-protect faults that would be EFAULT; here it must work
// g++ -O2 -std=c++17 wp_async_probe.cpp -o wp_async_probe.out
#include <chrono>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <linux/userfaultfd.h>
#include <sys/ioctl.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <vector>

#ifndef UFFD_FEATURE_WP_UNPOPULATED
#define UFFD_FEATURE_WP_UNPOPULATED (1 << 13)
#endif
#ifndef UFFD_FEATURE_WP_ASYNC
#define UFFD_FEATURE_WP_ASYNC (1 << 15)
#endif
#ifndef UFFD_USER_MODE_ONLY
#define UFFD_USER_MODE_ONLY 1
#endif
// linux/fs.h of 6.7+ (restated: the container's headers may be older than the running kernel)
struct pm_region { uint64_t start, end, categories; };
struct pm_scan { uint64_t size, flags, start, end, walk_end, vec, vec_len, max_pages, category_inverted, category_mask, category_anyof_mask, return_mask; };
#define PM_IOCTL _IOWR('f', 16, struct pm_scan)
enum { PG_WPALLOWED = 1, PG_WRITTEN = 2, PG_PRESENT = 8 };
enum { SCAN_WP_MATCHING = 1, SCAN_CHECK_WPASYNC = 2 };

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int g_uffd = -1, g_pm = -1;

static long scan(void *p, size_t n, bool reprotect, std::vector<pm_region> &out, uint64_t mask = PG_WRITTEN, uint64_t inverted = 0) {
  out.resize(256);
  pm_scan a;
  memset(&a, 0, sizeof a);
  a.size = sizeof a;
  a.flags = reprotect ? (SCAN_WP_MATCHING | SCAN_CHECK_WPASYNC) : 0;
  a.start = (uint64_t)p;
  a.end = (uint64_t)p + n;
  a.vec = (uint64_t)out.data();
  a.vec_len = out.size();
  a.category_mask = mask;
  a.category_inverted = inverted;
  a.return_mask = PG_WPALLOWED | PG_WRITTEN | PG_PRESENT;
  long r = ioctl(g_pm, PM_IOCTL, &a);
  out.resize(r < 0 ? 0 : r);
  return r < 0 ? -errno : r;
}
static int reg(void *p, size_t n) {
  uffdio_register r;
  memset(&r, 0, sizeof r);
  r.range.start = (uint64_t)p;
  r.range.len = n;
  r.mode = UFFDIO_REGISTER_MODE_WP;
  return ioctl(g_uffd, UFFDIO_REGISTER, &r) ? -errno : 0;
}
static size_t pages_of(const std::vector<pm_region> &v) {
  size_t s = 0;
  for (auto &r : v) s += (r.end - r.start) / 4096;
  return s;
}

int main_orig() {
  g_uffd = (int)syscall(SYS_userfaultfd, O_CLOEXEC | O_NONBLOCK | UFFD_USER_MODE_ONLY);
  if (g_uffd < 0) { printf("userfaultfd: %s\n", strerror(errno)); return 1; }
  uffdio_api api;
  memset(&api, 0, sizeof api);
  api.api = UFFD_API;
  api.features = UFFD_FEATURE_WP_ASYNC | UFFD_FEATURE_WP_UNPOPULATED;
  if (ioctl(g_uffd, UFFDIO_API, &api)) { printf("UFFDIO_API(WP_ASYNC | WP_UNPOPULATED): %s\n", strerror(errno)); return 1; }
  printf("uffd features granted 0x%llx\n", (unsigned long long)api.features);
  g_pm = open("/proc/self/pagemap", O_RDONLY | O_CLOEXEC);
  if (g_pm < 0) { printf("pagemap: %s\n", strerror(errno)); return 1; }
  std::vector<pm_region> v;
  printf("== 1. scan cost (get written pages + protect them again)\n");
  for (size_t n : {64ul << 10, 1ul << 20, 4ul << 20, 64ul << 20}) {
    char *h = (char *)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    memset(h, 1, n);
    int e = reg(h, n);
    long first = scan(h, n, true, v);
    size_t fp = pages_of(v);
    const int R = 200;
    double t0 = now_us();
    for (int i = 0; i < R; ++i) scan(h, n, true, v);
    double clean = (now_us() - t0) / R;
    t0 = now_us();
    for (int i = 0; i < R; ++i) { h[n / 2] = (char)i; scan(h, n, true, v); }
    double one = (now_us() - t0) / R;
    size_t onep = pages_of(v);
    t0 = now_us();
    for (int i = 0; i < 20; ++i) { memset(h, i, n); scan(h, n, true, v); }
    double all = (now_us() - t0) / 20;
    t0 = now_us();
    for (int i = 0; i < 20; ++i) memset(h, i, n);
    double ms = (now_us() - t0) / 20;
    printf("  %6zu KiB: register %d, first scan %ld regions / %zu pages; clean scan %.2f us; one page written %.2f us (%zu page); all written: memset + scan %.1f us (memset alone on unprotected-after-first-touch pages %.1f us)\n",
           n >> 10, e, first, fp, clean, one, onep, all, ms);
    munmap(h, n);
  }
  printf("== 2. lifetime: what a scan says after the range changed hands\n");
  {
    size_t n = 1 << 20;
    char *h = (char *)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    memset(h, 1, n);
    reg(h, n);
    scan(h, n, true, v);
    munmap(h, n);
    long r = scan(h, n, true, v);
    printf("  unmapped range: scan -> %ld (%s)\n", r, r < 0 ? strerror(-r) : "regions");
    char *h2 = (char *)mmap(h, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_FIXED, -1, 0);
    h2[0] = 5;
    r = scan(h, n, true, v);
    printf("  mapped again at the same address (not registered): scan with CHECK_WPASYNC -> %ld (%s)\n", r, r < 0 ? strerror(-r) : "regions");
    r = scan(h, n, false, v, PG_WPALLOWED, PG_WPALLOWED); // pages that are NOT wp-allowed
    printf("  ... pages that are not WPALLOWED: %ld regions, %zu pages (of %zu)\n", r, pages_of(v), n / 4096);
    munmap(h2, n);
    // a heap chunk: registered, scanned, freed, malloc'ed again, written by its new owner
    mallopt(-3 /*M_MMAP_THRESHOLD*/, 64 << 20);
    char *keep = (char *)malloc(256);
    char *c = (char *)malloc(n + 4096);
    memset(c, 1, n + 4096);
    char *lo = (char *)(((uintptr_t)c + 4095) & ~(uintptr_t)4095);
    int e = reg(lo, n);
    scan(lo, n, true, v);
    free(c);
    r = scan(lo, n, true, v);
    printf("  heap chunk (register %d): after free() the scan reports %ld regions, %zu written pages (free's own list pointers / trim)\n", e, r, pages_of(v));
    char *c2 = (char *)malloc(n + 4096);
    memset(c2, 7, n + 4096);
    r = scan(lo, n, true, v);
    printf("  re-malloc()ed (%s address) and filled by its new owner: %ld regions, %zu written pages\n", c2 == c ? "same" : "another", r, pages_of(v));
    free(c2);
    free(keep);
  }
  printf("== 3. a write through a system call\n");
  {
    size_t n = 1 << 20;
    char *h = (char *)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    memset(h, 1, n);
    reg(h, n);
    scan(h, n, true, v);
    int z = open("/dev/zero", O_RDONLY);
    ssize_t got = read(z, h + 8192, 4096 * 3);
    int err = errno;
    close(z);
    long r = scan(h, n, true, v);
    printf("  read(/dev/zero) into protected pages -> %zd (%s); the scan then reports %ld regions, %zu written pages\n", got, got < 0 ? strerror(err) : "ok", r,
           pages_of(v));
  }
  return 0;
}

#include <sys/wait.h>
static void show(const char *what, char *lo, size_t len) {
  std::vector<pm_region> v;
  long r = scan(lo, len, true, v);
  printf("  %-46s %ld regions, %zu of %zu pages written\n", what, r, pages_of(v), len / 4096);
  fflush(stdout);
}
int main(int argc, char **argv) {
  g_uffd = (int)syscall(SYS_userfaultfd, O_CLOEXEC | O_NONBLOCK | UFFD_USER_MODE_ONLY);
  uffdio_api api; memset(&api, 0, sizeof api); api.api = UFFD_API; api.features = UFFD_FEATURE_WP_ASYNC | UFFD_FEATURE_WP_UNPOPULATED;
  if (ioctl(g_uffd, UFFDIO_API, &api)) { printf("api fail\n"); return 1; }
  g_pm = open("/proc/self/pagemap", O_RDONLY | O_CLOEXEC);
  { FILE *f = fopen("/proc/sys/kernel/numa_balancing", "r"); int v = -1; if (f) { if (fscanf(f, "%d", &v) != 1) v = -1; fclose(f); } printf("numa_balancing = %d\n", v); }
  { FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/khugepaged/scan_sleep_millisecs", "r"); int v = -1; if (f) { if (fscanf(f, "%d", &v) != 1) v = -1; fclose(f); } printf("khugepaged scan_sleep_millisecs = %d\n", v); }
  const size_t len = (4u << 20) + 4096;
  for (int huge = 0; huge < 2; ++huge) {
    char *raw = (char *)malloc(len + (4u << 20));
    char *lo = (char *)(((uintptr_t)raw + 4095) & ~(uintptr_t)4095) + 0x1000;
    if (huge) printf("madvise(HUGEPAGE) -> %d\n", madvise((void *)(((uintptr_t)raw + 4095) & ~(uintptr_t)4095), len + (2u << 20), MADV_HUGEPAGE));
    memset(raw, 1, len + (4u << 20));
    printf("%s buffer at %p: register %d\n", huge ? "huge-page-advised" : "plain", (void *)lo, reg(lo, len));
    show("first scan", lo, len);
    show("second scan", lo, len);
    std::vector<char> tmp(len);
    for (int i = 0; i < 4; ++i) {
      sleep(3);
      memcpy(tmp.data(), lo, len);
      show("after 3 s + a read of the whole range", lo, len);
    }
    pid_t p = fork();
    if (p == 0) { execl("/bin/true", "true", (char *)nullptr); _exit(0); }
    int st; waitpid(p, &st, 0);
    show("after fork + exec of /bin/true", lo, len);
    memcpy(tmp.data(), lo, len);
    show("... and a read", lo, len);
    p = fork();
    if (p == 0) { volatile char c = lo[5]; (void)c; _exit(0); }
    waitpid(p, &st, 0);
    show("after fork, child reads + exits", lo, len);
    lo[100] = 3;
    show("after ONE write by the parent", lo, len);
    show("then", lo, len);
  }
  return 0;
}
