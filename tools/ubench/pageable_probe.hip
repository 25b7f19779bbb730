// pageable_probe - can this box's GPU use plain host memory (malloc / memref.global-style buffers) in place?
// Round 6, review item 1: an UNMODIFIED tpp-run hands host pointers to the xsmm entry points (MLIRBench.cpp:207-246),
// and mirroring them over PCIe per invoke makes C2 run at 7 TFLOP/s. This probe measures every mechanism that could
// avoid the per-invoke copy, so the decision (build / kill) is taken on numbers:
//   1. device attributes: pageable memory access (HMM / XNACK), managed memory, host-register support
//   2. hipMemAdvise / hipMemPrefetchAsync on malloc memory (needs HMM)
//   3. hipHostRegister of malloc memory: cost, then a kernel reading it in place over PCIe (zero copy) and
//      hipMemcpyAsync from it (pinned DMA) against the pageable copy
//   4. hipMallocManaged: kernel access after prefetch, host read-back cost
//   5. host page protection as a coherence mechanism: mprotect cost by range size, SIGSEGV round trip,
//      userfaultfd availability (write-protect / async write-protect / unmap events)
//   6. LAST and only if attribute 1 says yes: a kernel reading + writing plain malloc memory (in a child process:
//      without HMM that access is a GPU memory fault that ends the process)
#include <hip/hip_runtime.h>
#include <chrono>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <linux/userfaultfd.h>
#include <sys/ioctl.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <sys/utsname.h>
#include <sys/wait.h>
#include <unistd.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("  %s -> %s\n", #x, hipGetErrorString(e_)); (void)hipGetLastError(); } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void sum_kernel(const float4 *p, size_t n4, float *out) {
  float s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = p[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) out[0] = s;
}
__global__ void inc_kernel(float *p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] += 1.0f;
}

static volatile sig_atomic_t g_faults = 0;
static char *g_prot_base = nullptr;
static size_t g_prot_len = 0;
static void on_segv(int, siginfo_t *si, void *) {
  char *a = (char *)si->si_addr;
  if (g_prot_base && a >= g_prot_base && a < g_prot_base + g_prot_len) {
    ++g_faults;
    mprotect(g_prot_base, g_prot_len, PROT_READ | PROT_WRITE);
    return;
  }
  _exit(99);
}

static void touch_mode();
int main(int argc, char **argv) {
  if (argc > 1 && !strcmp(argv[1], "--touch")) touch_mode();
  struct utsname u;
  uname(&u);
  printf("kernel %s %s\n", u.sysname, u.release);
  const char *x = getenv("HSA_XNACK");
  printf("HSA_XNACK=%s\n", x ? x : "(unset)");
  int dev = 0;
  CHECK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, dev));
  printf("device %s (%s)\n", prop.name, prop.gcnArchName);
  printf("== 1. attributes\n");
  struct { const char *n; hipDeviceAttribute_t a; } attrs[] = {
      {"PageableMemoryAccess", hipDeviceAttributePageableMemoryAccess},
      {"PageableMemoryAccessUsesHostPageTables", hipDeviceAttributePageableMemoryAccessUsesHostPageTables},
      {"ConcurrentManagedAccess", hipDeviceAttributeConcurrentManagedAccess},
      {"ManagedMemory", hipDeviceAttributeManagedMemory},
      {"DirectManagedMemAccessFromHost", hipDeviceAttributeDirectManagedMemAccessFromHost},
      {"CanMapHostMemory", hipDeviceAttributeCanMapHostMemory},
      {"HostRegisterSupported", hipDeviceAttributeHostRegisterSupported},
      {"CanUseHostPointerForRegisteredMem", hipDeviceAttributeCanUseHostPointerForRegisteredMem},
  };
  int pageable = 0;
  for (auto &a : attrs) {
    int v = -1;
    hipError_t e = hipDeviceGetAttribute(&v, a.a, dev);
    printf("  %-42s %d%s\n", a.n, v, e == hipSuccess ? "" : " (query failed)");
    (void)hipGetLastError();
    if (a.a == hipDeviceAttributePageableMemoryAccess && e == hipSuccess) pageable = v;
  }
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  float *dout;
  CHECK(hipMalloc(&dout, 256));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const size_t MB = 1 << 20;

  printf("== 2. hipMemAdvise / hipMemPrefetchAsync on malloc memory (needs HMM)\n");
  {
    size_t n = 4 * MB;
    char *h = (char *)aligned_alloc(4096, n);
    memset(h, 1, n);
    hipError_t e = hipMemAdvise(h, n, hipMemAdviseSetPreferredLocation, dev);
    printf("  hipMemAdvise(SetPreferredLocation) -> %s\n", hipGetErrorString(e));
    (void)hipGetLastError();
    e = hipMemPrefetchAsync(h, n, dev, s);
    printf("  hipMemPrefetchAsync(to device)     -> %s\n", hipGetErrorString(e));
    (void)hipGetLastError();
    hipStreamSynchronize(s);
    (void)hipGetLastError();
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    e = hipPointerGetAttributes(&at, h);
    printf("  hipPointerGetAttributes(malloc ptr) -> %s type %d\n", hipGetErrorString(e), (int)at.type);
    (void)hipGetLastError();
    free(h);
  }

  printf("== 3. hipHostRegister of malloc memory\n");
  for (size_t n : {4 * MB, 64 * MB}) {
    char *h = (char *)aligned_alloc(64, n);
    memset(h, 1, n);
    void *d;
    CHECK(hipMalloc(&d, n));
    // pageable copy first
    CHECK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s));
    hipStreamSynchronize(s);
    double t0 = now_us();
    for (int i = 0; i < 5; ++i) CHECK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s));
    hipStreamSynchronize(s);
    double tp = (now_us() - t0) / 5;
    t0 = now_us();
    hipError_t e = hipHostRegister(h, n, hipHostRegisterDefault);
    double treg = now_us() - t0;
    printf("  %3zu MiB: pageable H2D %.1f us (%.1f GB/s); hipHostRegister -> %s in %.0f us\n", n / MB, tp, n / tp / 1e3, hipGetErrorString(e), treg);
    (void)hipGetLastError();
    if (e == hipSuccess) {
      t0 = now_us();
      for (int i = 0; i < 5; ++i) CHECK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s));
      hipStreamSynchronize(s);
      double tr = (now_us() - t0) / 5;
      void *dp = nullptr;
      CHECK(hipHostGetDevicePointer(&dp, h, 0));
      float ms = 0;
      if (dp) {
        hipLaunchKernelGGL(sum_kernel, dim3(1024), dim3(256), 0, s, (const float4 *)dp, n / 16, dout);
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(sum_kernel, dim3(1024), dim3(256), 0, s, (const float4 *)dp, n / 16, dout);
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        hipEventElapsedTime(&ms, e0, e1);
      }
      printf("           registered H2D %.1f us (%.1f GB/s); kernel reading it IN PLACE %.1f us (%.1f GB/s); device ptr %s host ptr\n", tr, n / tr / 1e3,
             ms * 1e3 / 3, ms > 0 ? n / (ms * 1e3 / 3) / 1e3 : 0.0, dp == (void *)h ? "==" : "!=");
      t0 = now_us();
      CHECK(hipHostUnregister(h));
      printf("           hipHostUnregister %.0f us\n", now_us() - t0);
    }
    // device-resident read for scale
    float ms = 0;
    hipLaunchKernelGGL(sum_kernel, dim3(1024), dim3(256), 0, s, (const float4 *)d, n / 16, dout);
    hipEventRecord(e0, s);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(sum_kernel, dim3(1024), dim3(256), 0, s, (const float4 *)d, n / 16, dout);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    hipEventElapsedTime(&ms, e0, e1);
    printf("           the same kernel on the device copy %.1f us\n", ms * 1e3 / 3);
    CHECK(hipFree(d));
    free(h);
  }

  printf("== 4. hipMallocManaged\n");
  {
    size_t n = 16 * MB;
    float *m = nullptr;
    hipError_t e = hipMallocManaged((void **)&m, n, hipMemAttachGlobal);
    printf("  hipMallocManaged -> %s\n", hipGetErrorString(e));
    (void)hipGetLastError();
    if (e == hipSuccess && m) {
      for (size_t i = 0; i < n / 4; ++i) m[i] = 1.0f;
      double t0 = now_us();
      e = hipMemPrefetchAsync(m, n, dev, s);
      hipStreamSynchronize(s);
      printf("  prefetch to device -> %s, %.0f us\n", hipGetErrorString(e), now_us() - t0);
      (void)hipGetLastError();
      float ms = 0;
      hipLaunchKernelGGL(inc_kernel, dim3(1024), dim3(256), 0, s, m, n / 4);
      hipStreamSynchronize(s);
      hipEventRecord(e0, s);
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(inc_kernel, dim3(1024), dim3(256), 0, s, m, n / 4);
      hipEventRecord(e1, s);
      hipStreamSynchronize(s);
      hipEventElapsedTime(&ms, e0, e1);
      printf("  kernel read+write of 16 MiB managed: %.1f us per launch (%.1f GB/s)\n", ms * 1e3 / 3, 2.0 * n / (ms * 1e3 / 3) / 1e3);
      t0 = now_us();
      double acc = 0;
      for (size_t i = 0; i < n / 4; i += 1024) acc += m[i];
      printf("  host read-back (one float per page): %.0f us, value %.0f (expect 5 per sample)\n", now_us() - t0, acc / (n / 4 / 1024));
      hipFree(m);
    }
  }

  printf("== 5. host page protection as the coherence mechanism\n");
  {
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = on_segv;
    sa.sa_flags = SA_SIGINFO;
    sigaction(SIGSEGV, &sa, nullptr);
    for (size_t n : {64 * 1024ul, 1 * MB, 4 * MB, 64 * MB}) {
      char *h = (char *)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      memset(h, 1, n);
      const int R = 200;
      double t0 = now_us();
      for (int i = 0; i < R; ++i) {
        mprotect(h, n, PROT_READ);
        mprotect(h, n, PROT_READ | PROT_WRITE);
      }
      double tpair = (now_us() - t0) / R;
      t0 = now_us();
      for (int i = 0; i < R; ++i) mprotect(h, n, PROT_READ | PROT_WRITE); // no change
      double tsame = (now_us() - t0) / R;
      g_prot_base = h;
      g_prot_len = n;
      g_faults = 0;
      t0 = now_us();
      for (int i = 0; i < R; ++i) {
        mprotect(h, n, PROT_NONE);
        ((volatile char *)h)[n / 2] = 2; // fault -> handler unprotects -> retry
      }
      double tfault = (now_us() - t0) / R;
      printf("  %6zu KiB: mprotect RO+RW pair %.2f us; unchanged mprotect %.2f us; PROT_NONE + fault + handler(mprotect RW) %.2f us (%d faults)\n", n / 1024, tpair,
             tsame, tfault, (int)g_faults);
      g_prot_base = nullptr;
      munmap(h, n);
    }
    FILE *f = fopen("/proc/sys/vm/unprivileged_userfaultfd", "r");
    int v = -1;
    if (f) { if (fscanf(f, "%d", &v) != 1) v = -1; fclose(f); }
    printf("  /proc/sys/vm/unprivileged_userfaultfd = %d, uid %d\n", v, (int)getuid());
    for (int flags : {O_CLOEXEC | O_NONBLOCK, O_CLOEXEC | O_NONBLOCK | UFFD_USER_MODE_ONLY}) {
      int fd = (int)syscall(SYS_userfaultfd, flags);
      if (fd < 0) { printf("  userfaultfd(%s) -> errno %d (%s)\n", (flags & UFFD_USER_MODE_ONLY) ? "USER_MODE_ONLY" : "full", errno, strerror(errno)); continue; }
      struct uffdio_api api;
      memset(&api, 0, sizeof(api));
      api.api = UFFD_API;
      api.features = 0;
      int r = ioctl(fd, UFFDIO_API, &api);
      printf("  userfaultfd(%s) ok; UFFDIO_API -> %d, features 0x%llx:%s%s%s%s\n", (flags & UFFD_USER_MODE_ONLY) ? "USER_MODE_ONLY" : "full", r,
             (unsigned long long)api.features, (api.features & UFFD_FEATURE_PAGEFAULT_FLAG_WP) ? " WP" : "",
#ifdef UFFD_FEATURE_WP_ASYNC
             (api.features & UFFD_FEATURE_WP_ASYNC) ? " WP_ASYNC" : "",
#else
             " (WP_ASYNC unknown to these headers)",
#endif
             (api.features & UFFD_FEATURE_EVENT_UNMAP) ? " EVENT_UNMAP" : "", (api.features & UFFD_FEATURE_EVENT_REMAP) ? " EVENT_REMAP" : "");
      close(fd);
    }
  }

  printf("== 6. kernel on plain malloc memory\n");
  if (!pageable) {
    printf("  skipped: PageableMemoryAccess = 0 - the access would be a GPU memory fault that ends the process\n");
  } else {
    fflush(stdout);
    pid_t pid = fork();
    if (pid == 0) {
      // a fresh HIP context in the child is not guaranteed after fork: exec ourselves with a marker instead
      execl("/proc/self/exe", "pageable_probe", "--touch", (char *)nullptr);
      _exit(98);
    }
    int st = 0;
    waitpid(pid, &st, 0);
    printf("  child exit status %d (signal %d)\n", WIFEXITED(st) ? WEXITSTATUS(st) : -1, WIFSIGNALED(st) ? WTERMSIG(st) : 0);
  }
  return 0;
}

// child mode: `pageable_probe --touch`
static void touch_mode() {
  size_t bytes = 16 << 20;
  float *h = (float *)aligned_alloc(64, bytes);
  for (size_t i = 0; i < bytes / 4; ++i) h[i] = 1.0f;
  hipStream_t s;
  hipStreamCreate(&s);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) {
      hipError_t e = hipMemPrefetchAsync(h, bytes, 0, s);
      printf("  child: hipMemPrefetchAsync -> %s\n", hipGetErrorString(e));
    }
    hipEventRecord(e0, s);
    hipLaunchKernelGGL(inc_kernel, dim3(1024), dim3(256), 0, s, h, bytes / 4);
    hipEventRecord(e1, s);
    hipError_t e = hipStreamSynchronize(s);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("  child: first-touch launch %d -> %s, %.1f us\n", pass, hipGetErrorString(e), ms * 1e3);
    hipEventRecord(e0, s);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(inc_kernel, dim3(1024), dim3(256), 0, s, h, bytes / 4);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    hipEventElapsedTime(&ms, e0, e1);
    printf("  child: steady launch %.1f us (%.1f GB/s)\n", ms * 1e3 / 3, 2.0 * bytes / (ms * 1e3 / 3) / 1e3);
    double t0 = now_us();
    double acc = 0;
    for (size_t i = 0; i < bytes / 4; i += 1024) acc += h[i];
    printf("  child: host read-back %.0f us, value %.1f\n", now_us() - t0, acc / (bytes / 4 / 1024));
  }
  fflush(stdout);
  _exit(0);
}
