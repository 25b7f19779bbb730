// driver.cpp - TEST INFRASTRUCTURE (tests/test_host_cache.py): the host cache of csrc/host_cache.cpp, driven through the C-ABI on
// HOST buffers the way an unmodified tpp-run would (memref globals / malloc'ed intermediates, lib/TPP/Runner/MLIRBench.cpp:207-246),
// with runtime.cpp + host_cache.cpp compiled unchanged against tests/tsan/fake_hip.cpp ("device" memory = malloc'ed blocks, kernels =
// scalar loops on the launching thread). The REAL kernel interface is used: userfaultfd async write-protect + PAGEMAP_SCAN.
// Every scenario runs twice - cache off (the plain per-invoke mirror) and cache on - and the host-visible results must be identical
// bit for bit; the counters say whether the cache did what it claims (no upload when nothing changed, one page when one page changed).
// Exit code 0 + a last line "OK"; 77 = the kernel lacks the interface (the test skips).
#include "../../include/tpp_xsmm_abi.h"
#include <cstddef>
extern "C" int hipMalloc(void **, size_t); // the fake device allocator of tests/tsan/fake_hip.cpp (hipError_t is an int-sized enum)
extern "C" int hipFree(void *);
#include <fcntl.h>
#include <malloc.h>
#include <sys/mman.h>
#include <unistd.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static int g_fail = 0;
#define EXPECT(c, ...)                                  \
  do {                                                  \
    if (!(c)) {                                         \
      ++g_fail;                                         \
      printf("FAIL %s:%d: %s - ", __FILE__, __LINE__, #c); \
      printf(__VA_ARGS__);                              \
      printf("\n");                                     \
    }                                                   \
  } while (0)

static void fill(float *p, size_t n, unsigned seed) {
  for (size_t i = 0; i < n; ++i) {
    seed = seed * 1664525u + 1013904223u;
    p[i] = (float)((int)((seed >> 16) & 7) - 3) * 0.25f;
  }
}
struct Stats {
  int64_t v[10];
  Stats() { xsmm_hip_host_cache_stats(v); }
  int64_t extents() const { return v[0]; }
  int64_t uploaded() const { return v[2]; }
  int64_t written_back() const { return v[4]; }
  int64_t skipped() const { return v[5]; }
  int64_t fast() const { return v[7]; }
  int64_t dropped() const { return v[9]; }
};

// an allocation that is deliberately NOT page aligned (memref.alloc: 64-byte alignment inside malloc(n + 64))
// HC_ALIGNED=1 (the ThreadSanitizer run): whole pages instead - uploads are page-granular, and the bytes of a shared edge page that
// belong to somebody else's live heap object would read as a (harmless, by design) race to the sanitizer
static const bool g_aligned = getenv("HC_ALIGNED") != nullptr;
static float *host_alloc(size_t n_floats, std::vector<void *> &keep) {
  if (g_aligned) {
    void *p = aligned_alloc(4096, (n_floats * 4 + 4095) & ~(size_t)4095);
    keep.push_back(p);
    return (float *)p;
  }
  char *raw = (char *)malloc(n_floats * 4 + 64 + 192);
  keep.push_back(raw);
  return (float *)((((uintptr_t)raw + 63) & ~(uintptr_t)63) + 192);
}

// ---- scenario 1: synchronous mode (the reference's contract): a loop of whole-matrix BRGEMMs, the host edits inputs in between
static void scenario_sync(bool cache, std::vector<float> &result, int64_t *uploaded_steady, int64_t *uploaded_one_page) {
  xsmm_hip_set_async(0);
  xsmm_hip_set_host_cache(cache ? 1 : 0);
  const int M = 96, N = 160, K = 64, BR = 4;
  std::vector<void *> keep;
  float *A = host_alloc((size_t)M * K * BR, keep), *B = host_alloc((size_t)K * BR * N, keep), *C = host_alloc((size_t)M * N, keep);
  float *bias = host_alloc(N, keep);
  fill(A, (size_t)M * K * BR, 1);
  fill(B, (size_t)K * BR * N, 2);
  fill(C, (size_t)M * N, 3);
  fill(bias, N, 4);
  const int64_t h = xsmm_brgemm_dispatch(XSMM_DTYPE_F32, M, N, K, K * BR, N, N, K, (int64_t)K * N, 0); // C += sum_b A_b B_b
  const int64_t hf = xsmm_fused_brgemm_dispatch(XSMM_DTYPE_F32, M, N, K, K * BR, N, N, K, (int64_t)K * N, XSMM_GEMM_FLAG_BETA_0, 0, XSMM_UNARY_RELU,
                                                XSMM_BINARY_FLAG_BCAST_COL_IN_0, XSMM_BINARY_ADD);
  const int64_t hr = xsmm_unary_dispatch(XSMM_UNARY_RELU, XSMM_DTYPE_F32, M, N, N, N, 0);
  xsmm_brgemm_invoke(XSMM_DTYPE_F32, h, A, 0, B, 0, C, 0, BR);
  result.insert(result.end(), C, C + M * N); // visible on return
  xsmm_brgemm_invoke(XSMM_DTYPE_F32, h, A, 0, B, 0, C, 0, BR); // nothing changed on the host: no upload
  const Stats s0;
  xsmm_brgemm_invoke(XSMM_DTYPE_F32, h, A, 0, B, 0, C, 0, BR);
  const Stats s1;
  *uploaded_steady = s1.uploaded() - s0.uploaded();
  result.insert(result.end(), C, C + M * N);
  // the host edits one element of A (one page), and a stretch of B through a system call (read(2) straight into the operand)
  A[(size_t)M * K * BR / 2] = 7.0f;
  const Stats s2;
  xsmm_brgemm_invoke(XSMM_DTYPE_F32, h, A, 0, B, 0, C, 0, BR);
  const Stats s3;
  *uploaded_one_page = s3.uploaded() - s2.uploaded();
  result.insert(result.end(), C, C + M * N);
  int z = open("/dev/zero", O_RDONLY);
  if (read(z, B + 1000, 8192) != 8192) abort();
  close(z);
  xsmm_brgemm_invoke(XSMM_DTYPE_F32, h, A, 0, B, 0, C, 0, BR);
  result.insert(result.end(), C, C + M * N);
  // the host edits the OUTPUT between two accumulating invokes
  for (int i = 0; i < M * N; i += 97) C[i] = -1.0f;
  xsmm_brgemm_invoke(XSMM_DTYPE_F32, h, A, 0, B, 0, C, 0, BR);
  result.insert(result.end(), C, C + M * N);
  // fused (BETA_0 + bias + relu), then an in-place relu on the output, then the host reads it
  xsmm_fused_brgemm_invoke(XSMM_DTYPE_F32, hf, A, 0, B, 0, C, 0, bias, 0, BR);
  bias[5] = 100.0f;
  xsmm_fused_brgemm_invoke(XSMM_DTYPE_F32, hf, A, 0, B, 0, C, 0, bias, 0, BR);
  for (int i = 0; i < M * N; i += 5) C[i] = -C[i];
  xsmm_unary_invoke(XSMM_DTYPE_F32, hr, C, 0, C, 0);
  result.insert(result.end(), C, C + M * N);
  // tiles of one row-major matrix from four threads (adjacent tiles share pages and rows): zero + brgemm per tile
  {
    const int TS = 32, KB = 32;
    const int64_t hz = xsmm_unary_dispatch(XSMM_UNARY_ZERO, XSMM_DTYPE_F32, TS, TS, N, N, 0);
    const int64_t hg = xsmm_brgemm_dispatch(XSMM_DTYPE_F32, TS, TS, KB, K * BR, N, N, KB, (int64_t)KB * N, 0);
    std::vector<std::thread> th;
    for (int t = 0; t < 4; ++t)
      th.emplace_back([&, t] {
        const int tiles_n = N / TS, tiles = (M / TS) * tiles_n;
        for (int q = t; q < tiles; q += 4) {
          const int i = q / tiles_n, j = q % tiles_n;
          xsmm_unary_invoke(XSMM_DTYPE_F32, hz, C, 0, C, (int64_t)i * TS * N + j * TS);
          xsmm_brgemm_invoke(XSMM_DTYPE_F32, hg, A, (int64_t)i * TS * K * BR, B, j * TS, C, (int64_t)i * TS * N + j * TS, (K * BR) / KB);
        }
      });
    for (auto &x : th) x.join();
    result.insert(result.end(), C, C + M * N);
  }
  xsmm_hip_set_host_cache(0);
  for (void *p : keep) free(p);
}

// ---- scenario 2: asynchronous mode + tile queue: a 3-layer MLP as 32x32x32 tile invokes on HOST buffers (what the compiler emits,
// pass-convert-mlp-to-parallel-tile.mlir:80-88), several timing-loop iterations per synchronisation epoch
static void scenario_async(bool cache, int threads, std::vector<float> &result, int64_t *uploaded_epoch2, int64_t *uploaded_after_edit, int64_t *fast_invokes) {
  xsmm_hip_set_host_cache(cache ? 1 : 0);
  xsmm_hip_set_async(1);
  xsmm_hip_set_tile_queue(getenv("HC_NOQUEUE") ? 0 : 1);
  const int M = 128, W = 128, TS = 32, L = 3, NB = W / TS, MB = M / TS, KBk = W / TS;
  std::vector<void *> keep;
  float *act[L + 1], *Wt[L], *bias[L];
  for (int l = 0; l <= L; ++l) act[l] = host_alloc((size_t)M * W, keep); // packed [MB][KB][32][32]
  for (int l = 0; l < L; ++l) {
    Wt[l] = host_alloc((size_t)W * W, keep); // packed [NB][KB][32][32]
    bias[l] = host_alloc(W, keep);
    fill(Wt[l], (size_t)W * W, 10 + l);
    fill(bias[l], W, 20 + l);
  }
  fill(act[0], (size_t)M * W, 5);
  for (int l = 1; l <= L; ++l) memset(act[l], 0, (size_t)M * W * 4);
  const int64_t h = xsmm_fused_brgemm_dispatch(XSMM_DTYPE_F32, TS, TS, TS, TS, TS, TS, TS * TS, TS * TS, XSMM_GEMM_FLAG_BETA_0, 0, XSMM_UNARY_RELU,
                                               XSMM_BINARY_FLAG_BCAST_COL_IN_0, XSMM_BINARY_ADD);
  auto iteration = [&]() {
    for (int l = 0; l < L; ++l) {
      auto tile = [&](int q) {
        const int i = q / NB, j = q % NB;
        xsmm_fused_brgemm_invoke(XSMM_DTYPE_F32, h, act[l], (int64_t)i * KBk * TS * TS, Wt[l], (int64_t)j * KBk * TS * TS, act[l + 1], (int64_t)(i * NB + j) * TS * TS,
                                 bias[l], j * TS, KBk);
      };
      if (threads <= 1) {
        for (int q = 0; q < MB * NB; ++q) tile(q);
      } else {
        std::vector<std::thread> th;
        for (int t = 0; t < threads; ++t)
          th.emplace_back([&, t] {
            for (int q = t; q < MB * NB; q += threads) tile(q);
          });
        for (auto &x : th) x.join();
      }
    }
  };
  // epoch 1: warm-up + a timed loop, like TppRunnerWrapper.cpp:115-130
  iteration();
  xsmm_hip_synchronize();
  result.insert(result.end(), act[L], act[L] + M * W);
  const Stats s0;
  int64_t t0 = perf_start_timer();
  for (int it = 0; it < 5; ++it) iteration();
  (void)perf_stop_timer(t0);
  const Stats s1;
  *uploaded_epoch2 = s1.uploaded() - s0.uploaded();
  *fast_invokes = s1.fast() - s0.fast();
  result.insert(result.end(), act[L], act[L] + M * W);
  result.insert(result.end(), act[1], act[1] + M * W);
  // between two epochs the host edits one weight and the input
  Wt[1][777] = 3.0f;
  act[0][1] = -2.0f;
  const Stats s2;
  t0 = perf_start_timer();
  for (int it = 0; it < 3; ++it) iteration();
  (void)perf_stop_timer(t0);
  const Stats s3;
  *uploaded_after_edit = s3.uploaded() - s2.uploaded();
  result.insert(result.end(), act[L], act[L] + M * W);
  xsmm_hip_set_tile_queue(0);
  xsmm_hip_set_async(0);
  xsmm_hip_set_host_cache(0);
  for (void *p : keep) free(p);
}

// ---- scenario 3: lifetime without a hook: buffers are freed and their addresses come back with other contents
static void scenario_lifetime(bool cache, std::vector<float> &result, int64_t *dropped_or_reuploaded) {
  xsmm_hip_set_async(0);
  xsmm_hip_set_host_cache(cache ? 1 : 0);
  const int M = 256, N = 256, K = 256;
  const int64_t h = xsmm_gemm_dispatch(XSMM_DTYPE_F32, M, N, K, K, N, N, XSMM_GEMM_FLAG_BETA_0);
  const Stats s0;
  for (int round = 0; round < 4; ++round) {
    // large chunks: glibc serves them with mmap and gives the pages back on free (the next malloc usually returns the same address)
    mallopt(M_MMAP_THRESHOLD, round < 2 ? 128 * 1024 : 64 * 1024 * 1024); // rounds 2-3: from the brk heap instead (free keeps the pages)
    float *A = (float *)malloc((size_t)M * K * 4), *B = (float *)malloc((size_t)K * N * 4), *C = (float *)malloc((size_t)M * N * 4);
    fill(A, (size_t)M * K, 100 + round);
    fill(B, (size_t)K * N, 200 + round);
    xsmm_gemm_invoke(XSMM_DTYPE_F32, h, A, 0, B, 0, C, 0);
    result.insert(result.end(), C, C + M * N);
    free(A);
    free(B);
    free(C);
  }
  // an mmap'ed buffer replaced in place by a file mapping (cannot be tracked: plain path), then by anonymous memory again
  {
    const size_t bytes = (size_t)M * K * 4;
    float *A = (float *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    float *B = (float *)malloc((size_t)K * N * 4), *C = (float *)malloc((size_t)M * N * 4);
    fill(A, (size_t)M * K, 300);
    fill(B, (size_t)K * N, 301);
    xsmm_gemm_invoke(XSMM_DTYPE_F32, h, A, 0, B, 0, C, 0);
    result.insert(result.end(), C, C + M * N);
    char name[] = "/tmp/hc_driver_XXXXXX";
    int fd = mkstemp(name);
    unlink(name);
    if (ftruncate(fd, (off_t)bytes) != 0) abort();
    float *A2 = (float *)mmap(A, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, fd, 0);
    if (A2 != A) abort();
    fill(A, (size_t)M * K, 302);
    xsmm_gemm_invoke(XSMM_DTYPE_F32, h, A, 0, B, 0, C, 0);
    result.insert(result.end(), C, C + M * N);
    float *A3 = (float *)mmap(A, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_FIXED, -1, 0);
    if (A3 != A) abort();
    fill(A, (size_t)M * K, 303);
    xsmm_gemm_invoke(XSMM_DTYPE_F32, h, A, 0, B, 0, C, 0);
    result.insert(result.end(), C, C + M * N);
    close(fd);
    munmap(A, bytes);
    free(B);
    free(C);
  }
  const Stats s1;
  *dropped_or_reuploaded = (s1.dropped() - s0.dropped()) + (s1.uploaded() - s0.uploaded());
  mallopt(M_MMAP_THRESHOLD, 128 * 1024);
  xsmm_hip_set_host_cache(0);
}

// ---- scenario 4 (cache on only): the asynchronous contract broken - an output is unmapped before the synchronisation point. The
// write-back must notice (the kernel no longer vouches for the range) and leave the address alone instead of faulting.
static void scenario_freed_before_sync() {
  xsmm_hip_set_host_cache(1);
  xsmm_hip_set_async(1);
  const int M = 128, N = 128, K = 128;
  const int64_t h = xsmm_gemm_dispatch(XSMM_DTYPE_F32, M, N, K, K, N, N, XSMM_GEMM_FLAG_BETA_0);
  const size_t bytes = (size_t)M * N * 4;
  float *A = (float *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  float *B = (float *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  float *C = (float *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  float *C2 = (float *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  fill(A, (size_t)M * K, 1);
  fill(B, (size_t)K * N, 2);
  xsmm_gemm_invoke(XSMM_DTYPE_F32, h, A, 0, B, 0, C, 0);
  xsmm_gemm_invoke(XSMM_DTYPE_F32, h, A, 0, B, 0, C2, 0);
  munmap(C, bytes); // too early: the contract says after the synchronisation point
  const Stats s0;
  xsmm_hip_synchronize();
  const Stats s1;
  EXPECT(s1.skipped() - s0.skipped() >= (int64_t)(bytes / 4096), "pages not written back: %ld", (long)(s1.skipped() - s0.skipped()));
  float ref = 0.0f;
  for (int k = 0; k < K; ++k) ref += A[k] * B[(size_t)k * N];
  EXPECT(C2[0] == ref, "the surviving output was written back: %g vs %g", C2[0], ref);
  xsmm_hip_set_async(0);
  xsmm_hip_set_host_cache(0);
  munmap(A, bytes);
  munmap(B, bytes);
  munmap(C2, bytes);
}

// ---- scenario 5: a timing loop of ONE whole-matrix BRGEMM in asynchronous mode (BASELINE config 2 on host buffers), then the same
// handle on device pointers with the cache still on, then everything switched off
static void scenario_c2_async() {
  xsmm_hip_set_host_cache(1);
  xsmm_hip_set_async(1);
  const int M = 256, N = 256, K = 64, BR = 4;
  std::vector<void *> keep;
  float *A = host_alloc((size_t)M * K * BR, keep), *B = host_alloc((size_t)K * BR * N, keep), *C = host_alloc((size_t)M * N, keep);
  fill(A, (size_t)M * K * BR, 1);
  fill(B, (size_t)K * BR * N, 2);
  memset(C, 0, (size_t)M * N * 4);
  const int64_t h = xsmm_brgemm_dispatch(XSMM_DTYPE_F32, M, N, K, K * BR, N, N, K, (int64_t)K * N, XSMM_GEMM_FLAG_BETA_0);
  xsmm_brgemm_invoke(XSMM_DTYPE_F32, h, A, 0, B, 0, C, 0, BR);
  xsmm_hip_synchronize();
  const Stats s0;
  const int64_t t0 = perf_start_timer();
  for (int i = 0; i < 200; ++i) xsmm_brgemm_invoke(XSMM_DTYPE_F32, h, A, 0, B, 0, C, 0, BR);
  (void)perf_stop_timer(t0);
  const Stats s1;
  EXPECT(s1.fast() - s0.fast() >= 199, "lock-free translations: %ld", (long)(s1.fast() - s0.fast()));
  EXPECT(s1.uploaded() - s0.uploaded() <= 8 * 4096, "uploaded in the loop: %ld", (long)(s1.uploaded() - s0.uploaded()));
  float *dA, *dB, *dC;
  hipMalloc((void **)&dA, (size_t)M * K * BR * 4);
  hipMalloc((void **)&dB, (size_t)K * BR * N * 4);
  hipMalloc((void **)&dC, (size_t)M * N * 4);
  memcpy(dA, A, (size_t)M * K * BR * 4);
  memcpy(dB, B, (size_t)K * BR * N * 4);
  xsmm_brgemm_invoke(XSMM_DTYPE_F32, h, dA, 0, dB, 0, dC, 0, BR);
  xsmm_hip_synchronize();
  EXPECT(!memcmp(C, dC, (size_t)M * N * 4), "host-buffer result != device-pointer result");
  xsmm_hip_set_async(0);
  xsmm_hip_set_host_cache(0);
  hipFree(dA);
  hipFree(dB);
  hipFree(dC);
  for (void *p : keep) free(p);
}

static bool same(const std::vector<float> &a, const std::vector<float> &b) { return a.size() == b.size() && !memcmp(a.data(), b.data(), a.size() * 4); }

int main() {
  const int prev = xsmm_hip_set_host_cache(1);
  if (prev < 0) {
    printf("SKIP: the kernel lacks userfaultfd WP_ASYNC / PAGEMAP_SCAN\n");
    return 77;
  }
  xsmm_hip_set_host_cache(0);
  {
    std::vector<float> off, on;
    int64_t u0, u1, v0, v1;
    scenario_sync(false, off, &u0, &u1);
    scenario_sync(true, on, &v0, &v1);
    EXPECT(same(off, on), "synchronous scenario: cache on != cache off (%zu values)", on.size());
    // (synchronous mode: the two edge pages of the output just written back - and a neighbour's page they are shared with - are not
    // trusted: other bytes of them may have been written meanwhile. Everything inside stays on the device.)
    EXPECT(v0 <= 4 * 4096, "steady state uploaded %ld bytes, expected at most the output's edge pages", (long)v0);
    EXPECT(v1 - v0 == 4096, "one edited element uploaded %ld bytes more than the steady state, expected one page", (long)(v1 - v0));
    printf("sync: identical to the plain mirror path (%zu values); steady-state upload %ld B, after a one-element edit %ld B\n", on.size(), (long)v0, (long)v1);
  }
  for (int threads : {1, 4}) {
    std::vector<float> off, on;
    int64_t a, b, c, x, y, z;
    scenario_async(false, threads, off, &a, &b, &c);
    scenario_async(true, threads, on, &x, &y, &z);
    EXPECT(same(off, on), "asynchronous scenario (%d threads): cache on != cache off", threads);
    // (an epoch starts with the edge pages of the runs written back at the synchronisation point - three outputs, two edges each - and
    // with whatever edge pages other heap objects were written in meanwhile; the buffers themselves are 112 pages)
    EXPECT(x <= 16 * 4096, "second epoch uploaded %ld bytes, expected at most a few edge pages", (long)x);
    EXPECT(y >= 2 * 4096 && y <= 24 * 4096, "two edited elements: %ld bytes uploaded (%ld without an edit)", (long)y, (long)x);
    EXPECT(z >= 5 * 3 * 16 - 3 * 16, "lock-free translations in the timed loop: %ld", (long)z);
    printf("async + tile queue, %d caller(s): identical; second-epoch upload %ld B; after two edits %ld B; %ld invokes translated lock-free\n", threads, (long)x, (long)y,
           (long)z);
  }
  {
    std::vector<float> off, on;
    int64_t a, b;
    scenario_lifetime(false, off, &a);
    scenario_lifetime(true, on, &b);
    EXPECT(same(off, on), "lifetime scenario: cache on != cache off");
    printf("lifetime: freed / re-mapped buffers never served from a stale mirror (%zu values identical)\n", on.size());
  }
  scenario_c2_async();
  printf("C2 loop, asynchronous: lock-free after the first invoke, nothing uploaded, equal to the device-pointer result\n");
  scenario_freed_before_sync();
  printf("freed before the synchronisation point: write-back skipped, no fault\n");
  if (g_fail) {
    printf("%d FAILURE(S)\n", g_fail);
    return 1;
  }
  printf("OK\n");
  return 0;
}
