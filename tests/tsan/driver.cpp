// driver.cpp - TEST INFRASTRUCTURE (tests/test_tsan_scheduler.py): drives the C-ABI the way the reference's
// compiled code does - several OpenMP-style worker threads invoking tile kernels, a barrier between layers
// (scf.parallel + omp.wsloop, pass-convert-mlp-to-parallel-tile.mlir:80-88) - against runtime.cpp built with
// ThreadSanitizer on top of tests/tsan/fake_hip.cpp. Exit code 0 = results identical to a serial run; TSAN
// reports (and fails the process) on any race in the ring / scheduler / mirror code.
#include "../../include/tpp_xsmm_abi.h"
#include <dirent.h>
#include <pthread.h>
#include <atomic>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

extern "C" int hipMalloc(void **, size_t); // the fake device allocator of fake_hip.cpp (hipError_t is an int-sized enum)
extern "C" int hipFree(void *);

static const int NT = 8, M = 128, N = 128, K = 128, TS = 32, KB = 32, LAYERS = 3;

static int launch_thread_exists() { // the runtime's launch thread (rt_launcher.h): exists while complete replays keep coming (+ ~2 s)
  int64_t st[2] = {0, 0};
  xsmm_hip_launch_thread_stats(st);
  return (int)st[1];
}
static int thread_count() { // threads of the process, the launch thread not counted
  const int lt = launch_thread_exists();
  int n = 0;
  if (DIR *d = opendir("/proc/self/task")) {
    while (dirent *e = readdir(d)) n += e->d_name[0] != '.';
    closedir(d);
  }
  return n - lt;
}

static float *dev_alloc(size_t n) {
  void *p = nullptr;
  if (hipMalloc(&p, n * sizeof(float)) != 0) abort();
  return (float *)p;
}

static void fill(float *p, size_t n, unsigned seed, int zero_every) {
  for (size_t i = 0; i < n; ++i) {
    seed = seed * 1664525u + 1013904223u;
    p[i] = (int)((seed >> 24) % (unsigned)zero_every) ? 0.0f : (float)((int)((seed >> 16) & 3) - 1);
  }
}

// serial reference with the fake kernels' summation order: zero, += batches in order, relu
static void serial_layer(const float *X, const float *W, float *Y) {
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < N; ++j) {
      float acc = 0.0f;
      for (int kk = 0; kk < K; ++kk) acc += X[i * K + kk] * W[kk * N + j];
      Y[i * N + j] = acc > 0.0f ? acc : 0.0f;
    }
}

// one layer's tiles, striped over the threads; every thread dispatches its own handles (dispatch cache under fire)
static void layer_tiles(int tid, float *X, float *W, float *Y) {
  const int64_t hz = xsmm_unary_dispatch(XSMM_UNARY_ZERO, XSMM_DTYPE_F32, TS, TS, N, N, 0);
  const int64_t hg = xsmm_brgemm_dispatch(XSMM_DTYPE_F32, TS, TS, KB, K, N, N, KB, (int64_t)KB * N, 0);
  const int64_t hr = xsmm_unary_dispatch(XSMM_UNARY_RELU, XSMM_DTYPE_F32, TS, TS, N, N, 0);
  const int tiles_n = N / TS, tiles = (M / TS) * tiles_n;
  for (int t = tid; t < tiles; t += NT) {
    const int i = t / tiles_n, j = t % tiles_n;
    const int64_t oc = (int64_t)i * TS * N + j * TS;
    xsmm_unary_invoke(XSMM_DTYPE_F32, hz, Y, oc, Y, oc);
    xsmm_brgemm_invoke(XSMM_DTYPE_F32, hg, X, (int64_t)i * TS * K, W, j * TS, Y, oc, K / KB);
    xsmm_unary_invoke(XSMM_DTYPE_F32, hr, Y, oc, Y, oc);
  }
}

static int run_mlp(bool device_operands, int reps, const char *what) {
  std::vector<float *> act(LAYERS + 1), w(LAYERS), ref(LAYERS + 1);
  auto alloc = [&](size_t n) { return device_operands ? dev_alloc(n) : (float *)malloc(n * sizeof(float)); };
  for (int l = 0; l <= LAYERS; ++l) {
    act[l] = alloc((size_t)M * N);
    ref[l] = (float *)malloc((size_t)M * N * sizeof(float));
  }
  for (int l = 0; l < LAYERS; ++l) {
    w[l] = alloc((size_t)K * N);
    fill(w[l], (size_t)K * N, 77u + l, 4);
  }
  fill(act[0], (size_t)M * K, 5u, 2);
  memcpy(ref[0], act[0], (size_t)M * K * sizeof(float));
  for (int l = 0; l < LAYERS; ++l) serial_layer(ref[l], w[l], ref[l + 1]);

  int bad = 0;
  pthread_barrier_t bar;
  pthread_barrier_init(&bar, nullptr, NT);
  for (int rep = 0; rep < reps; ++rep) {
    for (int l = 1; l <= LAYERS; ++l) memset(act[l], 0xff, (size_t)M * N * sizeof(float));
    std::vector<std::thread> th;
    for (int tid = 0; tid < NT; ++tid)
      th.emplace_back([&, tid] {
        for (int l = 0; l < LAYERS; ++l) {
          layer_tiles(tid, act[l], w[l], act[l + 1]);
          pthread_barrier_wait(&bar); // the implicit barrier at the end of the omp.wsloop
        }
        if (tid == NT - 1) xsmm_hip_synchronize(); // perf_stop_timer / the end of the entry point
      });
    for (auto &t : th) t.join();
    for (int l = 1; l <= LAYERS; ++l)
      if (memcmp(act[l], ref[l], (size_t)M * N * sizeof(float))) ++bad;
  }
  pthread_barrier_destroy(&bar);
  printf("%-46s reps %d: %s\n", what, reps, bad ? "MISMATCH" : "identical to the serial run");
  for (int l = 0; l <= LAYERS; ++l) {
    device_operands ? (void)hipFree(act[l]) : free(act[l]);
    free(ref[l]);
  }
  for (int l = 0; l < LAYERS; ++l) device_operands ? (void)hipFree(w[l]) : free(w[l]);
  return bad;
}

// The steady state of compiled code: fused BRGEMM tiles (BETA_0 + relu folded in, one invoke per tile, 32 tiles per layer), the same
// invokes iteration after iteration - recorded once, then replayed: members arrive through the direct window without the queue's
// lock while the first caller of the next layer closes it. In odd repetitions one thread also issues an invoke the recorded group
// does not know (a zero fill of a scratch tile) in the middle of a layer: the replay is abandoned under the other callers' feet.
static int run_fused(int reps, const char *what) {
  const int M2 = 256;
  std::vector<float *> act(LAYERS + 1), w(LAYERS), ref(LAYERS + 1);
  for (int l = 0; l <= LAYERS; ++l) {
    act[l] = dev_alloc((size_t)M2 * N);
    ref[l] = (float *)malloc((size_t)M2 * N * sizeof(float));
  }
  float *scratch = dev_alloc(TS * TS);
  for (int l = 0; l < LAYERS; ++l) {
    w[l] = dev_alloc((size_t)K * N);
    fill(w[l], (size_t)K * N, 177u + l, 4);
  }
  fill(act[0], (size_t)M2 * K, 15u, 2);
  memcpy(ref[0], act[0], (size_t)M2 * K * sizeof(float));
  for (int l = 0; l < LAYERS; ++l) {
    serial_layer(ref[l], w[l], ref[l + 1]);
    serial_layer(ref[l] + (size_t)M * K, w[l], ref[l + 1] + (size_t)M * N);
  }
  int bad = 0;
  pthread_barrier_t bar;
  pthread_barrier_init(&bar, nullptr, NT);
  for (int rep = 0; rep < reps; ++rep) {
    for (int l = 1; l <= LAYERS; ++l) memset(act[l], 0xff, (size_t)M2 * N * sizeof(float));
    std::vector<std::thread> th;
    for (int tid = 0; tid < NT; ++tid)
      th.emplace_back([&, tid] {
        const int64_t hf = xsmm_fused_brgemm_dispatch(XSMM_DTYPE_F32, TS, TS, KB, K, N, N, KB, (int64_t)KB * N, XSMM_GEMM_FLAG_BETA_0, 0,
                                                      XSMM_UNARY_RELU, 0, XSMM_BINARY_NONE);
        const int64_t hz = xsmm_unary_dispatch(XSMM_UNARY_ZERO, XSMM_DTYPE_F32, TS, TS, TS, TS, 0);
        const int tiles_n = N / TS, tiles = (M2 / TS) * tiles_n, per = tiles / NT;
        for (int l = 0; l < LAYERS; ++l) {
          for (int t = tid * per; t < (tid + 1) * per; ++t) { // static schedule: a contiguous share per thread
            const int i = t / tiles_n, j = t % tiles_n;
            xsmm_fused_brgemm_invoke(XSMM_DTYPE_F32, hf, act[l], (int64_t)i * TS * K, w[l], j * TS, act[l + 1], (int64_t)i * TS * N + j * TS,
                                     nullptr, 0, K / KB);
            if ((rep & 1) && tid == 3 && l == 1 && t == tid * per) xsmm_unary_invoke(XSMM_DTYPE_F32, hz, scratch, 0, scratch, 0);
          }
          pthread_barrier_wait(&bar);
        }
        if (tid == NT - 1) xsmm_hip_synchronize();
      });
    for (auto &t : th) t.join();
    for (int l = 1; l <= LAYERS; ++l)
      if (memcmp(act[l], ref[l], (size_t)M2 * N * sizeof(float))) ++bad;
  }
  pthread_barrier_destroy(&bar);
  printf("%-46s reps %d: %s\n", what, reps, bad ? "MISMATCH" : "identical to the serial run");
  for (int l = 0; l <= LAYERS; ++l) {
    hipFree(act[l]);
    free(ref[l]);
  }
  for (int l = 0; l < LAYERS; ++l) hipFree(w[l]);
  hipFree(scratch);
  return bad;
}

// SOLO -> MULTI. The first thread to use the queue runs its replayed groups with plain stores (DirectWindow, SOLO: no locked
// instruction on the caller's side); the first time another thread touches the queue state the process switches, once, to the
// two-sided protocol (membarrier + wait for the solo section in flight). Here: the main thread replays the fused tiles alone, and
// half way through a second thread starts flushing the queue and queueing its own tiles while the main thread keeps going.
// Must be the FIRST use of the tile queue in the process (afterwards the process is multi for good).
static int run_solo_handover(int reps, const char *what) {
  std::vector<float *> act(LAYERS + 1), w(LAYERS), ref(LAYERS + 1);
  for (int l = 0; l <= LAYERS; ++l) {
    act[l] = dev_alloc((size_t)M * N);
    ref[l] = (float *)malloc((size_t)M * N * sizeof(float));
  }
  float *scratch = dev_alloc((size_t)8 * TS * TS);
  for (int l = 0; l < LAYERS; ++l) {
    w[l] = dev_alloc((size_t)K * N);
    fill(w[l], (size_t)K * N, 277u + l, 4);
  }
  fill(act[0], (size_t)M * K, 25u, 2);
  memcpy(ref[0], act[0], (size_t)M * K * sizeof(float));
  for (int l = 0; l < LAYERS; ++l) serial_layer(ref[l], w[l], ref[l + 1]);
  const int64_t hf = xsmm_fused_brgemm_dispatch(XSMM_DTYPE_F32, TS, TS, KB, K, N, N, KB, (int64_t)KB * N, XSMM_GEMM_FLAG_BETA_0, 0,
                                                XSMM_UNARY_RELU, 0, XSMM_BINARY_NONE);
  const int64_t hz = xsmm_unary_dispatch(XSMM_UNARY_ZERO, XSMM_DTYPE_F32, TS, TS, TS, TS, 0);
  int bad = 0;
  std::atomic<int> stop{0};
  std::thread other;
  const int tiles_n = N / TS, tiles = (M / TS) * tiles_n;
  for (int rep = 0; rep < reps; ++rep) {
    // the second thread is started from INSIDE the one caller's tile loop: its first invoke - the process's one
    // solo -> two-sided switch - lands while the owner is marking members of a replayed group through the solo window
    auto start_other = [&] {
      other = std::thread([&] {
        int k = 0; // (bounded: thousands of locked arrivals from two threads would hand the queue to the scheduler thread)
        while (!stop.load(std::memory_order_acquire) && k < 600) {
          xsmm_unary_invoke(XSMM_DTYPE_F32, hz, scratch, (int64_t)(k & 7) * TS * TS, scratch, (int64_t)(k & 7) * TS * TS);
          if ((++k & 15) == 0) xsmm_hip_flush();
          usleep(20);
        }
      });
    };
    for (int l = 1; l <= LAYERS; ++l) memset(act[l], 0xff, (size_t)M * N * sizeof(float));
    for (int l = 0; l < LAYERS; ++l)
      for (int t = 0; t < tiles; ++t) {
        if (rep == reps / 2 && l == 1 && t == tiles / 2) start_other();
        const int i = t / tiles_n, j = t % tiles_n;
        xsmm_fused_brgemm_invoke(XSMM_DTYPE_F32, hf, act[l], (int64_t)i * TS * K, w[l], j * TS, act[l + 1], (int64_t)i * TS * N + j * TS,
                                 nullptr, 0, K / KB);
      }
    xsmm_hip_synchronize();
    for (int l = 1; l <= LAYERS; ++l)
      if (memcmp(act[l], ref[l], (size_t)M * N * sizeof(float))) ++bad;
  }
  stop.store(1, std::memory_order_release);
  other.join();
  xsmm_hip_synchronize();
  printf("%-46s reps %d: %s\n", what, reps, bad ? "MISMATCH" : "identical to the serial run");
  for (int l = 0; l <= LAYERS; ++l) {
    hipFree(act[l]);
    free(ref[l]);
  }
  for (int l = 0; l < LAYERS; ++l) hipFree(w[l]);
  hipFree(scratch);
  return bad;
}

// A chain of dependent in-place ops on ONE tile, each step issued by a different thread, handed over through an atomic
// (release / acquire) only: x = 0; then alternately x = 2 x and x = x + 1, i.e. after 2 n steps x = 2^n - 1 ... only if the
// scheduler takes the steps in the order the hand-overs define (the ops do not commute).
static int run_chain(int rounds, const char *what) {
  const int T = 16, STEPS = 40; // 20 doublings: values stay exact in f32
  float *x = dev_alloc(T * T), *two = dev_alloc(T * T), *one = dev_alloc(T * T);
  for (int i = 0; i < T * T; ++i) two[i] = 2.0f, one[i] = 1.0f;
  const int64_t hz = xsmm_unary_dispatch(XSMM_UNARY_ZERO, XSMM_DTYPE_F32, T, T, T, T, 0);
  const int64_t hm = xsmm_binary_dispatch(XSMM_BINARY_MUL, XSMM_DTYPE_F32, T, T, T, T, T, 0);
  const int64_t ha = xsmm_binary_dispatch(XSMM_BINARY_ADD, XSMM_DTYPE_F32, T, T, T, T, T, 0);
  std::atomic<int> turn{0};
  std::atomic<int> bad{0};
  std::vector<std::thread> th;
  for (int tid = 0; tid < NT; ++tid)
    th.emplace_back([&, tid] {
      for (int r = 0; r < rounds; ++r)
        for (int st = 0; st <= STEPS + 1; ++st) {
          const int my = r * (STEPS + 2) + st;
          if (my % NT != tid) continue;
          while (turn.load(std::memory_order_acquire) != my) {
          }
          if (st == 0) xsmm_unary_invoke(XSMM_DTYPE_F32, hz, x, 0, x, 0);
          else if (st == STEPS + 1) {
            xsmm_hip_synchronize();
            float want = 0.0f;
            for (int k = 1; k <= STEPS; ++k) want = (k & 1) ? want * 2.0f : want + 1.0f;
            for (int i = 0; i < T * T; ++i)
              if (x[i] != want) {
                bad.fetch_add(1);
                break;
              }
          } else if (st & 1) xsmm_binary_invoke(XSMM_DTYPE_F32, hm, x, 0, two, 0, x, 0);
          else xsmm_binary_invoke(XSMM_DTYPE_F32, ha, x, 0, one, 0, x, 0);
          turn.store(my + 1, std::memory_order_release);
        }
    });
  for (auto &t : th) t.join();
  printf("%-46s rounds %d: %s\n", what, rounds, bad.load() ? "MISMATCH" : "identical to the serial run");
  hipFree(x);
  hipFree(two);
  hipFree(one);
  return bad.load();
}

// Transposes folded into the gemm they feed (runtime.cpp, deferred transposes): ONE thread runs transpose -> temporary -> gemm per
// tile (the lowering of benchmarks/mlir/fp32-query-times-key.mlir, one temporary for all tiles) while three other threads keep
// invoking unrelated copies - every one of their invokes launches the remembered transpose if there is one, under the first
// thread's feet when it touches the record's destination or source (here: never; the flushes launch it) - and one of them flushes. In
// the second half of the repetitions a second thread runs the same script on its own temporary: one record per thread, both fold.
static int run_transposes(int reps, const char *what) {
  const int TILES = 48, T = 32, KQ = 64, LD = 512;
  float *Q = dev_alloc((size_t)T * LD), *Km = dev_alloc((size_t)T * LD), *out = dev_alloc((size_t)2 * TILES * T * T), *tmp = dev_alloc((size_t)2 * KQ * T);
  float *other_src = dev_alloc((size_t)8 * T * T), *other_dst = dev_alloc((size_t)8 * T * T);
  std::vector<float> ref((size_t)TILES * T * T);
  fill(Q, (size_t)T * LD, 91u, 2);
  fill(Km, (size_t)T * LD, 92u, 2);
  fill(other_src, (size_t)8 * T * T, 93u, 2);
  // tile t: columns 8 (t % 56) .. + 63 of Q and K (overlapping windows of one 32 x 512 matrix); out[t] = K_t Q_t^T
  for (int t = 0; t < TILES; ++t)
    for (int i = 0; i < T; ++i)
      for (int j = 0; j < T; ++j) {
        float acc = 0.0f;
        for (int kk = 0; kk < KQ; ++kk) acc += Km[i * LD + 8 * (t % 56) + kk] * Q[j * LD + 8 * (t % 56) + kk];
        ref[((size_t)t * T + i) * T + j] = acc;
      }
  const int64_t ht = xsmm_unary_dispatch(XSMM_UNARY_TRANSPOSE, XSMM_DTYPE_F32, T, KQ, LD, T, 0);
  const int64_t hg = xsmm_gemm_dispatch(XSMM_DTYPE_F32, T, T, KQ, LD, T, T, XSMM_GEMM_FLAG_BETA_0);
  const int64_t hc = xsmm_unary_dispatch(XSMM_UNARY_IDENTITY, XSMM_DTYPE_F32, T, T, T, T, 0);
  int bad = 0;
  int64_t st0[3], st1[3];
  xsmm_hip_fold_transpose_stats(st0);
  for (int rep = 0; rep < reps; ++rep) {
    const bool two = rep >= reps / 2;
    memset(out, 0xff, (size_t)2 * TILES * T * T * sizeof(float));
    std::atomic<int> done{0};
    std::vector<std::thread> th;
    auto qk = [&](int which) {
      for (int t = 0; t < TILES; ++t) {
        xsmm_unary_invoke(XSMM_DTYPE_F32, ht, Q, 8 * (t % 56), tmp, (int64_t)which * KQ * T);
        xsmm_gemm_invoke(XSMM_DTYPE_F32, hg, Km, 8 * (t % 56), tmp, (int64_t)which * KQ * T, out, ((int64_t)which * TILES + t) * T * T);
      }
      done.fetch_add(1, std::memory_order_release);
    };
    th.emplace_back(qk, 0);
    if (two) th.emplace_back(qk, 1);
    for (int o = 0; o < 3; ++o)
      th.emplace_back([&, o] {
        int k = 0;
        while (done.load(std::memory_order_acquire) < (two ? 2 : 1) && k < 4000) {
          xsmm_unary_invoke(XSMM_DTYPE_F32, hc, other_src, (int64_t)((o + k) & 7) * T * T, other_dst, (int64_t)o * T * T);
          if (o == 2 && (++k & 7) == 0) xsmm_hip_flush();
          else ++k;
          usleep(5);
        }
      });
    for (auto &t : th) t.join();
    xsmm_hip_synchronize();
    if (memcmp(out, ref.data(), ref.size() * sizeof(float))) ++bad;
    if (two && memcmp(out + (size_t)TILES * T * T, ref.data(), ref.size() * sizeof(float))) ++bad;
    // the temporary holds the last transpose
    for (int i = 0; i < T && !bad; ++i)
      for (int kk = 0; kk < KQ; ++kk)
        if (tmp[kk * T + i] != Q[i * LD + 8 * ((TILES - 1) % 56) + kk]) { ++bad; break; }
  }
  xsmm_hip_fold_transpose_stats(st1);
  printf("%-46s reps %d: %s (%ld gemms served from a transpose's source, %ld transposes dropped, %ld launched late)\n", what, reps,
         bad ? "MISMATCH" : "identical to the serial run", (long)(st1[0] - st0[0]), (long)(st1[1] - st0[1]), (long)(st1[2] - st0[2]));
  hipFree(Q); hipFree(Km); hipFree(out); hipFree(tmp); hipFree(other_src); hipFree(other_dst);
  return bad;
}

int main(int argc, char **argv) {
  int bad = 0;
  const int chain_rounds = argc > 1 ? atoi(argv[1]) : 200;
  // 1. synchronous mode, device operands: every invoke launches on the caller's thread
  xsmm_hip_set_async(0);
  xsmm_hip_set_tile_queue(0);
  bad += run_mlp(true, 2, "sync, device operands, 8 callers");
  // 2. host operands: each invoke mirrors its operands and copies ITS tile back (neighbours belong to other threads)
  bad += run_mlp(false, 2, "sync, host operands (mirror), 8 callers");
  // 3a. the tile queue, default mode: the bookkeeping under the queue's lock, replayed groups through the direct window - no
  //     scheduler thread for a program that repeats itself
  xsmm_hip_set_async(1);
  xsmm_hip_set_tile_queue(1);
  const int threads_before = thread_count();
  int64_t qs0[5], qs1[5];
  bad += run_solo_handover(12, "tile queue, one caller, then a second thread");
  xsmm_hip_tile_queue_stats(qs0);
  bad += run_fused(12, "tile queue, fused tiles replayed, 8 callers");
  xsmm_hip_tile_queue_stats(qs1);
  printf("direct window: %ld invokes replayed, %ld with full bookkeeping, %ld replays abandoned, %d threads\n", (long)(qs1[2] - qs0[2]),
         (long)(qs1[1] - qs0[1]), (long)(qs1[4] - qs0[4]), thread_count());
  if (qs1[2] - qs0[2] < 2 * 96 || qs1[4] - qs0[4] < 1 || thread_count() != threads_before) { // (how much is replayed depends on the interleaving: an abandoned replay backs the cache off for a growing number of groups)
    printf("direct window: UNEXPECTED\n");
    ++bad;
  }
  bad += run_mlp(true, 2, "tile queue, zero/brgemm/relu tiles, 8 callers");
  bad += run_transposes(8, "tile queue, transposes folded into gemms");
  // 3a'. the launch thread (rt_launcher.h): complete replays leave through it, everything else waits for it; off = the closing
  //      thread launches; it leaves after ~2 s without a hand-over and the next one starts a successor
  {
    int64_t lt0[2], lt1[2], lt2[2], lt3[2];
    xsmm_hip_launch_thread_stats(lt0);
    bad += run_fused(8, "launch thread on, fused tiles, 8 callers");
    xsmm_hip_launch_thread_stats(lt1);
    const int prev = xsmm_hip_set_launch_thread(0);
    bad += run_fused(8, "launch thread off, fused tiles, 8 callers");
    xsmm_hip_launch_thread_stats(lt2);
    xsmm_hip_set_launch_thread(1);
    int waited_lt = 0;
    while (launch_thread_exists() && waited_lt < 100) {
      usleep(100000);
      ++waited_lt;
    }
    const int gone = !launch_thread_exists();
    bad += run_solo_handover(12, "tile queue after the launch thread left");
    xsmm_hip_launch_thread_stats(lt3);
    printf("launch thread: %ld + %ld launches handed over, %ld while off, gone after %.1f s idle: %d, back: %d\n", (long)(lt1[0] - lt0[0]),
           (long)(lt3[0] - lt2[0]), (long)(lt2[0] - lt1[0]), waited_lt * 0.1, gone, (int)lt3[1]);
    if (prev != 1 || lt1[0] - lt0[0] < 1 || lt2[0] != lt1[0] || !gone || lt3[0] - lt2[0] < 1) {
      printf("launch thread: UNEXPECTED\n");
      ++bad;
    }
  }
  // 3b. mode 2: several callers hand their invokes to the ring + scheduler thread
  xsmm_hip_set_tile_queue(2);
  bad += run_mlp(true, 6, "tile queue, device operands, 8 callers");
  const int threads_busy = thread_count();
  bad += run_mlp(false, 2, "tile queue on, host operands, 8 callers");
  bad += run_chain(chain_rounds, "tile queue, 41-step dependent chain over 8 threads");
  // 4. the scheduler thread leaves after ~2-5 s without traffic, and the next push starts a new one
  int waited = 0;
  while (thread_count() >= threads_busy && threads_busy > threads_before && waited < 300) {
    usleep(100000);
    ++waited;
  }
  const int threads_idle = thread_count();
  printf("threads: %d before the queue, %d with the scheduler, %d after %.1f s idle\n", threads_before, threads_busy,
         threads_idle, waited * 0.1);
  if (threads_busy != threads_before + 1 || threads_idle != threads_before) {
    printf("scheduler thread life cycle: UNEXPECTED\n");
    ++bad;
  }
  bad += run_mlp(true, 3, "tile queue after the worker restarted");
  if (thread_count() != threads_before + 1) {
    printf("scheduler thread did not come back\n");
    ++bad;
  }
  // 5. mode switches between bursts
  xsmm_hip_set_async(0);
  bad += run_mlp(true, 1, "back to sync");
  xsmm_hip_set_async(1);
  bad += run_mlp(true, 2, "async again");
  bad += run_fused(4, "fused tiles through the scheduler thread");
  bad += run_transposes(4, "transposes + gemms through the scheduler thread");
  xsmm_hip_set_tile_queue(0);
  bad += run_mlp(true, 1, "async, queue off");
  printf("%s\n", bad ? "FAILED" : "OK");
  return bad ? 1 : 0;
}
