// tpp_replay - replays the dispatch/invoke call sequence that tpp-run's JIT'd code issues for
// the reference's mlir-gen kernels, against libtpp_xsmm_runner_utils.so, with the reference's
// timing definition (lib/TPP/Runner/TppRunnerWrapper.cpp:115-130: warm-up max(1, min(50, N/100))
// iterations, then ONE perf_start_timer/perf_stop_timer pair around N back-to-back kernel calls,
// mean = delta / N) and its metric (benchmarks/harness/controller.py:187-192:
// gflops = BENCH_TOTAL_FLOPS / mean / 1e9, FLOPs per tools/mlir-gen/MLIRGen.cpp:313-334).
//
// It is the stand-in for tpp-run (which needs MLIR) on the hot path only: buffers live in HBM
// (hipMalloc), filled like `--init-type const`; the calls are exactly the wire tuples of
// test/Passes/pass-convert-mlp-to-parallel-tile.mlir:80-88 (packed 32x32x32 tiles) or one
// whole-layer dispatch per layer.
//
//   tpp_replay --batch 256 --layers 1024,1024,1024,1024 --tiles 32 [--bias --relu] [--queue 1] [-n 100]
//   tpp_replay --batch 256 --layers 1024,1024,1024,1024 --whole-layer ...
//   tpp_replay --c1 [--queue 0|1]     BASELINE config 1: mlir-gen --kernel=args --batch=256 --layers=256,256
//                                     after the default pipeline = 192 block relayout invokes (xsmm.unary
//                                     identity, ld 256 -> 32), 64 brgemm invokes (32^3, br 8), 64 un-pack invokes
#include "../include/tpp_xsmm_abi.h"
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <functional>
#include <omp.h>
#include <string>
#include <vector>

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } \
  } while (0)

static std::vector<int64_t> parse_list(const char *s) {
  std::vector<int64_t> v;
  for (char *p = strdup(s), *t = strtok(p, ","); t; t = strtok(nullptr, ",")) v.push_back(atoll(t));
  return v;
}

static int run_case(int argc, char **argv);
static int run_script(const std::string &name, int queue, int64_t n_iter, int threads, int fold);

// --cases FILE: one case per line (the flags of a single run), all in this process - the shape sweeps of tools/refbench.py
// (benchmarks/config/matmul/*.json, fc/*.json) would otherwise pay a process start + HIP initialisation per row
int main(int argc, char **argv) {
  if (argc == 3 && std::string(argv[1]) == "--cases") {
    FILE *f = fopen(argv[2], "r");
    if (!f) { fprintf(stderr, "tpp_replay: cannot open %s\n", argv[2]); return 2; }
    char line[1024];
    int rc = 0;
    while (fgets(line, sizeof line, f)) {
      std::vector<char *> av{argv[0]};
      for (char *t = strtok(line, " \t\r\n"); t; t = strtok(nullptr, " \t\r\n")) av.push_back(t);
      if (av.size() < 2 || av[1][0] == '#') continue;
      fprintf(stderr, "tpp_replay: case:");
      for (size_t i = 1; i < av.size(); ++i) fprintf(stderr, " %s", av[i]);
      fprintf(stderr, "\n");
      rc |= run_case((int)av.size(), av.data());
    }
    fclose(f);
    return rc;
  }
  return run_case(argc, argv);
}

static int run_case(int argc, char **argv) {
  int64_t batch = 256, tile = 32, tile_n = 0, tile_k = 0, n_iter = 100;
  bool kernel_args = false; // mlir-gen --kernel=args: the output is an argument, the matmul accumulates into it (no BETA_0)
  int vnni = 2, split = -1, variant = -1, repeats = 1;
  int64_t block_pad = 0; // --block-pad P (experiments): P elements between consecutive packed blocks of A and of W (the block strides stop being powers of two)
  std::vector<int64_t> layers = {1024, 1024, 1024, 1024};
  bool bias = false, relu = false, whole = false, chain = false, print = false, c1 = false, rnd = false, bf16 = false, host_buffers = false;
  int queue = 1, threads = 1;
  std::string script;
  int fold = 1;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&]() { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
    if (a == "--batch") batch = atoll(next());
    else if (a == "--layers") layers = parse_list(next());
    else if (a == "--tiles") { // m[,n,k] like mlir-gen --tiles=64,48,64
      std::vector<int64_t> t = parse_list(next());
      tile = t[0];
      if (t.size() == 3) { tile_n = t[1]; tile_k = t[2]; }
    }
    else if (a == "-n") n_iter = atoll(next());
    else if (a == "--queue") queue = atoi(next());
    else if (a == "--threads") threads = atoi(next()); // the tile loop of a layer is an scf.parallel (OpenMP) in the reference
    else if (a == "--bias") bias = true;
    else if (a == "--relu") relu = true;
    else if (a == "--whole-layer") whole = true;
    else if (a == "--chain") whole = chain = true; // the whole-layer calls of an iteration handed over together (xsmm_hip_fused_brgemm_chain_invoke)
    else if (a == "--print") print = true;
    else if (a == "--c1") c1 = true;
    else if (a == "--script") script = next(); // the call scripts of benchmarks/mlir/*.mlir (base/mha.json, base/pack.json): see run_script
    else if (a == "--fold") fold = atoi(next()); // --script: xsmm_hip_set_fold_transpose (1 = default: transposes folded into the gemm they feed)
    else if (a == "--bf16") bf16 = true; // mlir-gen --float-type=bf16 --vnni=2: bf16 storage, W in VNNI-2 blocks
    else if (a == "--vnni") vnni = atoi(next()); // --vnni=4 (benchmarks/config/*: the *_dp4_* rows): W in [K/4][N][4] blocks
    else if (a == "--kernel") kernel_args = std::string(next()) == "args"; // const (default): zero fill folded into BETA_0; args: C += ...
    else if (a == "--split") split = atoi(next());     // xsmm_hip_force_split for this case (-1: the runtime's model)
    else if (a == "--variant") variant = atoi(next()); // xsmm_hip_force_variant at dispatch (-1: the runtime's choice)
    else if (a == "--block-pad") block_pad = atoll(next());
    else if (a == "--repeats") repeats = atoi(next()); // the timed loop R times (own timer each): min / median / max + the queue's abandon counter per case
    else if (a == "--host-buffers") host_buffers = true; // what an UNMODIFIED tpp-run hands over: plain malloc'ed host memory, modes from the environment only
    else if (a == "--random") rnd = true; // uniform [-1, 1) * fill instead of constant fills (switching power)
    else { fprintf(stderr, "unknown flag %s\n", a.c_str()); return 2; }
  }
  if (!host_buffers && xsmm_hip_device_count() < 1) { fprintf(stderr, "tpp_replay: no HIP device (there is no CPU fallback)\n"); return 1; }
  if (!script.empty()) return run_script(script, queue, n_iter, threads, fold);
  if (c1) {
    // A, W, C: 256x256 f32 filled 1.0; packed copies [8][8][32][32]; result 257 everywhere (C += A W)
    const int64_t N = 256, T = 32, NB = N / T, tt = T * T;
    float *buf[6];
    std::vector<float> ones((size_t)N * N, 1.0f);
    for (int i = 0; i < 6; ++i) {
      CHECK(hipMalloc((void **)&buf[i], N * N * sizeof(float)));
      CHECK(hipMemcpy(buf[i], ones.data(), N * N * sizeof(float), hipMemcpyHostToDevice));
    }
    float *A = buf[0], *W = buf[1], *C = buf[2], *Ap = buf[3], *Wp = buf[4], *Cp = buf[5];
    xsmm_hip_set_async(1);
    xsmm_hip_set_tile_queue(queue);
    const int64_t pack = xsmm_unary_dispatch(XSMM_UNARY_IDENTITY, 1, T, T, N, T, 0);
    const int64_t unpack = xsmm_unary_dispatch(XSMM_UNARY_IDENTITY, 1, T, T, T, N, 0);
    const int64_t hb = xsmm_brgemm_dispatch(1, T, T, T, T, T, T, tt, tt, 0);
    auto kernel = [&]() {
      for (int64_t bi = 0; bi < NB; ++bi)
        for (int64_t bj = 0; bj < NB; ++bj) {
          const int64_t blk = (bi * NB + bj) * tt;
          xsmm_unary_invoke(1, pack, A, bi * T * N + bj * T, Ap, blk);
          xsmm_unary_invoke(1, pack, W, bj * T * N + bi * T, Wp, blk); // W blocks stored [NB][KB]
          xsmm_unary_invoke(1, pack, C, bi * T * N + bj * T, Cp, blk);
        }
      for (int64_t bi = 0; bi < NB; ++bi)
        for (int64_t bj = 0; bj < NB; ++bj)
          xsmm_brgemm_invoke(1, hb, Ap, bi * NB * tt, Wp, bj * NB * tt, Cp, (bi * NB + bj) * tt, NB);
      for (int64_t bi = 0; bi < NB; ++bi)
        for (int64_t bj = 0; bj < NB; ++bj)
          xsmm_unary_invoke(1, unpack, Cp, (bi * NB + bj) * tt, C, bi * T * N + bj * T);
    };
    kernel();
    xsmm_hip_synchronize();
    std::vector<float> h((size_t)N * N);
    CHECK(hipMemcpy(h.data(), C, N * N * sizeof(float), hipMemcpyDeviceToHost));
    for (float v : h)
      if (v != 257.0f) { fprintf(stderr, "tpp_replay --c1: expected 257 everywhere, got %g\n", v); return 1; }
    const int64_t t0 = perf_start_timer();
    for (int64_t i = 0; i < n_iter; ++i) kernel();
    const double mean = perf_stop_timer(t0) / (double)n_iter;
    const double fl = 2.0 * N * N * N; // 33,554,432 (BENCH_TOTAL_FLOPS of the 256^3 matmul)
    printf("%g\n", mean);
    fprintf(stderr, "tpp_replay: C1 call script (320 invokes), queue %d: mean %.3f us, %.1f GFLOP/s, first result 257 checked\n",
            queue, mean * 1e6, fl / mean / 1e9);
    return 0;
  }
  const int L = (int)layers.size() - 1;
  // mlir-gen --kernel=const: zero fill folded into BETA_0; --kernel=args: the output tensor is a function argument the contraction
  // accumulates into (MLIRGen.cpp:249-253; test/Integration/mlir-gen.mlir:17,28: 10 * 1 + 1 = 11)
  int64_t gflags = kernel_args ? 0 : XSMM_GEMM_FLAG_BETA_0;
  const int64_t ukind = relu ? XSMM_UNARY_RELU : XSMM_UNARY_NONE;
  const int64_t bkind = bias ? XSMM_BINARY_ADD : XSMM_BINARY_NONE, bflags = bias ? XSMM_BINARY_FLAG_BCAST_COL_IN_0 : 0;
  double flops = 0;
  for (int l = 0; l < L; ++l) flops += 2.0 * batch * layers[l] * layers[l + 1] + (bias ? batch * layers[l + 1] : 0) + (relu ? batch * layers[l + 1] : 0);

  // buffers: activations [l] (batch x layers[l]), weights, biases; const 1.0 / (1/K) / 0.5 fills. With
  // constant fills the packed / VNNI-2 / flat layouts of a tensor are the same bytes.
  const int64_t dt = bf16 ? 2 : 1;
  const size_t es = bf16 ? 2 : 4;
  auto to_bf16 = [](float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); };
  std::vector<void *> act(L + 1), W(L), B(L);
  auto dalloc = [&](size_t n, float v) {
    std::vector<float> h(n, v);
    if (rnd) for (auto &x : h) x = v * (float)(2.0 * rand() / (double)RAND_MAX - 1.0);
    if (host_buffers) {
      // --host-buffers: memref.alloc as the LLVM lowering emits it - malloc(bytes + alignment), the pointer rounded up to 64 bytes (and a
      // memref.global sits 128-byte aligned in the JIT's data section: BuilderUtils.cpp:103) - never page aligned, never registered with HIP
      char *raw = (char *)malloc(n * es + 64);
      void *d = (void *)(((uintptr_t)raw + 63) & ~(uintptr_t)63);
      if (bf16) for (size_t i = 0; i < n; ++i) ((uint16_t *)d)[i] = to_bf16(h[i]);
      else memcpy(d, h.data(), n * 4);
      return d;
    }
    void *d; CHECK(hipMalloc(&d, n * es));
    if (bf16) {
      std::vector<uint16_t> hb(n);
      for (size_t i = 0; i < n; ++i) hb[i] = to_bf16(h[i]);
      CHECK(hipMemcpy(d, hb.data(), n * 2, hipMemcpyHostToDevice));
    } else {
      CHECK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    }
    return d;
  };
  const size_t padx = block_pad > 0 ? 2 : 1; // (padded blocks: room for the gaps - constant fills, so the gaps hold the same value)
  for (int l = 0; l <= L; ++l) act[l] = dalloc(padx * (size_t)batch * layers[l], 1.0f);
  for (int l = 0; l < L; ++l) { W[l] = dalloc(padx * (size_t)layers[l] * layers[l + 1], 1.0f / (float)layers[l]); B[l] = dalloc((size_t)layers[l + 1], 0.5f); }

  if (bf16) gflags |= XSMM_GEMM_WIRE_VNNI_B;
  // --host-buffers: NO xsmm_hip_* call before or inside the timed region - the program below is the reference's 13 + 2 symbols only,
  // the runtime's modes come from the environment (TPP_HIP_ASYNC / TPP_HIP_TILE_QUEUE / TPP_HIP_HOST_CACHE), like under an unmodified tpp-run
  const int old_vf = host_buffers ? 2 : xsmm_hip_set_vnni_factor(bf16 ? vnni : 2);
  if (!host_buffers) {
    xsmm_hip_force_split(split);
    xsmm_hip_force_variant(variant);
    xsmm_hip_set_async(1);
    xsmm_hip_set_tile_queue(queue);
  } else if ((bf16 && vnni != 2) || split != -1 || variant != -1 || chain) {
    fprintf(stderr, "tpp_replay: --host-buffers takes its settings from the environment only (no --vnni 4 / --split / --variant / --chain)\n");
    return 2;
  }
  std::vector<int64_t> handle(L);
  const int64_t tn = tile_n ? tile_n : tile, tk = tile_k ? tile_k : tile; // blocks: A [MB][KB][tm][tk], W [NB][KB][tk][tn], C [MB][NB][tm][tn]
  for (int l = 0; l < L; ++l) {
    const int64_t K = layers[l], N = layers[l + 1];
    if (whole) // one dispatch per layer on the flat row-major tensors
      handle[l] = xsmm_fused_brgemm_dispatch(dt, batch, N, 64, K, N, N, 64, 64 * N, gflags, 0, ukind, bflags, bkind);
    else       // packed tiles: [MB][KB][t][t] x [NB][KB][t][t] -> [MB][NB][t][t]
      handle[l] = xsmm_fused_brgemm_dispatch(dt, tile, tn, tk, tk, tn, tn, tile * tk + block_pad, tk * tn + block_pad, gflags, 0, ukind, bflags, bkind);
  }
  if (!host_buffers) xsmm_hip_force_variant(-1);
  int chained = -1;
  std::vector<void *> pa(L), pb(L), pc(L), pd(L);
  std::vector<int64_t> z(L, 0), br(L);
  for (int l = 0; l < L; ++l) pa[l] = act[l], pb[l] = W[l], pc[l] = act[l + 1], pd[l] = B[l], br[l] = layers[l] / 64;
  auto kernel = [&]() {
    if (chain) {
      chained = xsmm_hip_fused_brgemm_chain_invoke(dt, L, handle.data(), pa.data(), z.data(), pb.data(), z.data(), pc.data(), z.data(), pd.data(),
                                                   z.data(), br.data());
      return;
    }
    for (int l = 0; l < L; ++l) {
      const int64_t K = layers[l], N = layers[l + 1];
      if (whole) {
        xsmm_fused_brgemm_invoke(dt, handle[l], act[l], 0, W[l], 0, act[l + 1], 0, B[l], 0, K / 64);
      } else {
        const int64_t MB = batch / tile, NB = N / tn, KB = K / tk;
        auto run = [&](int64_t t0, int64_t t1) { // tiles [t0, t1) of the MB x NB grid, row-major (static schedule)
          for (int64_t t = t0; t < t1; ++t) {
            const int64_t i = t / NB, j = t % NB;
            xsmm_fused_brgemm_invoke(dt, handle[l], act[l], i * KB * (tile * tk + block_pad), W[l], j * KB * (tk * tn + block_pad), act[l + 1],
                                     (i * NB + j) * tile * tn, B[l], j * tn, KB);
          }
        };
        if (threads <= 1) { // the lowered scf.forall: two nested loops (no index arithmetic per tile)
          for (int64_t i = 0; i < MB; ++i)
            for (int64_t j = 0; j < NB; ++j)
              xsmm_fused_brgemm_invoke(dt, handle[l], act[l], i * KB * (tile * tk + block_pad), W[l], j * KB * (tk * tn + block_pad), act[l + 1],
                                       (i * NB + j) * tile * tn, B[l], j * tn, KB);
        }
        else { // the reference's scf.parallel -> OpenMP parallel-for over the tile grid (static schedule, barrier at the end)
          const int64_t total = MB * NB;
#pragma omp parallel for schedule(static) num_threads(threads)
          for (int64_t t = 0; t < total; ++t) run(t, t + 1);
        }
      }
    }
  };
  const int64_t warm = n_iter / 100 < 1 ? 1 : (n_iter / 100 > 50 ? 50 : n_iter / 100);
  for (int64_t i = 0; i < warm; ++i) kernel();
  if (host_buffers) (void)perf_stop_timer(perf_start_timer()); // (the ABI's own synchronisation point)
  else xsmm_hip_synchronize();
  const int64_t t0 = perf_start_timer();
  for (int64_t i = 0; i < n_iter; ++i) kernel();
  const double host_dt = (double)(perf_start_timer() - t0) * 1e-9; // all invokes returned (host side only)
  const double elapsed = perf_stop_timer(t0); // flushes the tile queue and drains the stream
  const double mean = elapsed / (double)n_iter;
  printf("%g\n", mean); // tpp-run prints the mean seconds (MLIRBench.cpp:297-300)
  // --repeats R: R - 1 more timed loops of the same N calls, each behind its own timer: a table row then says whether its mean is a
  // typical loop or one hiccup (a 51 us row among 4.2 us ones), and how many replays the tile queue abandoned
  std::vector<double> loops{mean * 1e6};
  int64_t q_before[5] = {0, 0, 0, 0, 0};
  xsmm_hip_tile_queue_stats(q_before);
  for (int r = 1; r < repeats; ++r) {
    const int64_t t1 = perf_start_timer();
    for (int64_t i = 0; i < n_iter; ++i) kernel();
    loops.push_back(perf_stop_timer(t1) / (double)n_iter * 1e6);
  }
  if (repeats > 1) {
    int64_t q_after[5];
    xsmm_hip_tile_queue_stats(q_after);
    std::vector<double> sl(loops);
    std::sort(sl.begin(), sl.end());
    fprintf(stderr, "tpp_replay: repeats %d x %ld calls: min %.3f median %.3f max %.3f us; replays abandoned in the repeats %ld\n", repeats, (long)n_iter,
            sl.front(), sl[sl.size() / 2], sl.back(), (long)(q_after[4] - q_before[4]));
  }
  fprintf(stderr, "tpp_replay: %s, batch %ld, %d layer(s), queue %d: mean %.3f us (host side of the invokes %.3f us), %.1f GFLOP/s (BENCH_TOTAL_FLOPS %.0f), kernel %s\n",
          chain ? (chained == 1 ? "whole-layer calls as ONE chain launch" : "whole-layer calls handed over together, run call by call") : whole ? "whole-layer dispatch" : "packed tile invokes", (long)batch, L, queue, mean * 1e6,
          host_dt / (double)n_iter * 1e6, flops / mean / 1e9, flops,
          (queue && !whole && xsmm_hip_last_grouped_kernel()[0]) ? xsmm_hip_last_grouped_kernel()
          : (whole && !chain && xsmm_hip_last_refined_kernel()[0])  ? xsmm_hip_last_refined_kernel()
                                                                    : xsmm_hip_kernel_name(handle[0]));
  if (queue) {
    int64_t qs[5];
    xsmm_hip_tile_queue_stats(qs);
    fprintf(stderr, "tpp_replay: tile queue: %ld grouped launches, %ld invokes with full bookkeeping, %ld replayed, %ld groups ended by a known terminator, %ld replays abandoned\n",
            (long)qs[0], (long)qs[1], (long)qs[2], (long)qs[3], (long)qs[4]);
  }
  if (host_buffers) {
    // the host reads its own output buffer, like tpp-run's print behind the timing loop (perf_stop_timer was the synchronisation
    // point). Constant fills: every layer is K * 1 * (1/K) (+ 0.5) from an input of that value - closed form, checked here.
    int64_t hs[10];
    xsmm_hip_host_cache_stats(hs);
    fprintf(stderr, "tpp_replay: host buffers (TPP_HIP_ASYNC=%s TPP_HIP_TILE_QUEUE=%s TPP_HIP_HOST_CACHE=%s): host cache %ld extents, %.1f MiB mirrored, %.1f MiB uploaded, "
                    "%.1f MiB written back, %ld invokes translated lock-free / %ld locked\n", getenv("TPP_HIP_ASYNC") ? getenv("TPP_HIP_ASYNC") : "-",
            getenv("TPP_HIP_TILE_QUEUE") ? getenv("TPP_HIP_TILE_QUEUE") : "-", getenv("TPP_HIP_HOST_CACHE") ? getenv("TPP_HIP_HOST_CACHE") : "-", (long)hs[0],
            hs[1] / 1048576.0, hs[2] / 1048576.0, hs[4] / 1048576.0, (long)hs[7], (long)hs[8]);
    if (!rnd && !kernel_args) {
      double v = 1.0;
      for (int l = 0; l < L; ++l) {
        v = v * 1.0 + (bias ? 0.5 : 0.0); // sum_k a * (1/K) = a
        if (relu && v < 0) v = 0;
        if (bf16) { float f = (float)v; uint32_t u = (uint32_t)to_bf16(f) << 16; memcpy(&f, &u, 4); v = f; }
      }
      const size_t n_out = (size_t)batch * layers[L];
      size_t bad = 0;
      for (size_t i = 0; i < n_out; ++i) {
        float got;
        if (bf16) { uint32_t u = (uint32_t)((uint16_t *)act[L])[i] << 16; memcpy(&got, &u, 4); }
        else got = ((float *)act[L])[i];
        bad += !(fabs((double)got - v) <= 1e-5 * fabs(v) + (bf16 ? 0.01 * fabs(v) : 0.0));
      }
      if (bad) { fprintf(stderr, "tpp_replay --host-buffers: WRONG RESULT in the host's output buffer (%zu of %zu values, expected %g)\n", bad, n_out, v); return 1; }
      fprintf(stderr, "tpp_replay: host output buffer checked (%zu values = %g)\n", n_out, v);
    }
    return 0; // (buffers live to the end of the process, like memref globals)
  }
  if (print) {
    std::vector<float> h(8);
    if (bf16) {
      uint16_t hb[8];
      CHECK(hipMemcpy(hb, act[L], 16, hipMemcpyDeviceToHost));
      for (int i = 0; i < 8; ++i) { uint32_t u = (uint32_t)hb[i] << 16; memcpy(&h[i], &u, 4); }
    } else {
      CHECK(hipMemcpy(h.data(), act[L], 8 * sizeof(float), hipMemcpyDeviceToHost));
    }
    printf("( %g, %g, %g, %g, %g, %g, %g, %g )\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  }
  xsmm_hip_set_vnni_factor(old_vf);
  xsmm_hip_force_split(-1);
  for (void *p : act) CHECK(hipFree(p));
  for (void *p : W) CHECK(hipFree(p));
  for (void *p : B) CHECK(hipFree(p));
  return 0;
}

// --script NAME: the xsmm call scripts of the reference's hand-written benchmark files (benchmarks/config/base/mha.json, pack.json ->
// benchmarks/mlir/*.mlir), f32. The calls per tile are the ones test/Conversion/LinalgToXsmm/linalg-to-gemm.mlir pins for exactly these
// functions (mha_projection :132-145, mha_query_times_key :46-62, mha_out_softmax_times_value :91-104), with the zero fill folded into
// BETA_0 as the default pipeline does (FoldXsmmFlags, lib/TPP/PassBundles/LinalgLowering.cpp:56; test/Passes/fold-xsmm-flags.mlir:3-20);
// the pack files are per-block 2-D copies (LowerPacksAndUnpacks.cpp:45-49,112-121 -> xsmm.unary identity). GFLOP/s uses each file's
// own BENCH_TOTAL_FLOPS line (for the pack files that number is the tensor's byte count).
//   mha_projection  fp32-projection.mlir            forall (64, 8): gemm [32,64,512,512,512,512] beta_0
//   mha_qk          fp32-query-times-key.mlir       forall (64, 8): unary transpose [32,64,512,32] into a 64x32 temporary (memref.alloc in
//                                                   the loop body: one per calling thread here), gemm [32,32,64,512,32,32] beta_0
//   mha_sv          fp32-out-softmax-times-value.mlir  forall (64, 8): gemm [32,64,32,32,512,512] beta_0
//   pack_a / pack_b / unpack_c   fp32-pack-gemm-operand-a-512x1024.mlir, -b-512x1024.mlir, fp32-unpack-gemm-operand-a-512x512.mlir
// Constant / index fills with closed-form results, checked on the first run (parity against the oracle: tests/test_mha_scripts_gpu.py).
static int run_script(const std::string &name, int queue, int64_t n_iter, int threads, int fold) {
  xsmm_hip_set_async(1);
  xsmm_hip_set_tile_queue(queue);
  xsmm_hip_set_fold_transpose(fold);
  auto dfill = [&](size_t n, auto gen) {
    float *d; CHECK(hipMalloc((void **)&d, n * 4));
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = gen(i);
    CHECK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
  };
  auto cst = [](float v) { return [v](size_t) { return v; }; };
  const int64_t Bt = 64, S = 32, H = 8, D = 64, E = H * D; // batch, sequence, heads, head size, embedding (512)
  std::vector<float *> bufs;
  std::function<void()> kernel;
  std::function<bool(const std::vector<float> &)> check;
  float *out = nullptr;
  size_t out_n = 0;
  double flops = 0;
  int invokes = 0;
  if (threads < 1) threads = 1;
  auto par2 = [threads](int64_t n0, int64_t n1, auto body) { // the lowered scf.forall (n0, n1): nested loops, or OpenMP over the flat grid
    if (threads <= 1) {
      for (int64_t i = 0; i < n0; ++i)
        for (int64_t j = 0; j < n1; ++j) body(i, j, 0);
    } else {
#pragma omp parallel for schedule(static) num_threads(threads)
      for (int64_t t = 0; t < n0 * n1; ++t) body(t / n1, t % n1, omp_get_thread_num());
    }
  };
  if (name == "mha_projection") {
    float *in = dfill(Bt * S * E, cst(1.0f)), *W = dfill(E * E, cst(0.5f));
    out = dfill(out_n = Bt * S * E, cst(-1.0f));
    bufs = {in, W, out};
    const int64_t h = xsmm_gemm_dispatch(1, S, D, E, E, E, E, XSMM_GEMM_FLAG_BETA_0);
    kernel = [=]() { par2(Bt, H, [=](int64_t b, int64_t hd, int) { xsmm_gemm_invoke(1, h, in, b * S * E, W, hd * D, out, b * S * E + hd * D); }); };
    check = [=](const std::vector<float> &r) { for (float v : r) if (v != 256.0f) return false; return true; };
    flops = 1073741824.0, invokes = 512;
  } else if (name == "mha_qk") {
    float *Q = dfill(Bt * S * E, cst(1.0f)), *K = dfill(Bt * S * E, cst(0.5f));
    out = dfill(out_n = Bt * H * S * S, cst(-1.0f));
    float *tmp = dfill((size_t)threads * D * S, cst(0.0f));
    bufs = {Q, K, out, tmp};
    const int64_t ht = xsmm_unary_dispatch(XSMM_UNARY_TRANSPOSE, 1, S, D, E, S, 0);
    const int64_t hg = xsmm_gemm_dispatch(1, S, S, D, E, S, S, XSMM_GEMM_FLAG_BETA_0);
    kernel = [=]() {
      par2(Bt, H, [=](int64_t b, int64_t hd, int th) {
        xsmm_unary_invoke(1, ht, Q, b * S * E + hd * D, tmp, th * D * S);
        xsmm_gemm_invoke(1, hg, K, b * S * E + hd * D, tmp, th * D * S, out, (b * H + hd) * S * S);
      });
    };
    check = [=](const std::vector<float> &r) { for (float v : r) if (v != 32.0f) return false; return true; };
    flops = 67108864.0, invokes = 1024;
  } else if (name == "mha_sv") {
    float *P = dfill(Bt * H * S * S, cst(1.0f)), *V = dfill(Bt * S * E, cst(0.5f));
    out = dfill(out_n = Bt * S * E, cst(-1.0f));
    bufs = {P, V, out};
    const int64_t h = xsmm_gemm_dispatch(1, S, D, S, S, E, E, XSMM_GEMM_FLAG_BETA_0);
    kernel = [=]() {
      par2(Bt, H, [=](int64_t b, int64_t hd, int) { xsmm_gemm_invoke(1, h, P, (b * H + hd) * S * S, V, b * S * E + hd * D, out, b * S * E + hd * D); });
    };
    check = [=](const std::vector<float> &r) { for (float v : r) if (v != 16.0f) return false; return true; };
    flops = 67108864.0, invokes = 512;
  } else if (name == "pack_a" || name == "pack_b" || name == "unpack_c") {
    // pack_a: [512][1024] -> [16][32][32][32];  pack_b: [1024][512] -> (outer_dims_perm [1,0]) [16 (n)][32 (k)][32][32];  unpack_c: [16][16][32][32] -> [512][512]
    const int64_t T = 32, R = name == "pack_b" ? 1024 : 512, Ccols = name == "pack_a" ? 1024 : 512;
    const int64_t RB = R / T, CB = Ccols / T, tt = T * T;
    float *in = dfill(R * Ccols, [](size_t i) { return (float)i; });
    out = dfill(out_n = R * Ccols, cst(-1.0f));
    bufs = {in, out};
    const bool un = name == "unpack_c", pb = name == "pack_b";
    const int64_t h = un ? xsmm_unary_dispatch(XSMM_UNARY_IDENTITY, 1, T, T, T, Ccols, 0) : xsmm_unary_dispatch(XSMM_UNARY_IDENTITY, 1, T, T, Ccols, T, 0);
    float *o = out;
    kernel = [=]() {
      if (pb) par2(CB, RB, [=](int64_t nb, int64_t kb, int) { xsmm_unary_invoke(1, h, in, kb * T * Ccols + nb * T, o, (nb * RB + kb) * tt); });
      else if (un) par2(RB, CB, [=](int64_t i, int64_t j, int) { xsmm_unary_invoke(1, h, in, (i * CB + j) * tt, o, i * T * Ccols + j * T); });
      else par2(RB, CB, [=](int64_t i, int64_t j, int) { xsmm_unary_invoke(1, h, in, i * T * Ccols + j * T, o, (i * CB + j) * tt); });
    };
    check = [=](const std::vector<float> &r) {
      for (int64_t i = 0; i < R; ++i)
        for (int64_t j = 0; j < Ccols; ++j) {
          const int64_t flat = i * Ccols + j, bi = i / T, bj = j / T, blk = ((pb ? bj * RB + bi : bi * CB + bj) * T + i % T) * T + j % T;
          if (un ? r[flat] != (float)blk : r[blk] != (float)flat) return false;
        }
      return true;
    };
    flops = (double)(R * Ccols * 4), invokes = (int)(RB * CB);
  } else {
    fprintf(stderr, "tpp_replay: unknown script %s\n", name.c_str());
    return 2;
  }
  kernel();
  xsmm_hip_synchronize();
  std::vector<float> res(out_n);
  CHECK(hipMemcpy(res.data(), out, out_n * 4, hipMemcpyDeviceToHost));
  const bool no_check = getenv("TPP_REPLAY_NO_CHECK") != nullptr; // the host-path rig (tools/host_path_rig.sh: this file against tests/tsan/fake_hip.cpp with FAKE_HIP_NO_COMPUTE=1)
  if (!no_check && !check(res)) { fprintf(stderr, "tpp_replay --script %s: WRONG RESULT\n", name.c_str()); return 1; }
  const int64_t warm = n_iter / 100 < 1 ? 1 : (n_iter / 100 > 50 ? 50 : n_iter / 100);
  for (int64_t i = 0; i < warm; ++i) kernel();
  xsmm_hip_synchronize();
  int64_t q0[5] = {0, 0, 0, 0, 0}, q1[5] = {0, 0, 0, 0, 0};
  if (queue) xsmm_hip_tile_queue_stats(q0);
  const int64_t t0 = perf_start_timer();
  for (int64_t i = 0; i < n_iter; ++i) kernel();
  const double host_dt = (double)(perf_start_timer() - t0) * 1e-9;
  const double mean = perf_stop_timer(t0) / (double)n_iter;
  if (queue) xsmm_hip_tile_queue_stats(q1);
  printf("%g\n", mean);
  fprintf(stderr, "tpp_replay: script %s (%d invokes per call, %d calling thread(s)), queue %d: mean %.3f us (host side of the invokes %.3f us), %.1f GFLOP/s (BENCH_TOTAL_FLOPS %.0f), kernel %s; result checked; %.1f launches per call (per call: %.1f invokes with full bookkeeping, %.1f replayed, %.2f replays abandoned)\n",
          name.c_str(), invokes, threads, queue, mean * 1e6, host_dt / (double)n_iter * 1e6, flops / mean / 1e9, flops,
          queue && xsmm_hip_last_grouped_kernel()[0] ? xsmm_hip_last_grouped_kernel() : "(one launch per invoke)",
          queue ? (double)(q1[0] - q0[0]) / (double)n_iter : (double)invokes, (double)(q1[1] - q0[1]) / (double)n_iter,
          (double)(q1[2] - q0[2]) / (double)n_iter, (double)(q1[4] - q0[4]) / (double)n_iter);
  // ... and the result once more behind the timed calls (every caller count, every replayed / merged / folded launch form)
  CHECK(hipMemset(out, 0xff, out_n * 4));
  kernel();
  xsmm_hip_synchronize();
  CHECK(hipMemcpy(res.data(), out, out_n * 4, hipMemcpyDeviceToHost));
  if (!no_check && !check(res)) { fprintf(stderr, "tpp_replay --script %s: WRONG RESULT after the timed calls\n", name.c_str()); return 1; }
  if (name == "mha_qk") {
    int64_t f[3];
    xsmm_hip_fold_transpose_stats(f);
    fprintf(stderr, "tpp_replay: transposes (process totals): %ld gemms served from a transpose's source, %ld transposes dropped as dead, %ld launched late; folding %s\n",
            (long)f[0], (long)f[1], (long)f[2], fold ? "on" : "off");
  }
  xsmm_hip_synchronize();
  for (float *p : bufs) CHECK(hipFree(p));
  xsmm_hip_set_fold_transpose(1);
  return 0;
}
