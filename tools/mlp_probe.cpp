// mlp_probe - a rank's share of the BASELINE config-4 MLP (3 layers 1024 -> 1024, bf16, bias + ReLU; mlir-gen's layer chain,
// MLIRGen.cpp:632-681) through the C-ABI, natively: `rows` rows of the batch as
//   (a) three whole-layer xsmm_fused_brgemm_invoke launches (what round 2 measured), and
//   (b) ONE xsmm_hip_fused_brgemm_chain_invoke (the persistent chain kernel when the runtime can run it as one launch),
// stream time per step by HIP events over `iters` back-to-back steps after a spin-up, plus each single layer alone.
// Prints ONE JSON line per row count. rocprofv3 target for the chain kernel's duration (--only chain|layers).
//   mlp_probe [--rows 512,1024,2048,4096] [--iters N] [--layers 3] [--width 1024] [--variant V] [--only chain|layers|all]
#include "../include/tpp_xsmm_abi.h"
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } \
  } while (0)

static unsigned short f2bf(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

int main(int argc, char **argv) {
  int iters = 400, layers = 3, width = 1024, variant = -1, pad = 0;
  std::string rows_arg = "512,1024,2048,4096", only = "all";
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "--iters" && i + 1 < argc) iters = atoi(argv[++i]);
    else if (a == "--rows" && i + 1 < argc) rows_arg = argv[++i];
    else if (a == "--layers" && i + 1 < argc) layers = atoi(argv[++i]);
    else if (a == "--width" && i + 1 < argc) width = atoi(argv[++i]);
    else if (a == "--variant" && i + 1 < argc) variant = atoi(argv[++i]);
    else if (a == "--only" && i + 1 < argc) only = argv[++i];
    else if (a == "--pad" && i + 1 < argc) pad = atoi(argv[++i]); // extra elements per row of the input / the activations (leading dimensions K + pad)
    else { fprintf(stderr, "unknown flag %s\n", a.c_str()); return 2; }
  }
  if (xsmm_hip_device_count() < 1) { fprintf(stderr, "mlp_probe: no HIP device (there is no CPU fallback)\n"); return 1; }
  if (layers < 1 || layers > 8 || width % 64) { fprintf(stderr, "layers 1..8, width a multiple of 64\n"); return 2; }
  std::vector<int> rows_list;
  for (size_t p = 0; p < rows_arg.size();) {
    size_t q = rows_arg.find(',', p);
    if (q == std::string::npos) q = rows_arg.size();
    rows_list.push_back(atoi(rows_arg.substr(p, q - p).c_str()));
    p = q + 1;
  }
  const int64_t N = width, K = width, br = K / 64;
  std::default_random_engine eng(7);
  std::uniform_real_distribution<float> du(-1.0f, 1.0f);
  // weights VNNI-2 [K/2][N][2], scaled so that activations stay O(1) through the layers
  std::vector<unsigned short *> W(layers), Bv(layers);
  {
    std::vector<unsigned short> h((size_t)K * N), hb(N);
    for (int l = 0; l < layers; ++l) {
      for (auto &v : h) v = f2bf(du(eng) * 0.06f);
      for (auto &v : hb) v = f2bf(du(eng) * 0.5f);
      CHECK(hipMalloc((void **)&W[l], h.size() * 2));
      CHECK(hipMalloc((void **)&Bv[l], hb.size() * 2));
      CHECK(hipMemcpy(W[l], h.data(), h.size() * 2, hipMemcpyHostToDevice));
      CHECK(hipMemcpy(Bv[l], hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    }
  }
  xsmm_hip_set_async(1);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rows : rows_list) {
    const int64_t LD = K + pad;
    std::vector<unsigned short> hx((size_t)rows * LD);
    for (auto &v : hx) v = f2bf(du(eng));
    unsigned short *x;
    std::vector<unsigned short *> act(layers);
    CHECK(hipMalloc((void **)&x, hx.size() * 2));
    CHECK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    for (int l = 0; l < layers; ++l) CHECK(hipMalloc((void **)&act[l], (size_t)rows * (N + pad) * 2));
    if (variant >= 0) xsmm_hip_force_variant(variant);
    const int64_t h = xsmm_fused_brgemm_dispatch(2, rows, N, 64, LD, N, N + pad, 64, 64 * N, XSMM_GEMM_FLAG_BETA_0 | XSMM_GEMM_WIRE_VNNI_B, 0,
                                                 XSMM_UNARY_RELU, XSMM_BINARY_FLAG_BCAST_COL_IN_0, XSMM_BINARY_ADD);
    xsmm_hip_force_variant(-1);
    std::vector<int64_t> hs(layers, h), zero(layers, 0), brs(layers, br);
    std::vector<void *> pa(layers), pb(layers), pc(layers), pd(layers);
    for (int l = 0; l < layers; ++l) {
      pa[l] = l ? (void *)act[l - 1] : (void *)x;
      pb[l] = W[l];
      pc[l] = act[l];
      pd[l] = Bv[l];
    }
    int fused = 0;
    auto step_layers = [&]() {
      for (int l = 0; l < layers; ++l) xsmm_fused_brgemm_invoke(2, h, pa[l], 0, pb[l], 0, pc[l], 0, pd[l], 0, br);
    };
    auto step_chain = [&]() {
      fused = xsmm_hip_fused_brgemm_chain_invoke(2, layers, hs.data(), pa.data(), zero.data(), pb.data(), zero.data(), pc.data(), zero.data(),
                                                 pd.data(), zero.data(), brs.data());
    };
    auto step_one = [&]() { xsmm_fused_brgemm_invoke(2, h, pa[0], 0, pb[0], 0, pc[0], 0, pd[0], 0, br); };
    auto time_it = [&](auto &&fn) {
      const auto t_up = std::chrono::steady_clock::now();
      while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_up).count() < 0.05) { // clocks up
        for (int i = 0; i < 100; ++i) fn();
        CHECK(hipDeviceSynchronize());
      }
      double best = 1e30;
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) fn();
        CHECK(hipEventRecord(e1, nullptr));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms * 1e3 / iters < best ? ms * 1e3 / iters : best;
      }
      return best;
    };
    double t_one = 0, t_layers = 0, t_chain = 0;
    if (only != "chain") {
      t_one = time_it(step_one);
      t_layers = time_it(step_layers);
    }
    if (only != "layers") t_chain = time_it(step_chain);
    xsmm_hip_synchronize();
    const double flops = 2.0 * rows * N * K * layers;
    printf("{\"rows\": %d, \"layers\": %d, \"width\": %d, \"pad\": %d, \"kernel\": \"%s\", \"one_layer_us\": %.3f, \"per_layer_launches_us\": %.3f, "
           "\"chain_us\": %.3f, \"chain_one_launch\": %s, \"chain_tflops\": %.1f, \"layers_tflops\": %.1f}\n",
           rows, layers, width, pad, xsmm_hip_kernel_name(h), t_one, t_layers, t_chain, fused ? "true" : "false",
           t_chain > 0 ? flops / t_chain * 1e-6 : 0.0, t_layers > 0 ? flops / t_layers * 1e-6 : 0.0);
    fflush(stdout);
    CHECK(hipFree(x));
    for (int l = 0; l < layers; ++l) CHECK(hipFree(act[l]));
  }
  return 0;
}
