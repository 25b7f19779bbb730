// tensor_init.cpp - restatement of tpp-run's input generators (TEST INFRASTRUCTURE,
// part of the oracle: only tests/, smoke() and bench.py's cpu_baseline use it).
//
// Follows /root/reference/lib/TPP/Transforms/Utils/TensorInit.cpp:75-144 and
// TensorInitFloat.cpp:54-95 / include/TPP/Transforms/Utils/TensorInitFloat.h:115-152:
//   const  -> 1.0                      simple -> 0.3, 0.6, 0.9 repeating
//   cont   -> i / size                 random -> uniform_real_distribution<float>(0,1)
//   normal -> normal_distribution<float>(0, 0.2) clamped to [0, 1]
// with std::default_random_engine(seed) (libstdc++: minstd_rand0). One generator
// object is cached per (type, dtype, seed) and keeps its state ACROSS tensors
// (TensorInit.cpp:60,84-86): a handle here plays that role - fill the kernel
// arguments of one dtype in argument order from one handle (MLIRBench.cpp:215-243).
// Values are then rounded RNE to the element type by the caller (bf16) as
// TensorInitFloat.cpp:40-52 does through APFloat.
//
// Pinned by tests/golden/xsmm_fusion_seed123.json: with seed 123 this stream
// reproduces the FileCheck'd result of test/Integration/xsmm-fusion.mlir:54-57.
#include <algorithm>
#include <cstdint>
#include <random>

namespace {
struct Init {
  int kind; // 0 const, 1 simple, 2 cont, 3 random, 4 normal
  std::default_random_engine gen;
  std::uniform_real_distribution<float> uni{0.0f, 1.0f};
  std::normal_distribution<float> nrm{0.0f, 0.2f};
  Init(int k, int seed) : kind(k), gen(seed) {}
};
} // namespace

extern "C" {
void *tinit_create(int kind, int seed) { return new Init(kind, seed); }
void tinit_destroy(void *h) { delete static_cast<Init *>(h); }
void tinit_fill(void *h, float *out, int64_t n) {
  Init *t = static_cast<Init *>(h);
  switch (t->kind) {
  case 0:
    for (int64_t i = 0; i < n; ++i) out[i] = 1.0f;
    break;
  case 1: {
    const float d[3] = {0.3f, 0.6f, 0.9f};
    for (int64_t i = 0; i < n; ++i) out[i] = d[i % 3];
    break;
  }
  case 2: {
    const float norm = static_cast<float>(n);
    for (int64_t i = 0; i < n; ++i) out[i] = static_cast<float>(i) / norm;
    break;
  }
  case 3:
    for (int64_t i = 0; i < n; ++i) out[i] = t->uni(t->gen);
    break;
  default:
    for (int64_t i = 0; i < n; ++i) {
      float v = t->nrm(t->gen);
      out[i] = std::clamp(v, 0.0f, 1.0f);
    }
  }
}
}
