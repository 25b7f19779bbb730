// fake_hip.cpp - TEST INFRASTRUCTURE for tests/test_tsan_scheduler.py, never part of the product library.
//
// runtime.cpp (the C-ABI layer: tile queue, multi-producer ring, scheduler thread, host-operand mirroring) is
// compiled unchanged with g++ -fsanitize=thread and linked against THIS file instead of libamdhip64 and the
// gfx950 kernels, so that ThreadSanitizer can watch the host-side concurrency on a machine without a GPU:
//   * the 17 HIP host entry points runtime.cpp uses, over plain host memory ("device" allocations are
//     malloc'd blocks kept in a table so that hipPointerGetAttributes / hipMemGetAddressRange answer like
//     the driver); streams are in-order and every operation completes before its call returns;
//   * the 7 launch / plan functions of xsmm_desc.h as scalar f32 loops executed on the launching thread
//     (so a launch that touches bytes another thread is using IS a data race TSAN reports).
#include "../../tpp-mlir_amd/csrc/xsmm_desc.h"
#include "../../tpp-mlir_amd/csrc/chain_args.h"
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

namespace {
std::mutex g_mu;
std::map<uintptr_t, size_t> g_dev; // base -> size of live "device" allocations
bool find_alloc(const void *p, uintptr_t *base, size_t *size) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_dev.upper_bound((uintptr_t)p);
  if (it == g_dev.begin()) return false;
  --it;
  if ((uintptr_t)p >= it->first + it->second) return false;
  *base = it->first;
  *size = it->second;
  return true;
}
} // namespace

extern "C" {
hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "fake hip error"; }
hipError_t hipMalloc(void **p, size_t bytes) {
  *p = aligned_alloc(256, (bytes + 255) & ~(size_t)255);
  if (!*p) return hipErrorOutOfMemory;
  std::lock_guard<std::mutex> lk(g_mu);
  g_dev[(uintptr_t)*p] = bytes;
  return hipSuccess;
}
hipError_t hipFree(void *p) {
  if (!p) return hipSuccess;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    g_dev.erase((uintptr_t)p);
  }
  free(p);
  return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned int) {
  *p = aligned_alloc(256, (bytes + 255) & ~(size_t)255);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *attr, const void *p) {
  uintptr_t base;
  size_t size;
  if (!find_alloc(p, &base, &size)) return hipErrorInvalidValue; // plain host memory
  memset(attr, 0, sizeof(*attr));
  attr->type = hipMemoryTypeDevice;
  attr->devicePointer = (void *)p;
  return hipSuccess;
}
hipError_t hipMemGetAddressRange(hipDeviceptr_t *base, size_t *size, hipDeviceptr_t p) {
  uintptr_t b;
  size_t s;
  if (!find_alloc((const void *)p, &b, &s)) return hipErrorInvalidValue;
  *base = (hipDeviceptr_t)b;
  *size = s;
  return hipSuccess;
}
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind) {
  memcpy(dst, src, bytes);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind, hipStream_t) {
  memcpy(dst, src, bytes);
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height,
                            hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < height; ++r) memcpy((char *)dst + r * dpitch, (const char *)src + r * spitch, width);
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus *s) { *s = hipStreamCaptureStatusNone; return hipSuccess; }
hipError_t hipExtStreamGetCUMask(hipStream_t, uint32_t n, uint32_t *m) { for (uint32_t i = 0; i < n; ++i) m[i] = i < 8 ? 0xffffffffu : 0u; return hipSuccess; }
hipError_t hipMemset(void *p, int v, size_t bytes) { memset(p, v, bytes); return hipSuccess; }
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = nullptr; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
}

namespace tpp {

// FAKE_HIP_NO_COMPUTE=1: launches return at once (host-side throughput runs of the enqueue path, e.g. tools/tpp_replay
// built against this file; results are then meaningless)
static const bool g_compute = getenv("FAKE_HIP_NO_COMPUTE") == nullptr;

bool plan_gemm(GemmDesc &d, int) {
  d.variant = 0;
  strcpy(d.name, "fake_host_gemm");
  return d.dtype == DT_F32 && !d.vnni_b && !d.vnni_c;
}

hipError_t launch_gemm(const GemmDesc &d, const void *A_, const void *B_, void *C_, const void *D_, int64_t br, hipStream_t) {
  const float *A = (const float *)A_, *B = (const float *)B_, *D = (const float *)D_;
  float *C = (float *)C_;
  if (!g_compute) return hipSuccess;
  for (int64_t i = 0; i < d.m; ++i)
    for (int64_t j = 0; j < d.n; ++j) {
      float acc = d.beta0 ? 0.0f : C[i * d.ldc + j];
      for (int64_t b = 0; b < br; ++b)
        for (int64_t kk = 0; kk < d.k; ++kk)
          acc += A[b * d.stride_a + i * d.lda + kk] * (d.b_trans ? B[b * d.stride_b + j * d.ldb + kk] : B[b * d.stride_b + kk * d.ldb + j]);
      if (d.bias) acc += D[j];
      if (d.relu && !(acc > 0.0f)) acc = 0.0f;
      C[i * d.ldc + j] = acc;
    }
  return hipSuccess;
}
// the bf16 chain kernel has no host stand-in: plan_gemm above refuses bf16, so try_chain_launch never gets this far
bool bf16_fast_eligible(const GemmDesc &) { return false; }
int bf16_lw_b_kind(const GemmDesc &) { return -1; }
void blw_tile_dims(int, int *bm, int *bn) { *bm = *bn = 128; }
hipError_t launch_bf16_chain(int, int, const ChainArgs &, hipStream_t) { return hipErrorNotSupported; }
int f32_chain_tile(const GemmDesc &) { return -1; }
int force_gemm_split(int) { return -1; }
static bool g_fake_strict = false;
int set_strict_kernels(int on) { const int prev = g_fake_strict; g_fake_strict = on != 0; return prev; }
bool strict_kernels() { return g_fake_strict; }
bool f32_chain_tile_dims(int, int *bm, int *bn) { *bm = *bn = 64; return false; }
hipError_t launch_f32_chain(int, const ChainArgs &, hipStream_t) { return hipErrorNotSupported; }
const char *last_grouped_kernel() { return "fake_host_gemm"; }
const char *last_refined_kernel() { return ""; }
hipError_t launch_gemm_grouped(const GemmDesc &d, const WorkItem *it, int n, bool, bool, bool, int64_t, hipStream_t s) {
  for (int i = 0; i < n; ++i) (void)launch_gemm(d, it[i].A, it[i].B, it[i].C, it[i].D, it[i].br, s);
  return hipSuccess;
}
bool gemm_quads_pay(const GemmDesc &, int, int64_t) { return false; } // (plan_gemm above refuses bf16: no quads on the host stand-in)
hipError_t launch_gemm_quads(const GemmDesc &, const QuadItem *, int, int64_t, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_unary(const UnaryDesc &d, const void *in_, float scalar, bool use_scalar, void *out_, hipStream_t) {
  const float *in = (const float *)in_;
  float *out = (float *)out_;
  if (!g_compute) return hipSuccess;
  if (d.op == 29) { // transpose: m x n in, n x m out
    for (int64_t i = 0; i < d.m; ++i)
      for (int64_t j = 0; j < d.n; ++j) out[j * d.ldo + i] = in[i * d.ldi + j];
    return hipSuccess;
  }
  for (int64_t i = 0; i < d.m; ++i)
    for (int64_t j = 0; j < d.n; ++j) {
      float v = d.op == 2 ? 0.0f : (use_scalar ? scalar : in[i * d.ldi + j]);
      if (d.op == 5 && !(v > 0.0f)) v = 0.0f;
      out[i * d.ldo + j] = v;
    }
  return hipSuccess;
}
hipError_t launch_unary_grouped(const UnaryDesc &d, const WorkItem *it, int n, hipStream_t s) {
  for (int i = 0; i < n; ++i) (void)launch_unary(d, it[i].A, 0.0f, false, it[i].C, s);
  return hipSuccess;
}
hipError_t launch_binary(const BinaryDesc &d, const void *l_, const void *r_, void *out_, hipStream_t) {
  const float *l = (const float *)l_, *r = (const float *)r_;
  float *out = (float *)out_;
  if (!g_compute) return hipSuccess;
  for (int64_t i = 0; i < d.m; ++i)
    for (int64_t j = 0; j < d.n; ++j) {
      const float a = l[i * d.ldi_lhs + j], b = r[i * d.ldi_rhs + j];
      out[i * d.ldo + j] = d.op == 1 ? a + b : d.op == 2 ? a * b : d.op == 3 ? a - b : a / b;
    }
  return hipSuccess;
}
hipError_t launch_binary_grouped(const BinaryDesc &d, const WorkItem *it, int n, hipStream_t s) {
  for (int i = 0; i < n; ++i) (void)launch_binary(d, it[i].A, it[i].B, it[i].C, s);
  return hipSuccess;
}

} // namespace tpp
