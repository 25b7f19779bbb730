// split_handoff - the hand-off of a SPLIT launch (brgemm_f32_lw<..., SPLIT>: S workgroups share one output tile's batch-reduce range,
// the last to arrive sums the S partial tiles in split order) by itself, two protocols ("attack R = 2.6 us once"):
//   A (the shipped one): partial stored write-through -> s_waitcnt vmcnt(0) (stores acknowledged) -> barrier -> counter += 1 (returns the
//     arrival index) -> the last arriver loads the S partials -> sums -> stores C. Three dependent trips to the memory side.
//   C (data-carried): the scratch slots hold a SENTINEL pattern between launches; partial stored write-through, the counter add is
//     issued WITHOUT waiting for the stores' acknowledgement; the last arriver loads the partials and re-loads any 16-byte piece that
//     still reads as the sentinel (the store that is in flight lands within one memory latency: no workgroup waits for another
//     workgroup's PROGRESS, only for the memory system - the property of protocol A is kept); after the sum it writes the sentinel
//     back. Two dependent trips.
// Nothing else in the kernel (no K loop): the kernel time IS the hand-off + launch floor. 128 tiles x S workgroups of 256 threads,
// a 32x32 f32 partial per workgroup... (4 KiB) or 64x64 (16 KiB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(1))) unsigned g_u32;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr unsigned SENT = 0xFFBADA55u;

template <int MODE, int PIECES> // PIECES 16-byte pieces per thread (256 threads): 1 = 4 KiB (32x32 f32), 4 = 16 KiB (64x64)
__global__ __launch_bounds__(256) void k(float *scratch, unsigned *cnt, float *out, int S, float seed) {
  __shared__ unsigned flag;
  const int tile = blockIdx.x / S, sp = blockIdx.x % S, tid = threadIdx.x;
  constexpr int TILE = 256 * PIECES * 4;
  float *scr = scratch + (size_t)tile * S * TILE;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)scr, 0, 0x7fffffff, 0x00020000);
  f32x4 part[PIECES];
#pragma unroll
  for (int j = 0; j < PIECES; ++j) part[j] = f32x4{seed + sp, seed + tid, seed + j, 1.0f};
#pragma unroll
  for (int j = 0; j < PIECES; ++j)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, part[j]), r, (unsigned)(tid * 16), (unsigned)((sp * TILE + j * 1024) * 4), 16);
  if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) flag = __hip_atomic_fetch_add((g_u32 *)(cnt + tile), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (flag != (unsigned)(S - 1)) return;
  if (tid == 0) __hip_atomic_store((g_u32 *)(cnt + tile), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  f32x4 acc[PIECES];
#pragma unroll
  for (int j = 0; j < PIECES; ++j) acc[j] = f32x4{0, 0, 0, 0};
  for (int s2 = 0; s2 < S; ++s2) {
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      u32x4 v;
      if (MODE == 1 && s2 == sp) {
        v = __builtin_bit_cast(u32x4, part[j]); // its own partial: from the registers (its store may still be in flight)
      } else {
        v = __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)(tid * 16), (unsigned)((s2 * TILE + j * 1024) * 4), 16);
        if (MODE == 1) {
          const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
          while ((v[0] == SENT || v[1] == SENT || v[2] == SENT || v[3] == SENT) && __builtin_amdgcn_s_memrealtime() - t0 < 2000ull) // 20 us bound
            v = __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)(tid * 16), (unsigned)((s2 * TILE + j * 1024) * 4), 16);
        }
      }
      acc[j] += __builtin_bit_cast(f32x4, v);
    }
  }
#pragma unroll
  for (int j = 0; j < PIECES; ++j) *(f32x4 *)(out + (size_t)tile * TILE + j * 1024 + tid * 4) = acc[j];
  if (MODE == 1) { // the sentinel goes back into the slots (behind the result: off this launch's critical path)
    const u32x4 sv = {SENT, SENT, SENT, SENT};
    for (int s2 = 0; s2 < S; ++s2)
#pragma unroll
      for (int j = 0; j < PIECES; ++j) __builtin_amdgcn_raw_buffer_store_b128(sv, r, (unsigned)(tid * 16), (unsigned)((s2 * TILE + j * 1024) * 4), 16);
  }
}

template <int MODE, int PIECES> static void run(const char *name, int tiles, int S) {
  constexpr int TILE = 256 * PIECES * 4;
  float *scratch, *out;
  unsigned *cnt;
  CHECK(hipMalloc(&scratch, (size_t)tiles * S * TILE * 4));
  CHECK(hipMalloc(&out, (size_t)tiles * TILE * 4));
  CHECK(hipMalloc(&cnt, tiles * 4));
  CHECK(hipMemset(cnt, 0, tiles * 4));
  std::vector<unsigned> sent((size_t)tiles * S * TILE, SENT);
  CHECK(hipMemcpy(scratch, sent.data(), sent.size() * 4, hipMemcpyHostToDevice));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int N = 3000;
  for (int i = 0; i < 500; ++i) hipLaunchKernelGGL((k<MODE, PIECES>), dim3(tiles * S), dim3(256), 0, s, scratch, cnt, out, S, (float)(i & 7));
  CHECK(hipStreamSynchronize(s));
  CHECK(hipEventRecord(e0, s));
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL((k<MODE, PIECES>), dim3(tiles * S), dim3(256), 0, s, scratch, cnt, out, S, (float)(i & 7));
  CHECK(hipEventRecord(e1, s));
  CHECK(hipStreamSynchronize(s));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  // check the last launch: out = sum over sp of (seed + sp, seed + tid, seed + j, 1)
  std::vector<float> h((size_t)tiles * TILE);
  CHECK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
  const float seed = (float)((N - 1) & 7);
  size_t bad = 0;
  for (int t = 0; t < tiles; ++t)
    for (int j = 0; j < PIECES; ++j)
      for (int tid = 0; tid < 256; ++tid) {
        const float *v = &h[(size_t)t * TILE + j * 1024 + tid * 4];
        float e0_ = 0;
        for (int sp = 0; sp < S; ++sp) e0_ += seed + sp;
        bad += !(v[0] == e0_ && v[1] == S * (seed + tid) && v[2] == S * (seed + j) && v[3] == (float)S);
      }
  printf("  %-34s tiles %3d S %d partial %2d KiB: %6.2f us per launch, %zu wrong pieces\n", name, tiles, S, TILE * 4 / 1024, ms * 1e3 / N, bad);
  CHECK(hipFree(scratch));
  CHECK(hipFree(out));
  CHECK(hipFree(cnt));
}

__global__ void empty_k(float *p) { if (p == (float *)1) p[0] = 0; }

int main() {
  {
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 500; ++i) hipLaunchKernelGGL(empty_k, dim3(256), dim3(256), 0, s, (float *)nullptr);
    CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < 3000; ++i) hipLaunchKernelGGL(empty_k, dim3(256), dim3(256), 0, s, (float *)nullptr);
    CHECK(hipEventRecord(e1, s));
    CHECK(hipStreamSynchronize(s));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("  empty launch <<<256, 256>>>: %.2f us\n", ms * 1e3 / 3000);
  }
  for (int rep = 0; rep < 2; ++rep) {
    for (int S : {2, 4, 8}) {
      const int tiles = 256 / S;
      run<0, 1>("A counter, stores acknowledged", tiles, S);
      run<1, 1>("C sentinel, no ack wait", tiles, S);
      run<0, 4>("A counter, stores acknowledged", tiles, S);
      run<1, 4>("C sentinel, no ack wait", tiles, S);
    }
  }
  return 0;
}
