// Hand-off latency between two workgroups on the SAME XCD and on DIFFERENT XCDs (gfx950), by cache policy - the question behind the
// seam of the layer chain: a producer stores 1 KiB, drains (s_waitcnt vmcnt(0)), raises a flag; the consumer
// polls the flag, loads the 1 KiB, checks it and raises the acknowledge flag; the producer polls that. One round trip = two hand-offs.
//   mode G ("agent scope", what the chain kernel does): write-through stores (sc1), agent-scope atomics, sc1 loads
//   mode L ("the XCD's own L2"): plain stores (write-back into the L2), workgroup-scope atomics (executed in the L2), sc0 loads (the
//          vector L1 is bypassed, the L2 answers) - only meaningful when both workgroups sit on one XCD; on different XCDs it must
//          FAIL (stale data or a flag that never arrives): the probe reports that too, it is the check that the policy is understood
//   mode M: the DATA as in L (plain stores, sc0 loads), the FLAGS as in G
// Workgroups that are not in an active pair run dependent FMAs for ~40 ms so that the clocks are up while the pairs measure (with
// all 128 pairs active the chip is idle - every wave sleeps in a poll loop - and the round trips come out 4x longer).
// Every workgroup records its XCC_ID (s_getreg_b32 HW_REG_XCC_ID) so that "same XCD" is measured, not assumed from blockIdx % 8.
// One workgroup per CU (LDS request), 256 workgroups, pairs (b, b + 8) [same XCD if block b runs on XCD b % 8] or (b, b ^ 1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __attribute__((address_space(1))) unsigned g_u32;

__device__ __forceinline__ unsigned ld(const unsigned* p, int mode) { // flag / data load that must not be served by the vector L1
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000);
  return mode ? __builtin_amdgcn_raw_buffer_load_b32(r, 0, 0, 1) : __builtin_amdgcn_raw_buffer_load_b32(r, 0, 0, 16);
}
__device__ __forceinline__ void st(unsigned* p, unsigned v, int mode) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000);
  if (mode) __builtin_amdgcn_raw_buffer_store_b32(v, r, 0, 0, 0);
  else __builtin_amdgcn_raw_buffer_store_b32(v, r, 0, 0, 16);
}
__device__ __forceinline__ void bump(unsigned* p, int mode) {
  if (mode) __hip_atomic_fetch_add((g_u32*)p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_fetch_add((g_u32*)p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait until *p >= want; false after ~2 ms
__device__ __forceinline__ bool wait_for(const unsigned* p, unsigned want, int mode) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    if ((int)(ld(p, mode) - want) >= 0) return true;
    if (__builtin_amdgcn_s_memrealtime() - t0 > 200000ull) return false;
    __builtin_amdgcn_s_sleep(1);
  }
}

// buffers per pair: data[256] (1 KiB), flag, ack - each in its own 128-byte line
__global__ __launch_bounds__(64) void k(unsigned* buf, unsigned* xcc, unsigned long long* out, int active_pairs, int partner_add, int mode, int iters, unsigned base) {
  extern __shared__ char smem[];
  const int b = blockIdx.x, lane = threadIdx.x;
  if (lane == 0) xcc[b] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15; // HW_REG_XCC_ID[3:0]
  int pair, role; // role 0 producer, 1 consumer
  if (partner_add) { role = (b / partner_add) & 1; pair = (b / (2 * partner_add)) * partner_add + b % partner_add; }
  else { role = b & 1; pair = b >> 1; }
  if (pair >= active_pairs) { // keep the chip busy (clocks up) while the active pairs measure: ~40 ms of dependent FMAs
    float a = (float)lane, c = 1.0001f;
    for (int i = 0; i < 2500000; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a = __builtin_fmaf(a, c, 0.5f);
    }
    if (a == 123.0f) out[0] = 1;
    return;
  }
  const int dmode = mode == 2 ? 1 : mode, fmode = mode == 2 ? 0 : mode; // M: data through the XCD's L2, flags as in G
  unsigned* data = buf + (size_t)pair * 1024;
  unsigned* flag = data + 512;
  unsigned* ack = data + 768;
  unsigned bad = 0, lost = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 1; i <= iters && !lost; ++i) {
    const unsigned tag = base + (unsigned)i;
    if (role == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) st(data + u * 64 + lane, tag + u, dmode);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) bump(flag, fmode);
      if (!wait_for(ack, tag, fmode)) lost = 1;
    } else {
      if (!wait_for(flag, tag, fmode)) lost = 1;
#pragma unroll
      for (int u = 0; u < 4; ++u) bad += ld(data + u * 64 + lane, dmode) != tag + u;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) bump(ack, fmode);
    }
    lost = __builtin_amdgcn_readfirstlane(lost);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (bad) atomicAdd((unsigned*)&out[3 * b + 1], bad);
  if (lane == 0) { out[3 * b] = t1 - t0; out[3 * b + 2] = lost; }
}

int main(int argc, char** argv) {
  const int NB = 256, iters = 2000, active = argc > 1 ? atoi(argv[1]) : 128; // pairs that run (the rest leave at once)
  unsigned *buf, *xcc; unsigned long long* out;
  hipMalloc(&buf, (size_t)NB * 4096); hipMalloc(&xcc, NB * 4); hipMalloc(&out, NB * 24);
  hipMemset(buf, 0, (size_t)NB * 4096);
  std::vector<unsigned> hx(NB); std::vector<unsigned long long> ho(3 * NB);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  const unsigned base = 0;
  for (int rep = 0; rep < 2; ++rep) for (int add : {8, 1}) for (int mode : {0, 1, 2}) {
    hipMemset(out, 0, NB * 24);
    hipMemset(buf, 0, (size_t)NB * 4096); // flags start at 0 in every launch (a launch that lost a flag leaves them anywhere)
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k, dim3(NB), dim3(64), 100 * 1024, 0, buf, xcc, out, active, add, mode, iters, base);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    hipMemcpy(hx.data(), xcc, NB * 4, hipMemcpyDeviceToHost); hipMemcpy(ho.data(), out, NB * 24, hipMemcpyDeviceToHost);
    int same = 0, pairs = 0, lost = 0; unsigned long long bad = 0; std::vector<double> rt;
    for (int b = 0; b < NB; ++b) {
      const int role = (b / add) & 1; if (role) continue;
      const int c = b + add, pr = add == 8 ? (b / 16) * 8 + b % 8 : b / 2;
      if (pr >= active) continue;
      ++pairs; same += hx[b] == hx[c];
      lost += (int)(ho[3 * b + 2] + ho[3 * c + 2]); bad += ho[3 * c + 1];
      rt.push_back(ho[3 * b] * 0.01 / iters); // us per round trip (100 MHz)
    }
    std::sort(rt.begin(), rt.end());
    printf("pairs (b, b+%d), mode %s: %3d of %3d pairs on one XCD | round trip (2 hand-offs) min / median / max %.2f / %.2f / %.2f us | pairs that lost a flag %d, stale data words %llu\n",
           add, mode == 2 ? "M (data: plain stores + sc0 loads; flags as in G)    " : mode ? "L (plain stores, workgroup-scope atomics, sc0 loads)" : "G (sc1 stores, agent atomics, sc1 loads)            ", same, pairs, rt.front(),
           rt[rt.size() / 2], rt.back(), lost, bad);
  }
  int ok = 0; for (int b = 0; b < NB; ++b) ok += (int)hx[b] == b % 8;
  printf("XCC_ID == blockIdx %% 8 for %d of %d workgroups\n", ok, NB);
  return 0;
}
