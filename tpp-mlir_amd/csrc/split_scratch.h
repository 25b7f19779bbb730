// split_scratch.h - device scratch of the SPLIT kernels (the batch-reduce range of an output tile over several workgroups,
// brgemm_f32_lw.hip / brgemm_bf16_small.hip): per (device, stream) ONE block of arrival counters (zero between launches: the last
// workgroup of a tile resets its counter) and partial tiles. Launches on one stream are ordered, so they share the block; launches
// on different streams never do. Allocated on first use, never freed (like the descriptors: the reference has no teardown call).
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <vector>

namespace tpp {

constexpr int SPLIT_MAX = 16;                       // workgroups per output tile at most
constexpr int SPLIT_MAX_TILES = 4096;               // arrival counters per block
constexpr size_t SPLIT_SCRATCH_FLOATS = 8u << 20;   // 32 MiB of partial tiles per stream (tiles * split * BM * BN floats per launch)

struct SplitScratch {
  int device;
  hipStream_t stream;
  unsigned *cnt;  // SPLIT_MAX_TILES words, zero whenever no split launch of this stream is in flight
  float *partial; // SPLIT_SCRATCH_FLOATS floats
};

// nullptr if the block cannot be allocated (the caller then launches without a split) or the launch does not fit it
static inline const SplitScratch *split_scratch_for(hipStream_t s, long long tiles, long long floats) {
  if (tiles > SPLIT_MAX_TILES || floats > (long long)SPLIT_SCRATCH_FLOATS) return nullptr;
  static std::mutex mu;
  static std::vector<SplitScratch *> blocks; // (pointers: a block's address is handed to launches while the vector grows)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  // a stream that is being captured into a graph: the launch runs unsplit - a captured split launch would carry this stream's block
  // into every replay of the graph, on whatever stream and next to whatever other launch (and nothing may be allocated now anyway)
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cs) != hipSuccess) (void)hipGetLastError();
  if (cs != hipStreamCaptureStatusNone) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  for (const SplitScratch *b : blocks)
    if (b->device == dev && b->stream == s) return b;
  SplitScratch *b = new SplitScratch{dev, s, nullptr, nullptr};
  if (hipMalloc((void **)&b->cnt, SPLIT_MAX_TILES * sizeof(unsigned)) != hipSuccess ||
      hipMalloc((void **)&b->partial, SPLIT_SCRATCH_FLOATS * sizeof(float)) != hipSuccess ||
      hipMemset(b->cnt, 0, SPLIT_MAX_TILES * sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipGetLastError();
    if (b->cnt) (void)hipFree(b->cnt);
    if (b->partial) (void)hipFree(b->partial);
    delete b;
    return nullptr;
  }
  blocks.push_back(b);
  return b;
}

} // namespace tpp
