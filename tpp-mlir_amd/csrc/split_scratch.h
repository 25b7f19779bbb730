// split_scratch.h - device scratch of the SPLIT kernels (the batch-reduce range of an output tile over several workgroups,
// brgemm_f32_lw.hip / brgemm_bf16_small.hip): per (device, stream) ONE block of arrival counters (zero between launches: the last
// workgroup of a tile resets its counter) and partial tiles. Launches on one stream are ordered, so they share the block; launches
// on different streams never do. Allocated on first use and grown to the launches seen (1 MiB doubling up to 32 MiB), at most
// SPLIT_MAX_BLOCKS of them, never freed (like the descriptors: the reference has no teardown call).
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
  float *partial; // `floats` floats
  size_t floats;  // sized to the launches seen (round 6, ADVICE r5: was 32 MiB per stream from the first split launch on)
};
constexpr int SPLIT_MAX_BLOCKS = 16; // (device, stream) pairs that get a block; a 17th stream's launches run unsplit (correct, slower):
                                     // an application that creates and destroys streams no longer leaks a block per stream

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
  SplitScratch *b = nullptr;
  for (SplitScratch *c : blocks)
    if (c->device == dev && c->stream == s) b = c;
  if (b && b->floats >= (size_t)floats) return b;
  if (!b && (int)blocks.size() >= SPLIT_MAX_BLOCKS) return nullptr;
  // a new block, or a larger partial area for this stream (its earlier launches may still read the old one: drain the stream first -
  // once per growth step, the sizes double)
  size_t want = (size_t)1 << 18; // 1 MiB
  while (want < (size_t)floats) want <<= 1;
  if (want > SPLIT_SCRATCH_FLOATS) want = SPLIT_SCRATCH_FLOATS;
  float *area = nullptr;
  if (hipMalloc((void **)&area, want * sizeof(float)) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  if (b) {
    if (hipStreamSynchronize(s) != hipSuccess) (void)hipGetLastError();
    (void)hipFree(b->partial);
    b->partial = area;
    b->floats = want;
    return b;
  }
  b = new SplitScratch{dev, s, nullptr, area, want};
  // (the counters are zeroed ON the stream, in front of the launch that asked for the block: no device-wide synchronisation)
  if (hipMalloc((void **)&b->cnt, SPLIT_MAX_TILES * sizeof(unsigned)) != hipSuccess ||
      hipMemsetAsync(b->cnt, 0, SPLIT_MAX_TILES * sizeof(unsigned), s) != hipSuccess) {
    (void)hipGetLastError();
    if (b->cnt) (void)hipFree(b->cnt);
    (void)hipFree(b->partial);
    delete b;
    return nullptr;
  }
  blocks.push_back(b);
  return b;
}

} // namespace tpp
