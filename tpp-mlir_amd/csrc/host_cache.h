// host_cache.h - host operands kept on the device between invokes (round 6).
//
// An UNMODIFIED tpp-run hands host pointers to the xsmm entry points (memref globals / malloc: lib/TPP/Runner/MLIRBench.cpp:207-246;
// the reference's "completion / ownership" contract). The plain mirror path copies every operand over PCIe on every
// invoke (C2: 12 MiB up, 4 MiB down = 295 us around a 17 us kernel). This box has no HMM (hipDeviceAttributePageableMemoryAccess = 0,
// xnack-, hipMemPrefetchAsync on malloc memory "not supported": profiles/r06_pageable_probe.txt), so the GPU cannot use that memory in
// place. What the kernel DOES offer to an unprivileged process: userfaultfd write-protection in ASYNC mode (UFFD_FEATURE_WP_ASYNC,
// Linux 6.7+) - a write to a protected page never faults to user space, the kernel just remembers it - and PAGEMAP_SCAN, which returns
// "the pages of this range written since I last asked" and protects them again in one call. On top of that:
//
//   * a host operand gets a device MIRROR that outlives the invoke (an "extent": the page-aligned hull of the operands seen at those
//     addresses, grown / merged as tiles of one buffer arrive);
//   * before a kernel uses mirror pages, the pages the host has written since the last look are uploaded again - nothing else is;
//   * what a kernel writes is copied back to the host bytes it wrote (footprint-exact, like the plain path): in synchronous mode
//     before the invoke returns (the reference's contract, unchanged), in asynchronous mode at the next synchronisation point
//     (xsmm_hip_synchronize / perf_stop_timer / xsmm_hip_set_async(0) / xsmm_hip_set_stream) - the async contract of
//     include/tpp_xsmm_abi.h ("operands may be freed / re-allocated only after ...") extended by its natural twin: the host neither
//     reads outputs nor writes inputs of the region in between. In asynchronous mode an extent is polled once per synchronisation
//     epoch, so the steady state of a timing loop costs a table lookup per operand and the tile queue sees device pointers.
//   * lifetime needs no hook: a range that was unmapped and mapped again is no longer registered - the scan refuses it (EPERM) and
//     the extent is dropped; a freed-and-reused heap chunk reads as "written". Nothing is ever trusted that the kernel does not vouch for.
//   * no fault handler, no signal, no thread: system calls that write into tracked memory just work (tools/ubench/wp_async_probe.cpp).
//
//   * one thing defeats the tracking without breaking it: a buffer that was ever the source / destination of a plain hipMemcpy stays in
//     the ROCm runtime's pin cache, and the driver validates its pages again WITH WRITE ACCESS after every later piece of driver activity -
//     they then read as written (tools/ubench/wp_vs_hipmemcpy.cpp). The cache uploads such a buffer again and again: correct, as slow as
//     the plain path. The cache itself never hands caller pages to a driver copy (its own pinned staging buffer + CPU copies).
//
// OFF by default (TPP_HIP_HOST_CACHE=1 / xsmm_hip_set_host_cache(1)); needs Linux >= 6.7 with userfaultfd(UFFD_USER_MODE_ONLY).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstddef>
#include <cstdint>

namespace tpp {
namespace hc {

struct OpRef {
  void **ptr;      // in: host (or device) pointer of the operand; out: its mirror address if translated
  size_t bytes;    // bounding range
  size_t rows, row_bytes, pitch; // 2-D footprint (rows == 0: one dense range)
  bool read, written;
  void *host;      // set by translate(): the original host pointer (nullptr: not translated)
  int kind;        // set by translate(): 0 untouched, 1 mirror of an extent, 2 scratch (a pure output of a synchronous invoke)
};

struct Hooks {
  void (*flush_queue)();                 // launch whatever the tile queue (and the deferred transposes) still hold
  bool (*is_device)(const void *p, int pos); // device memory (the caller's per-thread cache)
};
void set_hooks(const Hooks &h);

extern std::atomic<int> g_on;
inline bool enabled() { return g_on.load(std::memory_order_relaxed) != 0; }
int set_enabled(int on); // previous setting, or -1 if the kernel interface is missing (the cache stays off)

// One invoke. translate() maps every host operand to its mirror (creating / growing extents, polling, uploading); the calling thread is
// inside a "reader section" until leave() - mirrors are not moved meanwhile. Returns the number of operands translated (0: nothing to do;
// leave() must still be called when it returned > 0 ... it is harmless otherwise).
int translate(OpRef *ops, int n, bool async, uint64_t epoch, hipStream_t s);
// synchronous mode: the kernel has completed - copy what it wrote back to the host now. asynchronous mode: remember the footprints.
void complete(OpRef *ops, int n, bool async, uint64_t epoch, hipStream_t s);
void leave();
// Asynchronous mode, whole-invoke memo: a timing loop issues the SAME invokes every iteration (descriptor, four pointers, batch count).
// memo_hit() answers such an invoke from the calling thread's memo without looking at the operands at all - it replaces the pointers,
// enters the reader section and returns a token for memo_done() (which notes the written footprint once per epoch and leaves the
// section); nullptr: not in the memo (or no longer good: an extent moved, a page of one of its extents lost its valid bit, another
// epoch / stream) - the caller takes translate() and offers the result to memo_store().
void *memo_hit(const void *desc, void **p0, void **p1, void **p2, void **p3, int64_t br, uint64_t epoch, hipStream_t s);
void memo_done(void *token, uint64_t epoch);
void memo_store(const void *desc, int64_t br, const OpRef *ops, int n, uint64_t epoch, hipStream_t s);
// the stream has been drained (synchronisation point): write back everything pending
void on_sync_point(hipStream_t s);
void stats(int64_t out[10]);

} // namespace hc
} // namespace tpp
