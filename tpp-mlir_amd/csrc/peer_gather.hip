// peer_gather.hip - the all-gather of the row-sharded MLP's output WITHOUT a collective library call: every rank stores its
// row block straight into every peer's output buffer through IPC-mapped device pointers (xGMI writes on a node; the same HBM when
// two ranks share a device), completion by flags. One process per GPU as before; RCCL stays the fallback (mlp.py).
//
// Why ("prefer a direct / all-to-all-style algorithm or a custom P2P kernel: RCCL launch latency dominates at this
// size"): the output of the bs = 4096 MLP is 8 MiB in total - 1 MiB per rank at 8 GPUs - and a dist.all_gather_into_tensor
// call costs >= 11 us of launch + protocol latency before a byte moves, as much as a rank's whole
// three-layer step. Here the gather is TWO small launches on the rank's own stream:
//   scatter : block (c, w) copies chunk c of the local block to peer w's buffer at this rank's row offset (16 B / lane); every
//             block fences (system scope) and takes a ticket; the last one raises flag[rank] = epoch in EVERY peer's flag array;
//   wait    : one wave polls this rank's own flag array until every peer's flag has reached the epoch (bounded; a timeout sets
//             *err, which the host checks) - kernels behind it on the stream see the gathered output.
// Re-use across steps: the output buffers are double-buffered by epoch parity, and before a block writes into peer w's buffer it
// waits until peer w has ENTERED the previous step's gather (ready[w] >= epoch - 1: everything peer w enqueued before that -
// the consumers of the buffer being overwritten - has finished). That wait is one step old: satisfied on arrival in steady state.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>

namespace {

typedef unsigned int u32x4_pg __attribute__((ext_vector_type(4)));
constexpr unsigned long long PG_TIMEOUT_TICKS = 200000000ull; // 2 s of s_memrealtime (100 MHz): peers are other processes

struct ScatterArgs {
  const void *src;
  long long bytes;      // of the local block (multiple of 16)
  long long dst_offset; // this rank's offset inside every peer's buffer
  void *dst[16];        // peer output buffers of this epoch's parity (own rank: the local one)
  unsigned *flags[16];  // peer flag arrays [world]: flags[w][rank] = epoch when this rank's block has landed in w
  unsigned *ready[16];  // peer ready arrays [world]: ready[w][rank] = epoch when this rank has entered the gather of that epoch
  unsigned *my_ready;   // this rank's own ready array (what the peers wrote)
  unsigned *ticket;     // device counter, zero between launches
  unsigned *err;
  unsigned epoch;
  int world, rank, chunks;
};

__device__ __forceinline__ bool pg_wait_ge(unsigned *word, unsigned want, unsigned *err, unsigned code) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    const unsigned v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((int)(v - want) >= 0) return true;
    if (__builtin_amdgcn_s_memrealtime() - t0 > PG_TIMEOUT_TICKS) {
      __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return false;
    }
    __builtin_amdgcn_s_sleep(8);
  }
}

__global__ __launch_bounds__(256) void peer_scatter_kernel(ScatterArgs a) {
  const int c = blockIdx.x, w = blockIdx.y;
  __shared__ int go;
  if (threadIdx.x == 0) {
    // announce: this rank has entered the gather of `epoch` (its earlier work on this stream - the consumers of the buffer
    // peers will overwrite NEXT step - is done). One block per peer does it.
    if (c == 0) __hip_atomic_store(a.ready[w] + a.rank, a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // peer w must have entered the PREVIOUS gather before its buffer of this parity is overwritten
    go = (w == a.rank || a.epoch < 2) ? 1 : (int)pg_wait_ge(a.my_ready + w, a.epoch - 1, a.err, 0x100u + (unsigned)w);
  }
  __syncthreads();
  if (go) {
    const long long per = ((a.bytes / 16 + a.chunks - 1) / a.chunks) * 16; // bytes per chunk (16-byte pieces)
    const long long b0 = (long long)c * per, b1 = b0 + per < a.bytes ? b0 + per : a.bytes;
    const char *s = (const char *)a.src;
    char *d = (char *)a.dst[w] + a.dst_offset;
    for (long long o = b0 + (long long)threadIdx.x * 16; o < b1; o += 256 * 16)
      __builtin_nontemporal_store(*(const u32x4_pg *)(s + o), (u32x4_pg *)(d + o));
  }
  __threadfence_system(); // this block's stores are visible to every agent before its ticket is
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned total = gridDim.x * gridDim.y;
    const unsigned t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t == total - 1) { // the last block: every block's stores are out (they fenced before their ticket)
      __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      for (int p = 0; p < a.world; ++p) __hip_atomic_store(a.flags[p] + a.rank, a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ __launch_bounds__(64) void peer_wait_kernel(unsigned *my_flags, int world, unsigned epoch, unsigned *err) {
  if ((int)threadIdx.x < world) (void)pg_wait_ge(my_flags + threadIdx.x, epoch, err, 0x200u + threadIdx.x);
  __threadfence_system();
}

[[noreturn]] void pg_die(const char *what, hipError_t e) {
  fprintf(stderr, "tpp-xsmm-hip: %s failed: %s (peer gather; there is no CPU fallback)\n", what, hipGetErrorString(e));
  fflush(stderr);
  exit(-1);
}
#define PG_OK(expr)                                \
  do {                                             \
    hipError_t e_ = (expr);                        \
    if (e_ != hipSuccess) pg_die(#expr, e_);       \
  } while (0)

} // namespace

extern "C" void *xsmm_hip_get_stream(void);

namespace {
// OVERLAP mode: the wait kernel of a gather goes to a side stream. The scatter stays on the runtime's stream (it reads the rank's
// fresh output), but waiting for the PEERS' blocks to land - the xGMI transfer time of the step - no longer sits between this
// step's kernels and the next step's: the next step's compute starts right behind the scatter. The wait kernel needs no stream
// order to be correct (it polls the flags of ITS epoch; this rank's own flag is raised by its own scatter), it is one wave without
// LDS (it fits beside a chain kernel that fills every CU), and buffer re-use across steps is guarded by the ready words as before.
// What changes for the caller: "the gathered output of step e is complete" is the side stream's business - xsmm_hip_peer_drain()
// (host) or an event on xsmm_hip_peer_wait_stream() before consuming it.
std::atomic<int> g_overlap{0};
hipStream_t g_wait_stream = nullptr;
std::mutex g_wait_mu;
hipStream_t wait_stream() {
  std::lock_guard<std::mutex> lk(g_wait_mu);
  if (!g_wait_stream) PG_OK(hipStreamCreateWithFlags(&g_wait_stream, hipStreamNonBlocking));
  return g_wait_stream;
}
} // namespace

extern "C" int xsmm_hip_peer_overlap(int enable) {
  if (enable) (void)wait_stream();
  return g_overlap.exchange(enable != 0);
}
extern "C" void *xsmm_hip_peer_wait_stream(void) { return (void *)wait_stream(); }
extern "C" void xsmm_hip_peer_drain(void) {
  hipStream_t s = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_wait_mu);
    s = g_wait_stream;
  }
  if (s) PG_OK(hipStreamSynchronize(s));
}

// a dedicated, zeroed device allocation (IPC handles name whole allocations: nothing here is carved out of a caching allocator)
// Peer-visible allocations - the flag / ready words that peers on OTHER devices write and this device polls, and the output
// buffers they store their row blocks into - are taken FINE-GRAINED when the runtime offers it: such memory is not held in this
// device's L2, so neither a poll nor a later read of a gathered output can be served a stale line that a remote store never
// touched (a remote write reaches the memory, not this device's cache). Ordinary device memory is the fallback; PeerGather's
// self-test (peer.py: three gathers incl. a re-use of a buffer this device has read before) decides whether the result can be used.
extern "C" void *xsmm_hip_peer_alloc(int64_t bytes) {
  void *p = nullptr;
  static const bool fine = [] {
    const char *e = getenv("TPP_HIP_PEER_FINEGRAINED"); // 0: plain hipMalloc (A/B runs)
    return !e || atoi(e) != 0;
  }();
  if (fine && hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained) == hipSuccess && p) {
    hipIpcMemHandle_t probe;
    if (hipIpcGetMemHandle(&probe, p) == hipSuccess) { // (only if it can be shared like the others)
      PG_OK(hipMemset(p, 0, (size_t)bytes));
      PG_OK(hipDeviceSynchronize());
      return p;
    }
    (void)hipGetLastError();
    (void)hipFree(p);
    p = nullptr;
  } else {
    (void)hipGetLastError();
  }
  PG_OK(hipMalloc(&p, (size_t)bytes));
  PG_OK(hipMemset(p, 0, (size_t)bytes));
  PG_OK(hipDeviceSynchronize());
  return p;
}
extern "C" void xsmm_hip_peer_free(void *p) {
  if (p) PG_OK(hipFree(p));
}
// handle_out: 64 bytes (hipIpcMemHandle_t); returns 0 / -1
extern "C" int xsmm_hip_ipc_export(void *ptr, void *handle_out) {
  hipIpcMemHandle_t h;
  const hipError_t e = hipIpcGetMemHandle(&h, ptr);
  if (e != hipSuccess) {
    fprintf(stderr, "tpp-xsmm-hip: hipIpcGetMemHandle: %s\n", hipGetErrorString(e));
    (void)hipGetLastError();
    return -1;
  }
  static_assert(sizeof(h) == 64, "hipIpcMemHandle_t is 64 bytes");
  memcpy(handle_out, &h, sizeof(h));
  return 0;
}
// maps a peer process's allocation into this process (peer access enabled lazily); nullptr on failure
extern "C" void *xsmm_hip_ipc_open(const void *handle) {
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  void *p = nullptr;
  const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) {
    fprintf(stderr, "tpp-xsmm-hip: hipIpcOpenMemHandle: %s\n", hipGetErrorString(e));
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}
extern "C" int xsmm_hip_ipc_close(void *p) { return hipIpcCloseMemHandle(p) == hipSuccess ? 0 : -1; }

// One step of the gather on the runtime's stream. dst / flags / ready: `world` pointers each (this rank's own entries are its local
// buffers). ticket: one zeroed device word of this rank; err: one device word of this rank (non-zero after a timeout).
extern "C" void xsmm_hip_peer_gather(const void *src, int64_t bytes, int64_t dst_offset, int64_t world, int64_t rank, void *const *dst,
                                     void *const *flags, void *const *ready, void *my_flags, void *my_ready, void *ticket, void *err,
                                     int64_t epoch) {
  if (world < 1 || world > 16 || rank < 0 || rank >= world || (bytes & 15) || (dst_offset & 15)) {
    fprintf(stderr, "tpp-xsmm-hip: xsmm_hip_peer_gather: world 1..16, 16-byte-multiple sizes (world %ld rank %ld bytes %ld offset %ld)\n",
            (long)world, (long)rank, (long)bytes, (long)dst_offset);
    exit(-1);
  }
  ScatterArgs a;
  a.src = src;
  a.bytes = bytes;
  a.dst_offset = dst_offset;
  for (int w = 0; w < world; ++w) {
    a.dst[w] = dst[w];
    a.flags[w] = (unsigned *)flags[w];
    a.ready[w] = (unsigned *)ready[w];
  }
  a.my_ready = (unsigned *)my_ready;
  a.ticket = (unsigned *)ticket;
  a.err = (unsigned *)err;
  a.epoch = (unsigned)epoch;
  a.world = (int)world;
  a.rank = (int)rank;
  // about 16 KiB per block (4 x 16 bytes per lane), at most ~1024 blocks in the launch: a 1 MiB block is 64 blocks x world
  long long chunks = (bytes + 16383) / 16384;
  const long long cap = 1024 / world > 1 ? 1024 / world : 1;
  a.chunks = (int)(chunks < 1 ? 1 : chunks > cap ? cap : chunks);
  hipStream_t s = (hipStream_t)xsmm_hip_get_stream();
  {
    // the epoch is an argument of both kernels: replayed from a graph the flags would never reach it again
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
      fprintf(stderr, "tpp-xsmm-hip: xsmm_hip_peer_gather cannot be captured into a graph (its epoch is per call)\n");
      exit(-1);
    }
    (void)hipGetLastError();
  }
  hipLaunchKernelGGL(peer_scatter_kernel, dim3((unsigned)a.chunks, (unsigned)world), dim3(256), 0, s, a);
  PG_OK(hipGetLastError());
  hipStream_t ws = g_overlap.load(std::memory_order_relaxed) ? wait_stream() : s;
  hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(64), 0, ws, (unsigned *)my_flags, (int)world, (unsigned)epoch, (unsigned *)err);
  PG_OK(hipGetLastError());
}
