/*
 * tpp_xsmm_abi.h - the dispatch/invoke C-ABI of tpp-mlir's runtime/Xsmm,
 * re-declared for the MI355X-native runtime (libtpp_xsmm_runner_utils.so).
 *
 * Every entry point below replaces the symbol of the same name exported by the
 * reference's libtpp_xsmm_runner_utils.so. Reference declarations:
 *   runtime/Xsmm/XsmmRunnerUtils.h:22-83   (13 xsmm_* symbols)
 *   runtime/PerfRunnerUtils.h:22-24        (perf_start_timer / perf_stop_timer)
 * The reference header types the enum arguments as libxsmm enums; the compiler
 * always materialises them as i64 constants (lib/TPP/Conversion/ConvertXsmmToFunc/
 * ConvertXsmmToFunc.cpp:63-65, 319-340), and libxsmm is not a dependency of this
 * runtime, so they are int64_t here. On x86-64 SysV the two declarations are
 * call-compatible for the value range the compiler emits (non-negative, < 2^31).
 *
 * All matrices are row-major. `off*` are element offsets (memref offsets), added
 * as typed pointer arithmetic (XsmmRunnerUtils.cpp:63-75).
 *
 * Data pointers may be device pointers (used in place, zero copy) or host
 * pointers (staged through a device mirror; results are copied back before the
 * invoke returns, preserving the reference's synchronous completion contract).
 */
#ifndef TPP_XSMM_ABI_H
#define TPP_XSMM_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TPP_XSMM_EXPORT __attribute__((visibility("default")))

/* ---- wire values: include/TPP/Dialect/Xsmm/XsmmEnum.td:13-84 ---------------- */
enum { XSMM_DTYPE_F32 = 1, XSMM_DTYPE_BF16 = 2 };                          /* :13-20 */
enum {                                                                     /* :34-45 */
  XSMM_UNARY_NONE = 0, XSMM_UNARY_IDENTITY = 1, XSMM_UNARY_ZERO = 2,
  XSMM_UNARY_RELU = 5, XSMM_UNARY_VNNI2 = 28, XSMM_UNARY_TRANSPOSE = 29
};
enum {                                                                     /* :47-56 */
  XSMM_UNARY_FLAG_NONE = 0, XSMM_UNARY_FLAG_BCAST_ROW = 2,
  XSMM_UNARY_FLAG_BCAST_COL = 4, XSMM_UNARY_FLAG_BCAST_SCALAR = 8
};
enum {                                                                     /* :22-32 */
  XSMM_BINARY_NONE = 0, XSMM_BINARY_ADD = 1, XSMM_BINARY_MUL = 2,
  XSMM_BINARY_SUB = 3, XSMM_BINARY_DIV = 4
};
enum {                                                                     /* :58-70 */
  XSMM_BINARY_FLAG_NONE = 0,
  XSMM_BINARY_FLAG_BCAST_ROW_IN_0 = 1, XSMM_BINARY_FLAG_BCAST_ROW_IN_1 = 2,
  XSMM_BINARY_FLAG_BCAST_COL_IN_0 = 4, XSMM_BINARY_FLAG_BCAST_COL_IN_1 = 8,
  XSMM_BINARY_FLAG_BCAST_SCALAR_IN_0 = 16, XSMM_BINARY_FLAG_BCAST_SCALAR_IN_1 = 32
};
/* GEMM flags AS THEY ARRIVE ON THE WIRE. The dialect values are VNNI_A=2048,
 * VNNI_B=4096 (XsmmEnum.td:72-84) but ConvertXsmmToFunc.cpp:251-265 exchanges the
 * two because the runtime swaps A and B for libxsmm's column-major view: a
 * dialect `vnni_b` operand arrives as 2048, `vnni_a` as 4096. */
enum {
  XSMM_GEMM_FLAG_NONE = 0, XSMM_GEMM_FLAG_BETA_0 = 4,
  XSMM_GEMM_FLAG_NO_RESET_TILECONFIG = 64, XSMM_GEMM_FLAG_NO_SETUP_TILECONFIG = 128,
  XSMM_GEMM_WIRE_VNNI_B = 2048, /* row-major B operand is [K/2][N][2] */
  XSMM_GEMM_WIRE_VNNI_A = 4096, /* row-major A operand is [M][K/2][2] = plain row-major bytes (VNNIUtils.cpp:75-77) */
  XSMM_GEMM_FLAG_VNNI_C = 8192  /* C stored / read as VNNI-2 [M/2][N][2], ldc = pair-row stride / 2 (generic kernel) */
};

/* ---- dispatch: build (or look up) a kernel descriptor, return opaque handle ----
 * Handles are never freed (the reference has no destroy call); dispatching the
 * same arguments again returns the same handle. Unsupported arguments: message
 * on stderr + exit(-1), as XsmmRunnerUtils.cpp:132-137,170-176,202-208,352-358. */

/* replaces XsmmRunnerUtils.cpp:95-140 */
TPP_XSMM_EXPORT int64_t xsmm_gemm_dispatch(int64_t dtype, int64_t m, int64_t n, int64_t k,
                                           int64_t lda, int64_t ldb, int64_t ldc,
                                           int64_t flags);
/* replaces XsmmRunnerUtils.cpp:308-361 */
TPP_XSMM_EXPORT int64_t xsmm_brgemm_dispatch(int64_t dtype, int64_t m, int64_t n, int64_t k,
                                             int64_t lda, int64_t ldb, int64_t ldc,
                                             int64_t stride_a, int64_t stride_b,
                                             int64_t flags);
/* replaces XsmmRunnerUtils.cpp:385-457 */
TPP_XSMM_EXPORT int64_t xsmm_fused_brgemm_dispatch(int64_t dtype, int64_t m, int64_t n,
                                                   int64_t k, int64_t lda, int64_t ldb,
                                                   int64_t ldc, int64_t stride_a,
                                                   int64_t stride_b, int64_t gemm_flags,
                                                   int64_t unary_flags, int64_t unary_kind,
                                                   int64_t binary_flags, int64_t binary_kind);
/* replaces XsmmRunnerUtils.cpp:142-179 */
TPP_XSMM_EXPORT int64_t xsmm_unary_dispatch(int64_t unary_kind, int64_t dtype, int64_t m,
                                            int64_t n, int64_t ldi, int64_t ldo,
                                            int64_t flags);
/* replaces XsmmRunnerUtils.cpp:181-211 */
TPP_XSMM_EXPORT int64_t xsmm_binary_dispatch(int64_t binary_kind, int64_t dtype, int64_t m,
                                             int64_t n, int64_t ldi_lhs, int64_t ldi_rhs,
                                             int64_t ldo, int64_t flags);
/* replaces XsmmRunnerUtils.cpp:213-246 (Intel AMX only: a no-op here) */
TPP_XSMM_EXPORT int64_t xsmm_intel_amx_tile_config_dispatch(int64_t dtype, int64_t m,
                                                            int64_t n, int64_t k,
                                                            int64_t lda, int64_t ldb,
                                                            int64_t ldc, int64_t stride_a,
                                                            int64_t stride_b, int64_t flags);

/* ---- invoke: run a dispatched kernel; synchronous unless async mode is on ---- */

/* replaces XsmmRunnerUtils.cpp:79-93 */
TPP_XSMM_EXPORT void xsmm_gemm_invoke(int64_t dtype, int64_t handle, void *a, int64_t off_a,
                                      void *b, int64_t off_b, void *c, int64_t off_c);
/* replaces XsmmRunnerUtils.cpp:288-306 */
TPP_XSMM_EXPORT void xsmm_brgemm_invoke(int64_t dtype, int64_t handle, void *a,
                                        int64_t off_a, void *b, int64_t off_b, void *c,
                                        int64_t off_c, int64_t num_batches);
/* replaces XsmmRunnerUtils.cpp:363-383 */
TPP_XSMM_EXPORT void xsmm_fused_brgemm_invoke(int64_t dtype, int64_t handle, void *a,
                                              int64_t off_a, void *b, int64_t off_b, void *c,
                                              int64_t off_c, void *d, int64_t off_d,
                                              int64_t num_batches);
/* replaces XsmmRunnerUtils.cpp:248-259 */
TPP_XSMM_EXPORT void xsmm_unary_invoke(int64_t dtype, int64_t handle, void *in,
                                       int64_t off_in, void *out, int64_t off_out);
/* replaces XsmmRunnerUtils.cpp:276-286 (the scalar is ALWAYS an f32) */
TPP_XSMM_EXPORT void xsmm_unary_scalar_invoke(int64_t dtype, int64_t handle, float scalar,
                                              void *out, int64_t off_out);
/* replaces XsmmRunnerUtils.cpp:261-274 */
TPP_XSMM_EXPORT void xsmm_binary_invoke(int64_t dtype, int64_t handle, void *lhs,
                                        int64_t off_lhs, void *rhs, int64_t off_rhs,
                                        void *out, int64_t off_out);
/* replaces XsmmRunnerUtils.cpp:459-469 (no-op) */
TPP_XSMM_EXPORT void xsmm_intel_amx_tile_config_invoke(int64_t dtype, int64_t handle,
                                                       void *tile_state, int64_t off);

/* ---- timers: runtime/PerfRunnerUtils.cpp:23-35. perf_stop_timer drains the
 * device queue before reading the clock so that async-mode launches are counted. */
TPP_XSMM_EXPORT int64_t perf_start_timer(void);
TPP_XSMM_EXPORT double perf_stop_timer(int64_t start);

/* ---- extensions (NOT in the reference; the reference ABI has no config call) --
 * The JIT'd code never calls these; harnesses (bench.py, tests, tpp_replay) do. */

/* 0 (default): every invoke returns after its kernel completed (reference
 * semantics). 1: invokes only enqueue on the runtime's stream; call
 * xsmm_hip_synchronize() or perf_stop_timer() to drain. Returns previous mode.
 * Also settable with env TPP_HIP_ASYNC=1.
 * BUFFER LIFETIME in async mode: operands may be freed / re-allocated only after
 * xsmm_hip_synchronize(), perf_stop_timer() or xsmm_hip_set_async(0) returned -
 * those calls launch whatever the tile queue still holds, drain the stream and
 * forget the device allocation ranges the runtime has cached since the last such
 * call. Synchronising the device some other way (hipDeviceSynchronize,
 * torch.cuda.synchronize) leaves queued invokes unlaunched and the cache in place. */
TPP_XSMM_EXPORT int xsmm_hip_set_async(int enable);
/* Tile queue (async mode only, device pointers only): invokes of ONE small-tile handle - GEMM
 * family, unary or binary, m, n <= 64: the compiler's native 32x32x32 call pattern and its per-block
 * pack / unpack tiles - are collected and run as ONE grouped launch at the next flush point (other
 * handle / other op / data dependence on a queued output or overwrite of a queued input /
 * xsmm_hip_flush / xsmm_hip_synchronize / perf_stop_timer). Program order is preserved. Operands
 * of queued invokes must stay valid until the flush. Returns the previous setting. Also env
 * TPP_HIP_TILE_QUEUE=1. The queue serves ONE device per process (the device current on the first
 * queued invoke): a queued invoke from a thread whose current device differs is a fatal error.
 * Several calling threads (the reference's OpenMP team over a layer's tile grid): a group that was collected once is
 * REPLAYED when its invokes come again - each caller marks its own invokes in the recorded group without taking a lock, and
 * the complete group launches from a work list that already sits in device memory. Programs that do not repeat themselves
 * are handed to a scheduler thread after a few thousand locked arrivals; enable = 2 (TPP_HIP_TILE_QUEUE=2) does that as soon
 * as a second thread shows up (the round-2 behaviour, kept for comparison runs). */
TPP_XSMM_EXPORT int xsmm_hip_set_tile_queue(int enable);
TPP_XSMM_EXPORT void xsmm_hip_flush(void);
/* n fused_brgemm invokes in ONE call (arrays of length n, one entry per call): exactly the effect of
 *   for (i = 0; i < n; ++i) xsmm_fused_brgemm_invoke(dtype, handles[i], a[i], off_a[i], b[i], off_b[i], c[i], off_c[i],
 *                                                    d[i], off_d[i], num_batches[i]);
 * (XsmmRunnerUtils.cpp:363-457 per call). This is how a harness hands over a rank's MLP step (mlir-gen's layer chain,
 * MLIRGen.cpp:632-681: every layer one whole-layer fused_brgemm): when the calls form a CHAIN - call i+1 reads call i's
 * output as its A operand (same pointer, lda = ldc), bf16 with ONE kind of B operand (VNNI-2, flat or VNNI-4), beta 0, equal m and n, device pointers,
 * asynchronous mode, the outputs overlap no other operand, and one of the 32x64 .. 128x128 tiles covers m x n with at most
 * one workgroup per compute unit - the whole chain runs as ONE persistent kernel (rows of layer i+1 start as soon as the
 * same rows of layer i are stored: no kernel boundary, the next layer's weight panels are prefetched under the epilogue).
 * Same arithmetic as the separate launches (f32 accumulation in k order, one rounding per layer); bit-identical to them when
 * they run on the same tile, which is the case whenever dispatch planned the layers with a loader-wave tile (variants 20 .. 23)
 * that fits the chip. Returns 1 if the chain ran as one launch, 0 if it ran call by call.
 * f32 (round 4): chains of whole-layer f32 calls (no VNNI operand, beta 0, 16-byte aligned operands and bias) run as one launch when
 * EVERY call was planned on the same 64-row K-split loader-wave tile (64x64 + K2 or 64x32 + K4: 512 .. 1024 rows of a 1024-wide
 * layer) and that tile fits the chip - bit-identical to the calls, 1-2.5 % faster; 32-row tiles (256 rows) stay call by call, which
 * measured faster there (profiles/r04_f32_chain.txt).
 * Call i > 0 must keep every batch element inside its predecessor's rows ((br - 1) * stride_a + k <= lda): the hand-off is per row
 * block; a row-striding later call runs call by call.
 * RESIDENCY: the single launch needs every workgroup of its grid on a compute unit at the same time (one per CU). The grid is
 * checked against the CUs the stream may use (a CU mask is honoured); what cannot be checked is another process - or another
 * stream's LDS-heavy kernel - occupying CUs at that moment: the single-launch path needs the device to itself. Every wait inside
 * the kernel is bounded (50 ms): a starved launch never hangs, and since round 5 it does not end the process either - the library
 * DEGRADES (the reference never aborts on a valid invoke): the FIRST single launch of every (stream, tile grid, layer count) is
 * followed by a stream synchronisation and a look at its error word - a device that is shared from the start is found out before
 * anyone could consume the launch's outputs, the call runs call by call at once (return value 0; the inputs are intact: beta 0,
 * the outputs overlap no operand); later launches stay asynchronous and their calls are journaled - if the check at the next
 * xsmm_hip_synchronize / perf_stop_timer finds a starved launch, the journaled calls are re-run call by call, in launch order,
 * before that call returns. Either way one line goes to stderr and every later chain invoke of the process runs call by call.
 * (What the re-run cannot repair: work that OTHERS enqueued between a starved asynchronous launch and the synchronisation has read
 * invalid outputs.) A harness that knows it shares the GPU sets TPP_HIP_CHAIN=0 and spares itself the timeouts.
 * TPP_HIP_CHAIN=0 disables the single-launch path. */
TPP_XSMM_EXPORT int xsmm_hip_fused_brgemm_chain_invoke(int64_t dtype, int64_t n, const int64_t *handles, void *const *a,
                                                       const int64_t *off_a, void *const *b, const int64_t *off_b,
                                                       void *const *c, const int64_t *off_c, void *const *d,
                                                       const int64_t *off_d, const int64_t *num_batches);
/* STRICT mode (env TPP_HIP_STRICT=1): the kernel an invoke runs on - and with it the order of its floating-point additions - is a
 * function of its descriptor, its batch count and the alignment of its OWN pointers only (libxsmm's JIT'd kernel is a function of
 * the dispatch tuple: XsmmRunnerUtils.cpp:288-306). Off (default), the tile queue lets the GROUP choose: tile grids over flat operands
 * are merged into one launch, transposes are folded into the gemm they feed, the kernel family and the split follow the size of the
 * queued group - all within the 1e-5 bar, but the same invoke on the same data may return different last bits alone, in the first
 * pass of a loop and in its replays. On: no grid merge, no folded transposes, every size-dependent choice is taken as for a group of
 * one, a group holds invokes of one alignment class and one batch count only, single invokes of queue-sized tiles run on the kernel
 * their group runs on, and a chain call runs as one launch only on the tile its layers were planned on. DETERMINISM: run to run
 * always (no floating-point atomics anywhere); call to call - alone / first queued pass / replay - under strict mode
 * (tests/test_strict_gpu.py). Choose the mode before the first queued invoke: returns the previous setting, -1 if the tile queue has
 * already recorded groups under the other one. */
TPP_XSMM_EXPORT int xsmm_hip_set_strict(int enable);
TPP_XSMM_EXPORT int xsmm_hip_get_strict(void);
/* The launch thread (round 6; TPP_HIP_LAUNCH_THREAD=0|1, default 1; tile queue + asynchronous mode only). A recorded group that is
 * replayed completely is launched by a helper thread of the runtime instead of the calling thread: the caller goes on queueing the
 * next group while hipLaunchKernel (2.3-2.6 us of host time) runs beside it. Stream order is unchanged - the hand-overs leave in
 * order, and everything else that launches / copies / synchronises first waits until they have left. The thread spins while
 * launches keep coming, sleeps after ~0.2 ms without one and ends after ~2 s. Kernel choice and results do not depend on the setting.
 * xsmm_hip_set_launch_thread returns the previous setting; stats: out[0] = launches handed over since process start, out[1] = 1 if
 * the thread exists right now. */
TPP_XSMM_EXPORT int xsmm_hip_set_launch_thread(int enable);
TPP_XSMM_EXPORT void xsmm_hip_launch_thread_stats(int64_t out[2]);
/* Sticky status of the chain launches: the number of journaled launches that were found starved at a synchronisation point and
 * re-run call by call since process start (0: never). Round 6: every journaled launch has its own error word, so only the starved
 * launch and the later ones of ITS stream are re-run (healthy earlier launches are left alone), the journal is kept per stream, and
 * a launching thread that finds the pool of error words nearly used up synchronises and checks instead of dropping an entry.
 * TPP_HIP_CHAIN_STRICT=1: a starved launch ends the process (stderr + exit(-1)) instead of being repaired. */
TPP_XSMM_EXPORT int64_t xsmm_hip_chain_status(void);
/* Peer-store all-gather (one process per GPU, no collective library call): every rank stores its row block straight into
 * every peer's output buffer through IPC-mapped pointers, completion by flags (csrc/peer_gather.hip; host-side protocol:
 * tpp-mlir_amd/peer.py). _peer_alloc returns a dedicated zeroed device allocation (IPC handles name whole allocations);
 * _ipc_export writes its 64-byte handle; _ipc_open maps a peer's handle (nullptr on failure); _peer_gather enqueues one step on
 * the runtime's stream: scatter of `bytes` from src to dst[w] + dst_offset for every w < world, then a wait until every peer's
 * block has landed in THIS rank's buffer. dst / flags / ready: `world` pointers each; epoch: 1, 2, 3, ... (the same on every rank). */
TPP_XSMM_EXPORT void *xsmm_hip_peer_alloc(int64_t bytes);
TPP_XSMM_EXPORT void xsmm_hip_peer_free(void *ptr);
TPP_XSMM_EXPORT int xsmm_hip_ipc_export(void *ptr, void *handle_out_64_bytes);
TPP_XSMM_EXPORT void *xsmm_hip_ipc_open(const void *handle_64_bytes);
TPP_XSMM_EXPORT int xsmm_hip_ipc_close(void *ptr);
TPP_XSMM_EXPORT void xsmm_hip_peer_gather(const void *src, int64_t bytes, int64_t dst_offset, int64_t world, int64_t rank,
                                          void *const *dst, void *const *flags, void *const *ready, void *my_flags, void *my_ready,
                                          void *ticket, void *err, int64_t epoch);
/* OVERLAP mode of the peer gather (0 / 1, returns the previous setting): the wait kernel goes to a side stream, so the transfer
 * time of a step's blocks no longer sits between this step's kernels and the next step's on the runtime's stream. The gathered
 * output of a step is then complete when the SIDE stream has passed that step's wait kernel: xsmm_hip_peer_drain() blocks the
 * host until it has; xsmm_hip_peer_wait_stream() returns the stream (hipStream_t) for an event-based dependence. A consumer of
 * the output of step e must run before this rank enqueues the gather of step e+1 on the runtime's stream (the peers take "rank r
 * has entered gather e+1" as the licence to overwrite the buffer of step e at step e+2, exactly as without the overlap). */
TPP_XSMM_EXPORT int xsmm_hip_peer_overlap(int enable);
TPP_XSMM_EXPORT void *xsmm_hip_peer_wait_stream(void);
TPP_XSMM_EXPORT void xsmm_hip_peer_drain(void);
/* counters of the tile queue since process start: out[0] grouped launches, out[1] invokes queued with the full
 * dependence bookkeeping, out[2] invokes queued by replay of a recorded group (trace cache), out[3] groups ended by
 * a remembered terminator, out[4] replays abandoned (the caller left the recorded group) */
TPP_XSMM_EXPORT void xsmm_hip_tile_queue_stats(int64_t out[5]);
/* Stream the kernels are launched on (a hipStream_t). NULL = default stream. */
TPP_XSMM_EXPORT void xsmm_hip_set_stream(void *hip_stream);
TPP_XSMM_EXPORT void *xsmm_hip_get_stream(void);
TPP_XSMM_EXPORT void xsmm_hip_synchronize(void);
/* Host residents: the ABI has no allocation hook, so the runtime never caches a mirror of a host buffer
 * behind the caller's back. A harness that knows a host buffer is long-lived (weights, the operands of a
 * timing loop: lib/TPP/Runner/MLIRBench.cpp:207-246 allocates them once) can declare it: the buffer is
 * uploaded once, and invokes whose operands lie inside it run on the device copy without any upload
 * (operands the kernel writes are still copied back, so the host view stays current). _update re-uploads
 * after the host changed the buffer; _release drops the copy. Return 0, or -1 (unknown / overlapping range). */
TPP_XSMM_EXPORT int xsmm_hip_host_resident(const void *ptr, int64_t bytes);
TPP_XSMM_EXPORT int xsmm_hip_host_update(const void *ptr);
TPP_XSMM_EXPORT int xsmm_hip_host_release(const void *ptr);
/* Host cache (round 6; csrc/host_cache.h): the AUTOMATIC form of the host residents - for an unmodified harness that hands host
 * pointers to every invoke (memref globals / malloc: lib/TPP/Runner/MLIRBench.cpp:207-246) and can be given environment variables only.
 * enable = 1 (env TPP_HIP_HOST_CACHE=1): a host operand gets a device mirror that OUTLIVES the invoke; before a kernel uses mirror
 * pages, the pages the host has written since the runtime last looked are uploaded again - and only those (the kernel's own write
 * tracking says which: userfaultfd asynchronous write-protect + PAGEMAP_SCAN, Linux >= 6.7; no fault handler, no signal, no helper
 * thread; system calls that write into tracked memory work as always). What kernels write goes back to exactly the host bytes they
 * wrote: in synchronous mode before the invoke returns - the reference's contract (SURVEY.md 8b "Completion"), unchanged -, in
 * asynchronous mode at the next synchronisation point (xsmm_hip_synchronize / perf_stop_timer / xsmm_hip_set_async(0) /
 * xsmm_hip_set_stream). ASYNCHRONOUS MODE CONTRACT with host operands: the lifetime rule above, and between two synchronisation
 * points the host neither reads outputs nor writes operands of the invokes in between (the runtime looks at an extent once per
 * synchronisation epoch; the tile queue then sees device pointers, so TPP_HIP_ASYNC=1 TPP_HIP_TILE_QUEUE=1 TPP_HIP_HOST_CACHE=1 runs
 * the compiler's tile invokes on host buffers as grouped launches). A range that was unmapped and mapped again, or cannot be tracked
 * (file-backed / shared mappings), falls back to the plain per-invoke mirror. Returns the previous setting, or -1 if the kernel
 * interface is missing (the cache stays off). Switching it off writes everything back and frees the mirrors.
 * _stats: out[0] extents, [1] mirror bytes, [2] bytes uploaded, [3] page-table scans, [4] bytes written back, [5] pages NOT written back
 * (their range changed hands or the host wrote them while a device write was pending), [6] extents created / grown / merged,
 * [7] invokes translated on the lock-free path, [8] on the locked path, [9] extents given up. */
TPP_XSMM_EXPORT int xsmm_hip_set_host_cache(int enable);
TPP_XSMM_EXPORT void xsmm_hip_host_cache_stats(int64_t out[10]);
/* Number of visible HIP devices (0 on a CPU-only host; never exits). */
TPP_XSMM_EXPORT int xsmm_hip_device_count(void);
/* Name of the HIP kernel variant a GEMM-like handle selected, for profiles. */
TPP_XSMM_EXPORT const char *xsmm_hip_kernel_name(int64_t handle);
/* Name of the kernel family the tile queue's most recent grouped GEMM launch ran on ("" before the first): a group of queued tile
 * invokes may run on a faster family than a single invoke of its handle would (xsmm_hip_kernel_name), e.g. 32-k f32 tiles with even
 * batch counts on the loader-wave kernels. A static string; diagnostics only. */
TPP_XSMM_EXPORT const char *xsmm_hip_last_grouped_kernel(void);
/* The kernel of the most recent NON-queued gemm / brgemm / fused_brgemm invoke when it was refined at invoke time - the batch count
 * arrives with the invoke, so two choices are made there: f32 outputs with fewer tiles than CUs and a long reduction run with the
 * batch-reduce range of a tile split over several workgroups (xsmm_hip_force_split), small bf16 outputs with K >= 1536 on the 32x64
 * loader-wave tile. "" = the kernel xsmm_hip_kernel_name(handle) names ran. For tests, tools and profiles. */
TPP_XSMM_EXPORT const char *xsmm_hip_last_refined_kernel(void);
/* Force a GEMM tile variant for A/B benchmarking and tests (-1 = automatic): f32 0..4 (64x64,
 * 64x32+K2, 32x32+K4, 128x64, 64x64+K2), 5..7 the loader-wave kernels (64x64, 64x64+K2, 64x32+K4), 8 generic, 9 / 10 loader-wave 32x32+K4 / 128x64, bf16 16 / 17 / 18 / 19 (64x64, 128x128, 256x256, 32x32 + K split),
 * 20 .. 23 the bf16 loader-wave tiles for mid-size outputs (32x64 + K split, 64x64, 64x128, 128x128).
 * Honoured at dispatch when the shape divides the tile;
 * 24 .. 27 the same tiles for a flat bf16 B operand, 28 .. 31 for a VNNI-4 B operand. */
TPP_XSMM_EXPORT void xsmm_hip_force_variant(int variant);
/* Skinny f32 outputs (fewer output tiles than compute units, a long batch-reduce: the reference's M = 128 / 256 benchmark shapes,
 * benchmarks/config/matmul/128x1024x4096.json ...) run with the batch-reduce range of ONE output tile split over several workgroups:
 * each adds its chunks, the workgroup that finishes last sums the partial tiles in split order (a fixed order of additions: the same
 * call pattern gives the same bits every run; no float atomics). How many workgroups share a tile is chosen per launch from the
 * descriptor, the batch count and the number of tiles in the launch; this call overrides it for the launches that FOLLOW (a test /
 * measurement switch): -1 = the model (default), 0 or 1 = never split, n > 1 = n workgroups per tile (at most 16 and at most the
 * number of 64-k chunks). Also TPP_HIP_SPLIT. Returns the previous setting. */
TPP_XSMM_EXPORT int xsmm_hip_force_split(int workgroups_per_tile);
/* Transposes folded into the gemm they feed (tile queue on, asynchronous mode, device operands, f32). A contraction with a
 * transposed B operand reaches the runtime as xsmm.unary transpose into a small temporary + xsmm.gemm reading it, per tile and with
 * ONE temporary per caller (test/Conversion/LinalgToXsmm/linalg-to-gemm.mlir:46-62, the lowering of
 * benchmarks/mlir/fp32-query-times-key.mlir): a dependence chain through the temporary that no queue can batch. So a transpose of a
 * tile of at most 64x64 into a dense destination (ldo = m) is remembered instead of launched; a gemm of the same thread whose B
 * operand is exactly that destination (k, n, ldb matching, one batch element) reads B transposed from the transpose's SOURCE (all
 * such gemms of a loop are one queue group = one launch); a later transpose of the same handle into the same destination replaces
 * the remembered one (it is dead: fully overwritten, its readers were served); any other invoke of the thread, xsmm_hip_flush and
 * every synchronisation point launch the remembered transpose first - whenever anything can look at the destination it holds what
 * the program wrote. Results of a folded gemm are those of the generic kernel on the same values. One remembered transpose per
 * calling thread (the reference's OpenMP callers own a temporary each); an invoke of another thread launches it first only if it
 * touches the destination or writes the source. 0 turns it off (also TPP_HIP_FOLD_TRANSPOSE=0); returns the previous setting.
 * stats: [0] gemm invokes served from a transpose's source, [1] remembered transposes dropped as dead, [2] launched after all.
 *  Between a transpose invoke and the gemm invoke that consumes its temporary, no OTHER thread may write the transpose's
 * source: the folded gemm reads the source when it is enqueued, not when the transpose was invoked. */
TPP_XSMM_EXPORT int xsmm_hip_set_fold_transpose(int enable);
TPP_XSMM_EXPORT void xsmm_hip_fold_transpose_stats(int64_t out[3]);
/* The VNNI blocking factor v of bf16 B operands ([k/v][ldb][v]) of gemm / brgemm / fused_brgemm handles dispatched FROM NOW ON with
 * the VNNI_B wire flag: 2 (default) or 4; also TPP_HIP_VNNI_FACTOR. The factor is not on the wire - the reference's compiler and its
 * runtime library both ask libxsmm_cpuid_dot_pack_factor(LIBXSMM_DATATYPE_BF16) (lib/TPP/Transforms/Utils/VNNIUtils.cpp:25-45; the
 * `--vnni=4` rows of benchmarks/config/omp/mlir-bf16.json:68-100): a harness that lowers with vnni = 4 sets 4 here before it
 * dispatches. k must then be a multiple of 4 (dispatch dies otherwise); ldb is the k-group row stride / v as the compiler passes it
 * (ConvertLinalgToXsmm.cpp:1144). xsmm.unary VNNI2 (kind 28) always packs pairs - there is no VNNI-4 pack kind at this revision
 * (XsmmEnum.td:34-45). set returns the previous factor, -1 for an invalid one. The setting is process-wide and read once per
 * dispatch: set it BEFORE dispatching (a handle keeps the factor it was dispatched with); a VNNI A operand (wire flag 4096) is
 * checked against the same factor (k a multiple of it; [m][k/v][v] is byte-identical to the flat row). */
TPP_XSMM_EXPORT int xsmm_hip_set_vnni_factor(int factor);
TPP_XSMM_EXPORT int xsmm_hip_get_vnni_factor(void);
/* Library version string. */
TPP_XSMM_EXPORT const char *xsmm_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TPP_XSMM_ABI_H */
