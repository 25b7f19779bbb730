// rt_chain.h - chains of whole-layer fused BRGEMMs in one launch: blocks, probation, journal with one error word per launch, re-run
// One of the subsystem units of runtime.cpp (round 6, VERDICT r5 next 7: the 3 000-line file split by subsystem, no behaviour
// change). The units are INCLUDED into the one translation unit runtime.cpp, in dependence order, inside its anonymous namespace:
// the per-invoke host path (14-18 ns: enqueue_item -> join_window -> Segment::mark) crosses four of them and is inlined across
// their borders - as separate objects without LTO it would pay a call per border. Not a stand-alone header: include runtime.cpp's way only.

// ---- chains of whole-layer fused BRGEMMs in one launch (xsmm_hip_fused_brgemm_chain_invoke) -----------------------------
// Hand-off state of the chain kernel (brgemm_bf16_lw.hip, chain mode): arrival counters that only grow - a launch adds
// tiles_n to each, its target is epoch * tiles_n - so a block of counters is tied to ONE (stream, tile grid, layer count):
// launches of one block are ordered by their stream and issued under the mutex (epoch order = stream order). The err word
// lives in pinned host memory: a workgroup whose wait timed out writes it over PCIe, the host reads it at its sync points.
struct ChainBlock {
  hipStream_t stream;
  int tiles_m, tiles_n, nlayers;
  unsigned *cnt; // device: (CH_MAXL - 1) * tiles_m counters, CHAIN_CNT_STRIDE words apart
  unsigned *err; // pinned host
  unsigned epoch;
  int verified;  // launches of this block that were checked synchronously and had every hand-off succeed (probation, see below)
};
// A STARVED chain launch (another process, or another stream's LDS-heavy kernel, held compute units while it ran: not every
// workgroup became resident, a consumer's bounded wait ran out, the error word is set) used to end the process. Round 5 (VERDICT r4
// item 5): the library degrades instead - the reference never aborts on a valid invoke (XsmmRunnerUtils.cpp:363-383).
//  * PROBATION: the first launch of every block (stream, tile grid, layer count) is followed by a stream synchronisation and a look
//    at the error word. A device that is shared when the harness starts is found out here, before anything could consume the
//    launch's outputs: the chain call then runs call by call at once (its inputs are intact: beta 0, outputs overlap no operand),
//    and the process remembers that the device is shared - every later chain invoke runs call by call (TPP_HIP_CHAIN=0 behaviour).
//  * LATER launches stay asynchronous; the calls of every launch since the last check are kept in a journal (the last launch per
//    set of output pointers). If the check at a synchronisation point finds the error word set, the journal is re-run call by call
//    in launch order before the synchronisation returns: what the caller then reads is what the calls compute from the operands as
//    they are now. (Work that OTHERS enqueued between a starved launch and the synchronisation has read invalid outputs - the
//    stderr line says so; the chain contract of include/tpp_xsmm_abi.h asks for the device to oneself for this reason.)
// Round 6: every journaled launch has its OWN error word (a pool of pinned words), so the check knows WHICH launch
// starved: only that launch and the later ones of the same stream are re-run (a healthy earlier launch whose inputs have since
// been overwritten is left alone); the journal is looked at per stream - the one that has just been drained - and the entries of
// other streams stay; when the pool runs dry the launching thread synchronises and checks instead of dropping entries; the re-run
// goes to the launch's stream through a thread-local override (the process-wide stream setting is not touched);
// xsmm_hip_chain_status() counts the repairs, TPP_HIP_CHAIN_STRICT=1 keeps fail-stop.
struct ChainCall {
  int n;
  int64_t dtype;
  int64_t handle[CH_MAXL];
  void *a[CH_MAXL], *b[CH_MAXL], *c[CH_MAXL], *d[CH_MAXL];
  int64_t br[CH_MAXL];
  hipStream_t stream;
  unsigned *err; // this launch's own error word (pinned host memory, from g_chain_err_free)
};
std::vector<ChainCall> g_chain_journal; // under g_chain_mu, in launch order
std::vector<unsigned *> g_chain_err_free; // under g_chain_mu
constexpr int CHAIN_ERR_POOL = 512;
std::atomic<int> g_chain_journaled{0};  // entries in the journal (read without the lock: "is the pool about to run dry")
std::atomic<int64_t> g_chain_repairs{0}; // starved launches found and re-run since process start (xsmm_hip_chain_status)
std::atomic<bool> g_chain_shared{false}; // a chain launch was starved once: no more chain launches in this process
std::mutex g_chain_mu;

std::vector<ChainBlock> g_chain_blocks;
std::atomic<int> g_chain_launched{0}; // chain launches since the last check of the err words

ChainBlock &chain_block(hipStream_t s, int tiles_m, int tiles_n, int nlayers) { // under g_chain_mu
  for (ChainBlock &b : g_chain_blocks)
    if (b.stream == s && b.tiles_m == tiles_m && b.tiles_n == tiles_n && b.nlayers == nlayers) return b;
  ChainBlock b{s, tiles_m, tiles_n, nlayers, nullptr, nullptr, 0, 0};
  const size_t bytes = sizeof(unsigned) * (size_t)(CH_MAXL - 1) * (size_t)tiles_m * CHAIN_CNT_STRIDE;
  HIP_OK(hipMalloc((void **)&b.cnt, bytes));
  HIP_OK(hipMemset(b.cnt, 0, bytes));
  HIP_OK(hipHostMalloc((void **)&b.err, sizeof(unsigned), hipHostMallocDefault));
  *b.err = 0;
  g_chain_blocks.push_back(b);
  return g_chain_blocks.back();
}
// after stream `s` has been drained: did a hand-off of a chain launch on it time out?
void dump_chain_stamps();
void chain_rerun_call_by_call(const ChainCall &c);
void check_chain_errors(hipStream_t s) {
  if (!g_chain_launched.load(std::memory_order_acquire)) return;
  std::vector<ChainCall> redo;
  unsigned layer = 0;
  {
    std::lock_guard<std::mutex> lk(g_chain_mu);
    dump_chain_stamps();
    std::vector<ChainCall> keep;
    for (const ChainCall &c : g_chain_journal) {
      if (c.stream != s) { // another stream's launch: not drained by this synchronisation, stays
        keep.push_back(c);
        continue;
      }
      const unsigned e = *(volatile unsigned *)c.err;
      if (e && !layer) layer = e;
      if (layer) redo.push_back(c); // the first starved launch of this stream and every later one (they may have consumed its outputs)
      *(volatile unsigned *)c.err = 0;
      g_chain_err_free.push_back(c.err);
    }
    g_chain_journal.swap(keep);
    g_chain_journaled.store((int)g_chain_journal.size(), std::memory_order_relaxed);
    if (g_chain_journal.empty()) g_chain_launched.store(0, std::memory_order_release);
  }
  if (!layer) return;
  static const bool strict = [] { const char *e = getenv("TPP_HIP_CHAIN_STRICT"); return e && atoi(e) != 0; }();
  if (strict)
    die("tpp-xsmm-hip: a fused-brgemm chain launch was starved (a hand-off for layer %u's input timed out: not every workgroup was resident - "
        "the device is shared) and TPP_HIP_CHAIN_STRICT=1 asks for fail-stop", layer - 1);
  // starved: the device is shared. The starved launch and the later ones of its stream run again, call by call, in launch order;
  // chains are off from now on.
  g_chain_shared.store(true, std::memory_order_release);
  g_chain_repairs.fetch_add((int64_t)redo.size(), std::memory_order_relaxed);
  fprintf(stderr, "[tpp-xsmm-hip] a fused-brgemm chain launch was starved (a hand-off for layer %u's input timed out: not every workgroup "
                  "was resident - the device is shared); that launch and the %zu later one(s) of its stream are re-run call by call now "
                  "(earlier launches completed and are left alone), and chain invokes run call by call from here on "
                  "(xsmm_hip_chain_status() counts; TPP_HIP_CHAIN_STRICT=1 ends the process instead). Work that others enqueued behind a "
                  "starved launch has read invalid data.\n",
          layer - 1, redo.size() - 1);
  for (const ChainCall &c : redo) chain_rerun_call_by_call(c);
  HIP_OK(hipStreamSynchronize(s));
}
void check_chain_errors() { check_chain_errors(cfg().stream.load()); }

int chip_cus() { // compute units of the current device (0: unknown)
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
// Compute units a launch on `s` can actually use: the device's, restricted by the stream's CU mask (hipExtStreamCreateWithCUMask,
// or the process-wide ROC_GLOBAL_CU_MASK / HSA_CU_MASK the runtime folds into every stream's mask). The persistent chain kernel
// needs all of its workgroups resident at once - one per CU - so its grid is checked against THIS number. What no query
// can see is another PROCESS (or another stream's LDS-heavy kernel) holding CUs at launch time: the chain launch needs the device
// to itself (include/tpp_xsmm_abi.h says so); every spin in the kernel is bounded and a starved launch is reported, not hung.
int stream_cus(hipStream_t s) {
  const int all = chip_cus();
  uint32_t mask[16] = {};
  if (all <= 0 || hipExtStreamGetCUMask(s, 16, mask) != hipSuccess) {
    (void)hipGetLastError();
    return all;
  }
  int bits = 0;
  for (uint32_t w : mask) bits += __builtin_popcount(w);
  return bits > 0 && bits < all ? bits : all;
}

// profiling (-DTPP_HIP_ABLATION side builds only, build.py --ablation): TPP_HIP_CHAIN_STAMPS=<file> makes every chain launch record s_memrealtime stamps (100 MHz) per workgroup and layer
// (see blw_stamp in brgemm_bf16_lw.hip) into pinned host memory; the LAST launch's stamps are written to the file at every sync point.
unsigned long long *g_stamps = nullptr;
size_t g_stamps_wgs = 0;
unsigned long long *chain_stamps(size_t wgs) { // under g_chain_mu
#ifdef TPP_HIP_ABLATION
  static const char *path = getenv("TPP_HIP_CHAIN_STAMPS");
  if (!path) return nullptr;
  if (!g_stamps) HIP_OK(hipHostMalloc((void **)&g_stamps, sizeof(unsigned long long) * 8 * CH_MAXL * 1024, hipHostMallocDefault));
  if (wgs > 1024) return nullptr;
  g_stamps_wgs = wgs;
  return g_stamps;
#else
  (void)wgs;
  return nullptr; // the shipped kernels carry no stamp code (brgemm_bf16_lw.hip: blw_stamp)
#endif
}
void dump_chain_stamps() {
#ifdef TPP_HIP_ABLATION
  const char *path = getenv("TPP_HIP_CHAIN_STAMPS");
#else
  const char *path = nullptr;
#endif
  if (!path || !g_stamps || !g_stamps_wgs) return;
  if (FILE *f = fopen(path, "w")) {
    for (size_t w = 0; w < g_stamps_wgs; ++w)
      for (int l = 0; l < CH_MAXL; ++l) {
        const unsigned long long *s = g_stamps + (w * CH_MAXL + l) * 8;
        if (!s[0] && !s[5]) continue;
        fprintf(f, "%zu %d", w, l);
        for (int i = 0; i < 8; ++i) fprintf(f, " %llu", s[i]);
        fputc('\n', f);
      }
    fclose(f);
  }
  // the loaders' per-chunk records of the first 16 workgroups (TPP_HIP_CHAIN_DBG & 1024; brgemm_bf16_lw.hip BlwChunkStamps)
  if (chain_ablation_bits() & 1024) {
    const std::string p2 = std::string(path) + ".chunks";
    if (FILE *f = fopen(p2.c_str(), "w")) {
      const unsigned long long *base = g_stamps + g_stamps_wgs * CH_MAXL * 8;
      for (int w = 0; w < 16 && (size_t)w < g_stamps_wgs; ++w)
        for (int which = 0; which < 2; ++which) {
          const unsigned long long *r = base + ((size_t)w * 2 + which) * (64 * 3 + 1);
          const int n = (int)(r[0] > 64 ? 64 : r[0]);
          for (int i = 0; i < n; ++i) fprintf(f, "%d %d %d %llu %llu %llu\n", w, which, i, r[1 + 3 * i], r[2 + 3 * i], r[3 + 3 * i]);
        }
      fclose(f);
    }
  }
}

bool ranges_overlap(const void *a, size_t na, const void *b, size_t nb) {
  return (const char *)a < (const char *)b + nb && (const char *)b < (const char *)a + na;
}

// true: the chain was launched as ONE kernel. false: the caller runs the invokes one by one (same result).
bool try_chain_launch(int n, const GemmDesc *const *d, void *const *pa, void *const *pb, void *const *pc, void *const *pd, const int64_t *br,
                      hipStream_t s) {
  // with TPP_HIP_TRACE >= 1 the reason for running call by call goes to stderr
#define NOCHAIN(why)                                                                             \
  do {                                                                                           \
    if (cfg().trace) fprintf(stderr, "[tpp-xsmm-hip] fused_brgemm_chain: call by call (%s)\n", why); \
    return false;                                                                                \
  } while (0)
  if (n < 2 || n > CH_MAXL) NOCHAIN("fewer than 2 or more than 8 calls");
  if (!cfg().async.load(std::memory_order_relaxed)) NOCHAIN("synchronous mode");
  {
    // a launch's hand-off target (epoch x tiles per row block) is baked into its arguments: replayed from a graph it would be
    // stale - the consumers would not wait. Captured streams get the separate launches.
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(cfg().stream.load(std::memory_order_relaxed), &cs) != hipSuccess) (void)hipGetLastError();
    else if (cs != hipStreamCaptureStatusNone) NOCHAIN("the stream is being captured into a graph");
  }
  static const int enabled = [] {
    const char *e = getenv("TPP_HIP_CHAIN");
    return e ? atoi(e) : 1;
  }();
  if (!enabled) NOCHAIN("TPP_HIP_CHAIN=0");
  if (g_chain_shared.load(std::memory_order_acquire)) NOCHAIN("an earlier chain launch was starved: the device is shared");
  const int64_t m = d[0]->m, nn = d[0]->n;
  thread_local DeviceRanges devmem;
  devmem.refresh();
  // f32 chains (round 4): every call planned on the SAME K-split loader-wave tile (the launch is then bit-identical to the calls)
  const bool f32 = d[0]->dtype == DT_F32;
  const int f32_tile = f32 ? f32_chain_tile(*d[0]) : -1;
  for (int i = 0; i < n; ++i) {
    const GemmDesc &g = *d[i];
    if (f32) {
      if (g.dtype != DT_F32 || !g.beta0 || f32_chain_tile(g) < 0 || f32_chain_tile(g) != f32_tile)
        NOCHAIN("an f32 call is not beta 0 / not planned on the K-split loader-wave tile of the first call");
      if (g.bias && ((uintptr_t)pd[i] & 15)) NOCHAIN("an f32 bias operand is not 16-byte aligned");
      if (g.ldc & 3) NOCHAIN("an f32 output's leading dimension is not a multiple of 4");
    } else
    // every layer the same kind of B operand (VNNI-2, flat or VNNI-4: the B image is a template parameter of the launch)
    if (g.dtype != DT_BF16 || g.vnni_c || !g.beta0 || bf16_lw_b_kind(g) < 0 || bf16_lw_b_kind(g) != bf16_lw_b_kind(*d[0]))
      NOCHAIN("a call is not bf16 / beta 0 / aligned for the LDS-DMA tiles, or the calls' B operands differ in kind (VNNI-2 / flat / VNNI-4)");
    if (g.m != m || g.n != nn || br[i] < 1) NOCHAIN("the calls differ in m or n, or a batch is empty");
    if (g.variant == GEMM_VARIANT_GENERIC) NOCHAIN("a call was dispatched to the generic kernel"); // (a forced generic kernel stays generic)
    if (((uintptr_t)pa[i] | (uintptr_t)pb[i] | (uintptr_t)pc[i]) & 15) NOCHAIN("an operand is not 16-byte aligned");
    if (g.bias && (!pd[i] || ((uintptr_t)pd[i] & 7))) NOCHAIN("a bias operand is not 8-byte aligned");
    if (i > 0 && (pa[i] != pc[i - 1] || g.lda != d[i - 1]->ldc)) NOCHAIN("not a chain: a call does not read its predecessor's output");
    // The kernel hands layer i-1's output over row block by row block (a consumer waits for the producers of ITS rows only): every
    // batch element of layer i must stay inside its own rows, i.e. the batch strides walk along k within one leading dimension.
    // (A row-striding stride_a would read rows that other workgroups may not have stored yet.)
    if (i > 0 && (br[i] - 1) * g.stride_a + g.k > g.lda) NOCHAIN("a later call's batch elements leave the rows of its predecessor's output");
    if (!devmem.is_device(pa[i], 0) || !devmem.is_device(pb[i], 1) || !devmem.is_device(pc[i], 2) || (g.bias && !devmem.is_device(pd[i], 3)))
      NOCHAIN("a host operand");
  }
  // The tile: all workgroups must be co-resident (one per CU by LDS), so the grid may not exceed the CUs. If every layer was planned
  // with the same loader-wave tile and that tile fits, use it - the launch is then bit-identical to the separate launches; else
  // the smallest tile that fits (most CUs busy).
  int tile = -1, bm = 0, bn = 0;
  const int64_t cus = stream_cus(s); // (the stream the launch goes to: ADVICE r4)
  auto fits = [&](int t) {
    if (f32) (void)f32_chain_tile_dims(t, &bm, &bn);
    else blw_tile_dims(t, &bm, &bn);
    return m % bm == 0 && nn % bn == 0 && (m / bm) * (nn / bn) <= cus;
  };
  if (f32 && !fits(f32_tile)) NOCHAIN("more tiles than compute units");
  const int b_kind = f32 ? 0 : bf16_lw_b_kind(*d[0]);
  // (variants 20 .. 23 VNNI-2, 24 .. 27 flat B, 28 .. 31 VNNI-4: the same four tiles)
  const int planned = d[0]->variant - (b_kind == 2 ? GEMM_VARIANT_BF16_LW0 + 4 : b_kind == 4 ? GEMM_VARIANT_BF16_LW4_0 : GEMM_VARIANT_BF16_LW0);
  bool same = !f32 && planned >= 0 && planned < 4;
  for (int i = 1; i < n && same; ++i) same = d[i]->variant == d[0]->variant;
  if (f32) tile = f32_tile;
  if (same && fits(planned)) tile = planned;
  if (tile < 0 && cfg().strict.load(std::memory_order_relaxed)) NOCHAIN("strict mode: one launch only on the tile the layers were planned on");
  for (int t = 0; t < 4 && tile < 0; ++t)
    if (fits(t)) tile = t;
  if (tile < 0) NOCHAIN("more tiles than compute units");
  (void)fits(tile); // bm, bn of the chosen tile
  // no operand of the launch may overlap an output (a layer's input rows are read by other workgroups while later layers store)
  Operand A, B, C, D;
  struct Span { const void *p; size_t n; };
  Span outs[CH_MAXL], ins[2 * CH_MAXL + 1], a_in[CH_MAXL];
  int n_ins = 0;
  for (int i = 0; i < n; ++i) {
    gemm_operands(d[i], pa[i], pb[i], pc[i], pd[i], br[i], A, B, C, D);
    outs[i] = Span{C.ptr, C.bytes};
    a_in[i] = Span{A.ptr, A.bytes};
    ins[n_ins++] = Span{B.ptr, B.bytes};
    if (d[i]->bias) ins[n_ins++] = Span{D.ptr, D.bytes};
    if (i == 0) ins[n_ins++] = Span{A.ptr, A.bytes};
  }
  for (int i = 0; i < n; ++i) {
    for (int j = i + 1; j < n; ++j)
      if (ranges_overlap(outs[i].p, outs[i].n, outs[j].p, outs[j].n)) NOCHAIN("two outputs overlap");
    for (int j = 0; j < n_ins; ++j)
      if (ranges_overlap(outs[i].p, outs[i].n, ins[j].p, ins[j].n)) NOCHAIN("an output overlaps an input");
    // the A operand of a later layer is its predecessor's output by construction; what it reads (k may be wider than the
    // predecessor's n: the gap columns of the rows) may overlap no OTHER output of the launch
    for (int j = 1; j < n; ++j)
      if (j != i + 1 && ranges_overlap(outs[i].p, outs[i].n, a_in[j].p, a_in[j].n)) NOCHAIN("an output overlaps a later call's input");
  }
#undef NOCHAIN
  ChainArgs c;
  memset(&c, 0, sizeof(c));
  c.A = pa[0];
  c.lda = d[0]->lda;
  c.m = (int)m;
  c.n = (int)nn;
  c.nlayers = n;
  c.dbg = chain_ablation_bits();
  for (int i = 0; i < n; ++i)
    c.L[i] = ChainLayer{pb[i], pd[i], pc[i], d[i]->ldb, d[i]->ldc, d[i]->stride_a, d[i]->stride_b, (int)d[i]->k, (int)br[i],
                        EP_BETA0 | (d[i]->bias ? EP_BIAS : 0) | (d[i]->relu ? EP_RELU : 0), 0};
  // the pool of error words is about to run dry (hundreds of launches without a synchronisation): synchronise and check here
  // instead of ever dropping a journal entry
  if (g_chain_journaled.load(std::memory_order_relaxed) >= CHAIN_ERR_POOL - 8) {
    std::vector<hipStream_t> streams;
    {
      std::lock_guard<std::mutex> lk0(g_chain_mu);
      for (const ChainCall &j : g_chain_journal)
        if (std::find(streams.begin(), streams.end(), j.stream) == streams.end()) streams.push_back(j.stream);
    }
    for (hipStream_t st : streams) {
      HIP_OK(hipStreamSynchronize(st));
      check_chain_errors(st);
    }
    if (g_chain_shared.load(std::memory_order_acquire)) return false; // (found a starved launch: call by call from here on)
  }
  std::lock_guard<std::mutex> lk(g_chain_mu);
  ChainBlock &blk = chain_block(s, (int)(m / bm), (int)(nn / bn), n);
  c.cnt = blk.cnt;
  c.err = blk.err; // probation launches: the block's word (checked right behind the launch)
  if (blk.verified >= 1) {
    if (g_chain_err_free.empty() && g_chain_journal.empty()) { // first use: the pool
      unsigned *pool = nullptr;
      HIP_OK(hipHostMalloc((void **)&pool, sizeof(unsigned) * CHAIN_ERR_POOL, hipHostMallocDefault));
      for (int i = 0; i < CHAIN_ERR_POOL; ++i) {
        pool[i] = 0;
        g_chain_err_free.push_back(pool + i);
      }
    }
    if (g_chain_err_free.empty()) return false; // (cannot happen: the check above keeps 8 words spare; call by call is always right)
    c.err = g_chain_err_free.back();
    g_chain_err_free.pop_back();
  }
  c.target = ++blk.epoch * (unsigned)blk.tiles_n;
  c.stamps = chain_stamps((size_t)blk.tiles_m * (size_t)blk.tiles_n);
  if (f32) HIP_OK(launch_f32_chain(tile, c, s));
  else HIP_OK(launch_bf16_chain(tile, b_kind, c, s));
  if (blk.verified < 1) {
    // probation (comment at ChainBlock): wait for this launch and look at its error word before anyone can consume its outputs
    HIP_OK(hipStreamSynchronize(s));
    const unsigned e = *(volatile unsigned *)blk.err;
    if (e) {
      *(volatile unsigned *)blk.err = 0;
      g_chain_shared.store(true, std::memory_order_release);
      fprintf(stderr, "[tpp-xsmm-hip] the first fused-brgemm chain launch on this stream was starved (a hand-off for layer %u's input timed "
                      "out: not every workgroup was resident - the device is shared): this call and every later chain invoke run call "
                      "by call.\n", e - 1);
      return false; // the caller runs the calls one by one (inputs intact: beta 0, outputs overlap no operand)
    }
    ++blk.verified;
    return true;
  }
  // journal: the calls of this launch with its own error word, for a re-run should the check at the next synchronisation of this
  // stream find it starved
  {
    ChainCall j;
    j.n = n;
    j.dtype = d[0]->dtype;
    j.stream = s;
    j.err = c.err;
    for (int i = 0; i < n; ++i) {
      j.handle[i] = reinterpret_cast<int64_t>(d[i]);
      j.a[i] = pa[i]; j.b[i] = pb[i]; j.c[i] = pc[i]; j.d[i] = pd[i]; j.br[i] = br[i];
    }
    g_chain_journal.push_back(j);
    g_chain_journaled.store((int)g_chain_journal.size(), std::memory_order_relaxed);
  }
  g_chain_launched.store(1, std::memory_order_release);
  return true;
}

// the calls of one journaled chain launch, one by one (operands are pointers with offsets applied: offsets 0)
void chain_rerun_call_by_call(const ChainCall &c) {
  // on the stream the launch went to - through this thread's override: the process-wide setting is not touched (another thread may
  // invoke, or call xsmm_hip_set_stream, meanwhile: ADVICE r5)
  tl_hot.stream_override = c.stream;
  tl_hot.has_stream_override = true;
  for (int i = 0; i < c.n; ++i)
    xsmm_fused_brgemm_invoke(c.dtype, c.handle[i], c.a[i], 0, c.b[i], 0, c.c[i], 0, c.d[i], 0, c.br[i]);
  flush_tile_queue();
  tl_hot.has_stream_override = false;
}
