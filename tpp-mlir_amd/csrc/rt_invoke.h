// rt_invoke.h - the invoke side of the GEMM family: strict-mode single items, the host-cache scope, gemm_invoke_common
// One of the subsystem units of runtime.cpp (round 6, VERDICT r5 next 7: the 3 000-line file split by subsystem, no behaviour
// change). The units are INCLUDED into the one translation unit runtime.cpp, in dependence order, inside its anonymous namespace:
// the per-invoke host path (14-18 ns: enqueue_item -> join_window -> Segment::mark) crosses four of them and is inlined across
// their borders - as separate objects without LTO it would pay a call per border. Not a stand-alone header: include runtime.cpp's way only.

// ---- strict mode: single invokes of queue-sized tiles run on the grouped launcher with a work list of ONE item. The kernels read
// their item from device-visible memory: a per-thread ring of pinned (device-mapped) items, like the tile queue's lists; the stream
// is drained once per lap of the ring, so a slot is never rewritten while a launch may still read it.
struct StrictRing {
  static constexpr int N = 1024;
  WorkItem *items = nullptr;
  int next = 0;
  ~StrictRing() {
    if (items) (void)hipHostFree(items);
  }
};
WorkItem *strict_item_slot(hipStream_t s) {
  thread_local StrictRing r;
  if (!r.items) HIP_OK(hipHostMalloc((void **)&r.items, sizeof(WorkItem) * StrictRing::N, hipHostMallocDefault));
  if (r.next == StrictRing::N) {
    HIP_OK(hipStreamSynchronize(s));
    r.next = 0;
  }
  return &r.items[r.next++];
}
void strict_item_done(hipStream_t) {}

// ---- host cache (round 6, host_cache.h): host operands translated to device mirrors that outlive the invoke ---------------------
// One scope per ABI invoke: the constructor translates the host operands (their pointers are REPLACED by mirror addresses, so the
// tile queue, the deferred transposes and the launch paths below see device memory), the destructor - behind the launch and, in
// synchronous mode, behind finish()'s stream synchronisation - copies what was written back (synchronous mode) or remembers it for
// the next synchronisation point (asynchronous mode) and ends the reader section.
void hc_flush_hook() { flush_tile_queue(); }
bool hc_is_device_hook(const void *p, int pos) {
  DeviceRanges &dm = caller_state().devmem;
  if (cfg().async.load(std::memory_order_relaxed)) dm.refresh();
  else dm.known.clear(); // synchronous mode: every invoke is a point after which the caller may free buffers (see stage_in)
  return dm.is_device(p, pos);
}
bool hc_setup() {
  hc::set_hooks(hc::Hooks{&hc_flush_hook, &hc_is_device_hook});
  if (const char *e = getenv("TPP_HIP_HOST_CACHE"))
    if (atoi(e) != 0) (void)hc::set_enabled(1);
  return true;
}
inline bool hc_on() {
  static const bool once = hc_setup();
  (void)once;
  return hc::enabled();
}
struct HcScope {
  hc::OpRef ops[4];
  int n = 0, hits = 0;
  bool async = false;
  hipStream_t s = nullptr;
  void *memo = nullptr; // asynchronous mode: the invoke was answered from the thread's whole-invoke memo (hc::memo_hit)
  void add(void **pp, const Operand &o, bool read, bool written) {
    ops[n++] = hc::OpRef{pp, o.bytes, o.rows, o.row_bytes, o.pitch, read, written, nullptr, 0};
  }
  uint64_t epoch = 0;
  void go(hipStream_t stream) {
    s = stream;
    async = cfg().async.load(std::memory_order_relaxed) != 0;
    epoch = g_devmem_epoch.load(std::memory_order_relaxed);
    hits = hc::translate(ops, n, async, epoch, s);
  }
  ~HcScope() {
    if (memo) {
      hc::memo_done(memo, epoch);
      return;
    }
    if (!hits) return;
    hc::complete(ops, n, async, epoch, s);
    hc::leave();
  }
};

// the invoke outside the tile queue (host operands, big descriptors, synchronous mode, strict-mode single items): out of line - the
// entry points inline gemm_invoke_common's front (checks, host-cache memo, folded transposes, the queue's fast path) and call this
__attribute__((noinline)) void gemm_invoke_unqueued(const GemmDesc *d, void *pa, void *pb, void *pc, void *pd, int64_t br, hipStream_t s) {
  Operand A, B, C, D;
  gemm_operands(d, pa, pb, pc, pd, br, A, B, C, D);
  C.read = !d->beta0; // pure output under BETA_0: never uploaded
  std::vector<Operand *> ops = {&A, &B, &C, &D};
  stage_in(ops, s);
  if (cfg().strict.load(std::memory_order_relaxed) && d->m <= 64 && d->n <= 64) {
    // strict mode: a tile the queue would take runs on the kernel its group runs on - the grouped launcher with a work list of one
    // (launch_gemm_grouped decides as if every list held one item: xsmm_desc.h strict_kernels)
    const WorkItem one{A.dev, B.dev, C.dev, D.dev, br};
    WorkItem *slot = strict_item_slot(s);
    *slot = one;
    HIP_OK(launch_gemm_grouped(*d, slot, 1, ((((uintptr_t)A.dev) | ((uintptr_t)B.dev)) & 15) == 0,
                               (((uintptr_t)C.dev) & 15) == 0 && (((uintptr_t)D.dev) & 7) == 0 && br >= 1, !(br & 1), br, s));
    strict_item_done(s);
  } else {
    HIP_OK(launch_gemm(*d, A.dev, B.dev, C.dev, D.dev, br, s));
  }
  finish(ops, s);
}
__attribute__((always_inline)) inline void gemm_invoke_common(const char *who, bool want_fused, int64_t dtype, int64_t handle, void *a, int64_t off_a,
                        void *b, int64_t off_b, void *c, int64_t off_c, void *dptr, int64_t off_d, int64_t br) {
  const GemmDesc *d = as_desc<GemmDesc>(handle, KIND_GEMM, who);
  if (d->dtype != dtype) die("%s: invoke dtype %ld != dispatch dtype %ld", who, (long)dtype, (long)d->dtype);
  if (want_fused != (d->fused != 0)) die("%s: handle dispatched for a different gemm flavour", who);
  if (br < 0) die("%s: negative batch count %ld", who, (long)br);
  if (d->m == 0 || d->n == 0) return;
  TraceRange trace_range(who, d->trace);
  const size_t es = esize(dtype);
  void *pa = (char *)a + off_a * es, *pb = (char *)b + off_b * es, *pc = (char *)c + off_c * es;
  void *pd = dptr ? (char *)dptr + off_d * es : nullptr;
  if (d->bias && !dptr) die("%s: fused bias operand is null", who);
  hipStream_t s = invoke_stream();
  HcScope hcs;
  if (hc_on()) {
    const bool async = cfg().async.load(std::memory_order_relaxed) != 0;
    hcs.epoch = g_devmem_epoch.load(std::memory_order_relaxed);
    if (async) hcs.memo = hc::memo_hit(d, &pa, &pb, &pc, &pd, br, hcs.epoch, s);
    if (!hcs.memo) {
      Operand A, B, C, D;
      gemm_operands(d, pa, pb, pc, pd, br, A, B, C, D);
      hcs.add(&pa, A, true, false);
      hcs.add(&pb, B, true, false);
      hcs.add(&pc, C, !d->beta0, true);
      hcs.add(&pd, D, true, false);
      hcs.go(s);
      if (async && hcs.hits) hc::memo_store(d, br, hcs.ops, 4, hcs.epoch, s);
    }
  }
  if (g_dt_pending.load(std::memory_order_acquire)) { // a remembered transpose: this gemm reads its source instead, or it is launched now
    void *src = nullptr;
    const GemmDesc *sib = dt_gemm_fast(d, pa, pb, pc, pd, br, s, &src);
    if (!sib) sib = dt_gemm(d, pa, pb, pc, pd, br, s, &src);
    if (sib) {
      d = sib;
      pb = src;
    }
  }
  if (cfg().tile_queue.load(std::memory_order_relaxed)) {
    if (cfg().async.load(std::memory_order_relaxed) && try_enqueue(d, pa, pb, pc, pd, br, s)) return;
    flush_tile_queue();
  }
  gemm_invoke_unqueued(d, pa, pb, pc, pd, br, s);
}
