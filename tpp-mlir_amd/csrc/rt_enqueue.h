// rt_enqueue.h - the three ways into the tile queue (direct window / inline under the lock / handed to the scheduler), the calling
// thread's state behind one TLS pointer, enqueue_item: the per-invoke host path. A subsystem unit of runtime.cpp (one translation unit).

// Three ways into the queue state. DIRECT: a member of the recorded group that is being replayed is marked by its caller without
// any lock (DirectWindow) - the steady state of compiled code that repeats itself, from one thread or from the reference's OpenMP
// team alike. INLINE: everything else takes a spin lock and does the bookkeeping itself (45 ns per invoke for one caller).
// SCHEDULED: if several threads keep arriving on the locked path - a program the trace cache does not help, where one lock
// around the bookkeeping serialises the callers - the process switches, once and for good, to the rings + scheduler thread above
// (xsmm_hip_set_tile_queue(2) / TPP_HIP_TILE_QUEUE=2: as soon as a second thread shows up, the round-2 behaviour).
struct SpinLock {
  std::atomic<int> f{0};
  void lock() {
    for (unsigned spins = 0;; ++spins) {
      if (f.load(std::memory_order_relaxed) == 0 && f.exchange(1, std::memory_order_acquire) == 0) return; // (waiters spin on a shared line)
      if (spins < 4000) cpu_relax();
      else sched_yield();
    }
  }
  void unlock() { f.store(0, std::memory_order_release); }
};
struct InlineQueue {
  SpinLock mu;
  TileQueue q;
  DirectWindow dw;
  std::atomic<bool> scheduled{false}; // one-way switch, flipped under mu after q has been flushed
  uint64_t owner = 0;                 // thread that queued last (under mu)
  int foreign = 0;                    // arrivals of other threads since the last flush point (tile-queue mode 2)
  bool multi = false;                 // more than one thread has queued
  int64_t slow = 0, groups_at = 0;    // locked arrivals since a group was last replayed through the window / q.direct_groups then
  InlineQueue() { q.dw = &dw; }
};
InlineQueue &inl() {
  static InlineQueue i;
  return i;
}
std::atomic<int> g_dt_pending{0}; // number of remembered transposes (see "deferred transposes" below)
void dt_materialize();
void flush_tile_queue() {
  if (g_dt_pending.load(std::memory_order_acquire)) dt_materialize();
  if (!cfg().tile_queue.load(std::memory_order_relaxed)) {
    launcher_drain(); // (the queue was switched off behind a flush: nothing queued, but a handed-over launch may not have left yet)
    return;
  }
  InlineQueue &iq = inl();
  if (!iq.scheduled.load(std::memory_order_acquire)) {
    iq.dw.touch(thread_token());
    std::lock_guard<SpinLock> lk(iq.mu);
    if (!iq.scheduled.load(std::memory_order_relaxed)) {
      iq.q.flush();
      iq.foreign = 0;
      launcher_drain(); // whatever the caller does next on the stream (launch, copy, synchronise) is behind every queued launch
      return;
    }
  }
  launcher_drain();
  if (Scheduler *p = g_sched.load(std::memory_order_acquire)) p->drain();
}

// Queues one invoke of `desc`; true if queued (nothing launched yet), false if an operand is host memory (the
// caller flushes and takes the mirrored path). `ptrs` are the item's non-null operand pointers.
// The tile queue serves ONE device per process: the scheduler thread binds to the device of the first caller, work lists are
// plain pinned allocations and the tile heuristics cache that device's CU count. A caller on another device would get its
// grouped launches issued on the wrong GPU - refuse loudly instead (checked once per thread and synchronisation epoch, not per
// invoke). Non-queued invokes launch from the calling thread and follow its current device as usual.
std::atomic<int> g_queue_device{-1};
void check_queue_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return; // no device: the launch itself will fail loudly
  }
  int expect = -1;
  if (!g_queue_device.compare_exchange_strong(expect, dev) && expect != dev)
    die("tpp-xsmm-hip: the tile queue serves one device per process (first used on device %d, this thread's current device is %d); "
        "turn the queue off (xsmm_hip_set_tile_queue(0)) for multi-device processes", expect, dev);
}

// everything a calling thread keeps for the enqueue path, behind ONE thread-local lookup per invoke (in a shared library every
// thread_local access is a call into the dynamic TLS resolver)
struct CallerState;
// One pointer in the static TLS block (initial-exec: a %fs-relative load; the general-dynamic model of a shared library calls
// __tls_get_addr on every access - 10-15 cycles of an invoke), the state itself behind the usual thread_local so that it is
// destroyed with its thread. 8 bytes of the loader's static-TLS reserve: dlopen-safe.
// -DTPP_TLS_DEFAULT_MODEL: the compiler's default model for a shared object instead - for a process whose static-TLS
// surplus is already spent by other initial-exec libraries when this one is dlopen'ed ("cannot allocate memory in static TLS block").
#ifdef TPP_TLS_DEFAULT_MODEL
static __thread CallerState *tl_fast = nullptr;
#else
static __thread CallerState *tl_fast __attribute__((tls_model("initial-exec"))) = nullptr;
#endif
void dt_release_slot(int slot);
struct CallerState {
  DeviceRanges devmem; // per caller: no sharing, no lock
  DirectWindow::Caller *me = nullptr;
  bool claimed = false;
  int dt_slot = -1; // this thread's slot of remembered transposes ("deferred transposes" below)
  ~CallerState() {
    tl_fast = nullptr;
    if (me) me->owned.store(0, std::memory_order_release);
    if (dt_slot >= 0) dt_release_slot(dt_slot);
  }
};

InlineQueue *g_iq_fast = nullptr; // = &inl(), set before any thread's tl_fast (enqueue_item's fast path reads it without the static's guard)
static __attribute__((noinline)) CallerState &caller_state_slow() {
  thread_local CallerState tl;
  if (!__atomic_load_n(&g_iq_fast, __ATOMIC_ACQUIRE)) __atomic_store_n(&g_iq_fast, &inl(), __ATOMIC_RELEASE);
  tl_fast = &tl;
  return tl;
}
static inline CallerState &caller_state() {
  CallerState *p = tl_fast;
  return p ? *p : caller_state_slow();
}

bool enqueue_item_slow(const void *desc, const WorkItem &item, const void *const *ptrs, int n_ptrs, hipStream_t s);
// THE per-invoke path of compiled code that repeats itself from one thread (round 6): a SOLO caller whose invoke is the member the
// recorded group expects next, in a group whose pointers are proven for this epoch. Inlined into the entry points - no call, no
// callee-saved registers to spill, no hash: bracket (seq), window check, seven compares against items[hint], mark, count. Anything else
// - several callers, no window, another member, an unproven group, the first invoke of a thread - is enqueue_item_slow (the
// complete protocol, which starts with the same attempt; a failed attempt here leaves nothing behind but a tag / count reset that the
// slow path would have done too).
__attribute__((always_inline)) inline bool enqueue_item(const void *desc, const WorkItem &item, const void *const *ptrs, int n_ptrs, hipStream_t s) {
  CallerState *tl = tl_fast;
  if (__builtin_expect(tl != nullptr, 1)) {
    InlineQueue &iq = *g_iq_fast;
    DirectWindow::Caller *me = tl->me;
    const uint64_t c = iq.dw.cur.load(std::memory_order_acquire);
    const uint64_t epoch = g_devmem_epoch.load(std::memory_order_relaxed);
    if (c && me && tl->devmem.epoch == epoch) {
      const uint64_t seq0 = me->seq.load(std::memory_order_relaxed);
      me->seq.store(seq0 + 1, std::memory_order_relaxed); // BRACKET FIRST, then `multi`: see enqueue_item_slow
      std::atomic_signal_fence(std::memory_order_seq_cst);
      bool joined = false;
      if (!iq.dw.multi.load(std::memory_order_relaxed)) {
        me->busy.store(c, std::memory_order_relaxed);
        std::atomic_signal_fence(std::memory_order_seq_cst);
        if (iq.dw.cur.load(std::memory_order_relaxed) == c) {
          Segment &S = iq.q.segs[(c & 127) - 1];
          if (S.dev_epoch == epoch) {
            if (me->tag != c) {
              me->tag = c;
              me->count = 0;
            }
            const uint32_t hint = me->hint;
            if (hint < S.n_items && S.items[hint].same(desc, item, s) && S.mark_solo((int)hint)) {
              ++me->count;
              me->hint = hint + 1;
              joined = true;
            }
          }
        }
        me->busy.store(0, std::memory_order_release);
        std::atomic_signal_fence(std::memory_order_seq_cst);
      }
      me->seq.store(seq0 + 2, std::memory_order_release);
      if (joined) return true;
    }
  }
  return enqueue_item_slow(desc, item, ptrs, n_ptrs, s);
}
__attribute__((noinline)) bool enqueue_item_slow(const void *desc, const WorkItem &item, const void *const *ptrs, int n_ptrs, hipStream_t s) {
  CallerState &tl = caller_state();
  DeviceRanges &devmem = tl.devmem;
  static InlineQueue &iq = inl();
  // DIRECT: the invoke is a member of the recorded group being replayed. proven: only if the group's pointers have been proven
  // device memory in this epoch (Segment::prove) - the caller has not looked at its operands yet
  auto join_window = [&](bool proven, uint64_t epoch) __attribute__((always_inline)) -> bool {
    const uint64_t c = iq.dw.cur.load(std::memory_order_acquire);
    if (!c) return false;
    if (!tl.claimed) {
      tl.claimed = true;
      tl.me = iq.dw.claim();
    }
    DirectWindow::Caller *me = tl.me;
    if (!me) return false;
    // BRACKET FIRST: seq goes odd BEFORE `multi` is read, with a compiler barrier in between. The switching thread
    // sets multi, issues membarrier (an IPI = a full barrier at a precise point of this thread's instruction stream) and then waits
    // for an even seq. Interrupts are precise: either the seq store had retired when the IPI landed - then it is visible behind the
    // barrier and the switcher waits for this section to end -, or it had not - then the load of `multi` below had not retired
    // either, is re-executed behind the barrier and sees multi == true. (Round 4 read `multi` first: an IPI between the two
    // instructions let the switcher see an even seq while this thread went on into a solo section.)
    const uint64_t seq0 = me->seq.load(std::memory_order_relaxed);
    me->seq.store(seq0 + 1, std::memory_order_relaxed);
    std::atomic_signal_fence(std::memory_order_seq_cst);
    const bool solo = !iq.dw.multi.load(std::memory_order_relaxed); // (a thread that holds a slot and sees solo IS the one thread)
    if (solo) {
      me->busy.store(c, std::memory_order_relaxed);
      std::atomic_signal_fence(std::memory_order_seq_cst); // the compiler keeps busy-store, cur-load in this order (the hardware needs no fence: one thread)
    } else {
      me->seq.store(seq0 + 2, std::memory_order_release); // not solo after all: the bracket closes, the two-sided protocol from here
      me->busy.store(c, std::memory_order_seq_cst);
    }
    bool joined = false;
    if (iq.dw.cur.load(solo ? std::memory_order_relaxed : std::memory_order_seq_cst) == c) {
      Segment &S = iq.q.segs[(c & 127) - 1];
      if (!proven || S.dev_epoch == epoch) {
        if (me->tag != c) {
          me->tag = c;
          me->count = 0;
        }
        int idx = -1;
        if (me->hint < S.items.size() && S.items[me->hint].same(desc, item, s)) idx = (int)me->hint;
        else idx = S.index_of(desc, item, s);
        if (idx >= 0 && (solo ? S.mark_solo(idx) : S.mark(idx))) {
          ++me->count;
          me->hint = (uint32_t)idx + 1;
          joined = true;
        }
      }
    }
    me->busy.store(0, std::memory_order_release);
    if (solo) {
      std::atomic_signal_fence(std::memory_order_seq_cst);
      me->seq.store(seq0 + 2, std::memory_order_release); // even again: the solo section is over
    }
    return joined;
  };
  const uint64_t epoch = g_devmem_epoch.load(std::memory_order_relaxed);
  if (devmem.epoch == epoch && join_window(true, epoch)) return true; // (this thread has been through the checks below in this epoch)
  if (devmem.refresh()) check_queue_device();
  for (int i = 0; i < n_ptrs; ++i)
    if (!devmem.is_device(ptrs[i], i)) return false;
  if (join_window(false, 0)) return true;
  if (!iq.scheduled.load(std::memory_order_acquire)) {
    iq.dw.touch(thread_token());
    std::lock_guard<SpinLock> lk(iq.mu);
    if (!iq.scheduled.load(std::memory_order_relaxed)) {
      const uint64_t me = (uint64_t)(uintptr_t)&devmem; // the address of this thread's cache identifies the thread (one TLS lookup per invoke, not two)
      if (iq.owner != me) {
        if (iq.owner != 0) iq.multi = true;
        if (iq.owner != 0 && cfg().tile_queue.load(std::memory_order_relaxed) == 2 && ++iq.foreign > 4) iq.slow = 1 << 30;
        iq.owner = me;
      }
      if (iq.q.direct_groups != iq.groups_at) { // a group went through the window since the last look: the cache is working
        iq.groups_at = iq.q.direct_groups;
        iq.slow = 0;
      }
      if (iq.multi && ++iq.slow > 8192) { // several threads, and the locked path is where they meet: hand over to the scheduler
        iq.q.flush();
        (void)sched(); // create it (its worker thread starts with the first entry)
        iq.scheduled.store(true, std::memory_order_release);
      } else {
        submit_item(iq.q, devmem, desc, item, s);
        return true;
      }
    }
  }
  QEntry e;
  e.desc = desc;
  e.w = item;
  e.stream = s;
  sched().push(e);
  return true;
}

bool queue_active() {
  return cfg().tile_queue.load(std::memory_order_relaxed) && cfg().async.load(std::memory_order_relaxed);
}

bool try_enqueue(const GemmDesc *d, void *a, void *b, void *c, void *dp, int64_t br, hipStream_t s) {
  if (d->m > 64 || d->n > 64) return false; // big descriptors fill the chip on their own
  const void *ptrs[4] = {a, b, c, dp};
  return enqueue_item(d, WorkItem{a, b, c, dp, br}, ptrs, 4, s);
}
