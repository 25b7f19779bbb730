// rt_scheduler.h - several calling threads: one private single-producer ring per caller, merged by time stamp by ONE scheduler thread
// (parking / waking, idle retirement). A subsystem unit of runtime.cpp (see the note at the top of rt_core.h's siblings): included between
// the queue state (rt_tile_queue.h) and the enqueue path (rt_enqueue.h).

// The scheduler. The reference calls invoke from OpenMP workers (scf.parallel over the tile grid): with one
// lock around the dependence bookkeeping eight callers took 290 us for what one caller did in 45 (lock
// hand-offs, and interleaved callers defeat the interval merging). Callers therefore only HAND OVER their
// invokes; a single scheduler thread does the dependence bookkeeping without any lock and launches a group
// whenever the next invoke conflicts with it. Callers never touch HIP on this path; launches and slot waits
// happen on the scheduler thread, overlapped with the callers.
//
// Hand-over = one private single-producer ring per calling thread, merged by TIME STAMP. (Round 1 used one
// multi-producer ring with a ticket counter: on the 256-core host of the GPU box the counter's cache line
// hopping between the callers cost 130-230 ns per invoke - two callers took 180 us for what one did in 48.)
//   * An entry is ONE cache line: stamp, descriptor, the four operand pointers, batch count, stream; the
//     scheduler derives the footprints itself (queued_operands). A push writes that line and nothing shared.
//   * The stamp is the invariant TSC (`lfence; rdtsc`) when the kernel trusts it as its clock source, else a shared
//     counter. Either way  a happens-before b  =>  stamp(a) < stamp(b), and a is visible to whoever sees b.
//   * The scheduler keeps the non-empty rings in a min-heap on the stamp of their oldest entry and always takes
//     the smallest. A ring it finds empty stays WARM for a while: its next slot (a line in the scheduler's cache
//     until the producer writes it) is polled before every pop. A ring that stays empty for some thousand polls is
//     PARKED (flag in the ring, Dekker-style re-check); the producer's next push sees the flag and announces
//     the ring on a small wake list - the only shared write on the producer side, once per burst.
//   * Before every pop the warm rings and the wake list are polled until a whole pass finds nothing new. So when an
//     entry b is taken, every entry that happened before b is already consumed, or in the heap with a smaller stamp,
//     or behind such an entry in its own ring: the processing order respects every caller's program order and
//     every happens-before between callers (an OpenMP barrier, a join). Entries without such a relation are
//     concurrent invokes of the caller's program, and those do not conflict in a race-free program.
struct alignas(64) PSlot {
  std::atomic<uint32_t> seq; // (uint32_t)(index + 1) once the entry at `index` is complete
  int32_t br;
  uint64_t stamp;
  const void *desc; // nullptr: a fence, C = the std::atomic<int> to raise once everything before it is launched
  const void *A, *B;
  void *C;
  const void *D;
  hipStream_t stream;
};
static_assert(sizeof(PSlot) == 64, "one cache line per queued invoke");

struct PQueue {
  static constexpr uint64_t CAP = 2048, MASK = CAP - 1;
  PSlot *ring = nullptr;
  // producer side
  alignas(64) uint64_t tail = 0;
  uint64_t head_seen = 0;              // last value read from head_pub
  std::atomic<uint64_t> tail_pub{0};   // = tail, for drain()'s "anything pending?" test
  // consumer side
  alignas(64) uint64_t head = 0;       // next index to consume (owned by the live scheduler thread)
  std::atomic<uint64_t> head_pub{0};   // published every 16 entries and when the ring is parked: the producer reads it only when the ring looks full
  std::atomic<uint64_t> clean_head{0}; // every entry below this has been LAUNCHED
  // rarely written by either side
  alignas(64) std::atomic<int> parked{1}; // 1: the scheduler is not watching this ring - the next push must announce it
  std::atomic<int> owned{0};               // a caller thread holds this ring
};

struct Scheduler {
  static constexpr int MAXQ = 1024;
  std::atomic<PQueue *> queues[MAXQ];
  std::atomic<int> nq{0}; // high-water mark of allocated rings
  std::mutex alloc_mu;
  PQueue overflow; // more than MAXQ simultaneous caller threads: they share this ring under a mutex
  std::mutex overflow_mu;
  // wake list: ring indices + 1 (0 = empty cell); a ring is on it at most once, so MAXQ + 1 cells cannot overflow
  static constexpr uint32_t WCAP = 2048;
  alignas(64) std::atomic<uint32_t> wake_tail{0};
  alignas(64) std::atomic<uint32_t> wake_cell[WCAP];
  uint32_t wake_head = 0; // scheduler thread only

  const bool use_tsc;
  // Parking a ring is a Dekker pair (producer: publish entry, read `parked`; scheduler: set `parked`, re-read the slot). The
  // producer's side runs once per invoke and a full fence there stalls it on the slot line's ownership request (the line is in the
  // scheduler's cache from the previous lap: ~150 ns across cores, measured as 260 ns per invoke with two callers), so the
  // fence is moved to the side that runs once per burst: the scheduler issues membarrier(PRIVATE_EXPEDITED) - a full barrier on
  // every thread of the process - between its two steps, and the producers use plain release stores / loads. Without that
  // system call (old kernels, seccomp) the producers fall back to sequentially consistent stores.
  const bool asym_fence;
  alignas(64) std::atomic<uint64_t> stamp_ctr{1};

  std::atomic<bool> stop{false};
  std::thread worker;
  int device = 0;
  TileQueue q;
  DeviceRanges devmem; // the worker's allocation cache (per epoch, like the callers' own)
  std::vector<std::pair<uint64_t, int>> heap; // (stamp of the ring's oldest entry, ring index), min on top
  struct Warm {
    int qi;
    unsigned polls;
  };
  std::vector<Warm> warm; // rings found empty a moment ago
  static constexpr unsigned PARK_AFTER = 4096; // polls without an entry before a warm ring is parked

  // The worker exists only while there is traffic: after ~2 s without an entry it leaves (a library that was used
  // once must not keep a thread napping for the rest of the process), and the next push starts a new one. The
  // hand-over is a Dekker pair on (running, wake list): the worker clears `running` BEFORE it re-reads the wake list
  // (every ring is parked while the worker idles, so every push goes through that list), a producer announces its
  // ring BEFORE it reads `running` - at least one of them sees the other.
  alignas(64) std::atomic<bool> running{false}; // read by every producer on every push: its own cache line, written twice in a worker's life
  alignas(64) std::mutex life_mu;

  static bool kernel_trusts_tsc() {
    char buf[32] = {0};
    if (FILE *f = fopen("/sys/devices/system/clocksource/clocksource0/current_clocksource", "r")) {
      if (!fgets(buf, sizeof(buf), f)) buf[0] = 0;
      fclose(f);
    }
    return strncmp(buf, "tsc", 3) == 0;
  }
  static bool register_membarrier() {
    if (getenv("TPP_HIP_NO_MEMBARRIER")) return false;
    return syscall(__NR_membarrier, MEMBARRIER_CMD_REGISTER_PRIVATE_EXPEDITED, 0) == 0;
  }
  Scheduler() : use_tsc(kernel_trusts_tsc() && !getenv("TPP_HIP_NO_TSC")), asym_fence(register_membarrier()) {
    for (auto &c : queues) c.store(nullptr, std::memory_order_relaxed);
    for (auto &c : wake_cell) c.store(0, std::memory_order_relaxed);
    init_ring(overflow);
    if (hipGetDevice(&device) != hipSuccess) device = 0;
  }
  ~Scheduler() {
    stop.store(true);
    std::thread w;
    {
      std::lock_guard<std::mutex> lk(life_mu);
      w = std::move(worker);
    }
    if (!w.joinable()) return;
    // a fatal error on the scheduler thread itself exits the process from that thread: never join yourself
    if (w.get_id() == std::this_thread::get_id()) w.detach();
    else w.join(); // outside life_mu: a worker on its way out takes that lock
  }
  static void init_ring(PQueue &Q) {
    Q.ring = static_cast<PSlot *>(aligned_alloc(64, sizeof(PSlot) * PQueue::CAP));
    if (!Q.ring) die("tpp-xsmm-hip: out of memory for a caller's invoke ring");
    for (uint64_t i = 0; i < PQueue::CAP; ++i) new (&Q.ring[i].seq) std::atomic<uint32_t>(0);
  }
  uint64_t stamp() {
#if defined(__x86_64__)
    if (use_tsc) {
      unsigned lo, hi;
      asm volatile("lfence\n\trdtsc" : "=a"(lo), "=d"(hi)::"memory"); // after every earlier load (the caller's synchronisation) has completed
      return ((uint64_t)hi << 32) | lo;
    }
#endif
    return stamp_ctr.fetch_add(1, std::memory_order_seq_cst);
  }
  PQueue *ring_at(int i) { return i == MAXQ ? &overflow : queues[i].load(std::memory_order_acquire); }

  // ---- caller side -------------------------------------------------------------------------------------------
  // the calling thread's ring: claimed on first use, handed back when the thread ends (entries still in it stay
  // valid; the next owner continues at its tail)
  struct Lease {
    Scheduler *s = nullptr;
    int idx = -1;
    ~Lease() {
      if (s && idx >= 0 && idx < MAXQ) s->queues[idx].load(std::memory_order_relaxed)->owned.store(0, std::memory_order_release);
    }
  };
  int claim() {
    const int n = nq.load(std::memory_order_acquire);
    for (int i = 0; i < n; ++i) {
      PQueue *Q = queues[i].load(std::memory_order_acquire);
      int expect = 0;
      if (Q && Q->owned.load(std::memory_order_relaxed) == 0 && Q->owned.compare_exchange_strong(expect, 1, std::memory_order_acq_rel)) return i;
    }
    std::lock_guard<std::mutex> lk(alloc_mu);
    const int m = nq.load(std::memory_order_relaxed);
    if (m >= MAXQ) return MAXQ; // the shared overflow ring
    PQueue *Q = new PQueue;
    init_ring(*Q);
    Q->owned.store(1, std::memory_order_relaxed);
    queues[m].store(Q, std::memory_order_release);
    nq.store(m + 1, std::memory_order_release);
    return m;
  }
  int my_ring() {
    thread_local Lease lease;
    if (lease.s != this) {
      lease.s = this;
      lease.idx = claim();
    }
    return lease.idx;
  }
  void ensure_worker() {
    if (running.load(std::memory_order_seq_cst)) return;
    std::lock_guard<std::mutex> lk(life_mu);
    if (running.load(std::memory_order_relaxed) || stop.load()) return;
    if (worker.joinable()) worker.join(); // the previous worker has left (it cleared `running` on its way out)
    running.store(true, std::memory_order_seq_cst);
    worker = std::thread([this] { run(); });
  }
  void push_to(int qi, PQueue &Q, const QEntry &e) {
    const uint64_t h = Q.tail;
    if (h - Q.head_seen >= PQueue::CAP) {
      // Ring full: this caller outruns the scheduler. Wait until HALF of it is free again, not for one slot: the scheduler
      // then streams through a backlog of finished (prefetched) lines while the producer refills in a burst, instead of
      // the two moving in lockstep with every line crossing cores just in time.
      for (unsigned spins = 0; h - (Q.head_seen = Q.head_pub.load(std::memory_order_acquire)) > PQueue::CAP / 2; ++spins) {
        if (stop.load(std::memory_order_relaxed)) return; // the process is exiting (static destruction): nobody will consume the ring
        if (spins < 2000) cpu_relax();
        else {
          ensure_worker();
          sched_yield();
        }
      }
    }
    PSlot &s = Q.ring[h & PQueue::MASK];
    s.br = (int32_t)e.w.br;
    s.desc = e.desc;
    s.A = e.w.A;
    s.B = e.w.B;
    s.C = e.w.C;
    s.D = e.w.D;
    s.stream = e.stream;
    s.stamp = stamp();
    s.seq.store((uint32_t)(h + 1), asym_fence ? std::memory_order_release : std::memory_order_seq_cst); // Dekker with `parked`, see asym_fence
    Q.tail = h + 1;
    Q.tail_pub.store(h + 1, std::memory_order_relaxed);
    if (Q.parked.load(std::memory_order_seq_cst) && Q.parked.exchange(0, std::memory_order_seq_cst)) {
      const uint32_t pos = wake_tail.fetch_add(1, std::memory_order_seq_cst);
      std::atomic<uint32_t> &cell = wake_cell[pos % WCAP];
      while (cell.load(std::memory_order_acquire) != 0) cpu_relax(); // (a lap behind: cannot happen with <= MAXQ + 1 rings)
      cell.store((uint32_t)qi + 1, std::memory_order_seq_cst);
    }
    ensure_worker();
  }
  void push(const QEntry &e) {
    if (e.w.br > 0x7fffffff) die("tpp-xsmm-hip: batch count %ld is too large for the tile queue", (long)e.w.br);
    const int qi = my_ring();
    if (qi == MAXQ) {
      std::lock_guard<std::mutex> lk(overflow_mu);
      push_to(qi, overflow, e);
    } else {
      push_to(qi, *queues[qi].load(std::memory_order_relaxed), e);
    }
  }
  // everything pushed before this call (by this thread, or by another with a happens-before to this call) has been
  // launched on return
  void drain() {
    bool pending = false;
    const int n = nq.load(std::memory_order_acquire);
    for (int i = 0; i <= n && !pending; ++i) {
      PQueue *Q = i == n ? &overflow : queues[i].load(std::memory_order_acquire);
      pending = Q && Q->clean_head.load(std::memory_order_acquire) < Q->tail_pub.load(std::memory_order_acquire);
    }
    if (!pending || stop.load(std::memory_order_relaxed)) return;
    std::atomic<int> flag{0};
    QEntry f;
    f.w.C = &flag; // desc == nullptr: a fence
    push(f);
    for (unsigned spins = 0; !flag.load(std::memory_order_acquire); ++spins) {
      // the scheduler is being destroyed (exit() on another thread while this one flushes): its worker will not start again
      // (ensure_worker) and the fence would never be raised - give up instead of spinning through process teardown
      if (stop.load(std::memory_order_relaxed) && !running.load(std::memory_order_seq_cst)) return;
      if (spins < 4000) cpu_relax();
      else sched_yield();
    }
  }

  // ---- scheduler thread -----------------------------------------------------------------------------------------
  static bool later(const std::pair<uint64_t, int> &a, const std::pair<uint64_t, int> &b) { return a.first > b.first; }
  bool take_if_ready(int qi, PQueue &Q) { // the ring's next slot: into the heap with it if it is complete
    PSlot &s = Q.ring[Q.head & PQueue::MASK];
    if (s.seq.load(std::memory_order_acquire) != (uint32_t)(Q.head + 1)) return false;
    heap.emplace_back(s.stamp, qi);
    std::push_heap(heap.begin(), heap.end(), later);
    return true;
  }
  void examine(int qi, PQueue &Q) { // after a pop / a wake-up: heap or warm list
    if (!take_if_ready(qi, Q)) warm.push_back(Warm{qi, 0});
  }
  bool park(int qi, PQueue &Q) { // true: an entry slipped in and is in the heap now
    PSlot &s = Q.ring[Q.head & PQueue::MASK];
    Q.head_pub.store(Q.head, std::memory_order_release);
    Q.parked.store(1, std::memory_order_seq_cst);
    if (asym_fence && syscall(__NR_membarrier, MEMBARRIER_CMD_PRIVATE_EXPEDITED, 0) != 0) die("tpp-xsmm-hip: membarrier failed");
    if (s.seq.load(std::memory_order_seq_cst) == (uint32_t)(Q.head + 1) && Q.parked.exchange(0, std::memory_order_seq_cst)) {
      // an entry arrived while the ring was being parked and its producer has not taken the flag: it is ours again
      // (if the producer took the flag, the ring comes back through the wake list)
      heap.emplace_back(s.stamp, qi);
      std::push_heap(heap.begin(), heap.end(), later);
      return true;
    }
    return false;
  }
  bool drain_wake_list() {
    bool any = false;
    for (;;) {
      std::atomic<uint32_t> &cell = wake_cell[wake_head % WCAP];
      uint32_t v = cell.load(std::memory_order_seq_cst);
      if (!v) {
        // Producers RESERVE a cell (fetch_add on wake_tail) and fill it afterwards: an empty cell below the reserved tail is a
        // producer between its two steps. A later cell - or a warm ring - may already hold an entry that happened AFTER that
        // producer's push (it saw the push through a barrier), so stopping here would let that entry overtake it. Wait for the
        // laggard: the window is a few instructions unless the producer was preempted inside it.
        if (wake_tail.load(std::memory_order_seq_cst) == wake_head) return any;
        while (!(v = cell.load(std::memory_order_acquire))) cpu_relax();
      }
      cell.store(0, std::memory_order_release);
      ++wake_head;
      any = true;
      examine((int)v - 1, *ring_at((int)v - 1));
    }
  }
  // one pass over the warm rings; true if an entry turned up. count: this pass counts towards parking
  bool poll_warm(bool count) {
    bool any = false;
    for (size_t i = 0; i < warm.size();) {
      PQueue &Q = *ring_at(warm[i].qi);
      if (take_if_ready(warm[i].qi, Q)) {
        any = true;
      } else if (count && ++warm[i].polls > PARK_AFTER) {
        any = park(warm[i].qi, Q) || any;
      } else {
        ++i;
        continue;
      }
      warm[i] = warm.back();
      warm.pop_back();
    }
    return any;
  }
  // everything that happened before any entry now in the heap is consumed, in the heap, or behind a heap entry of its ring
  void collect() {
    bool any = drain_wake_list();
    any = poll_warm(true) || any;
    while (any) { // an entry turned up: whatever happened before IT was published earlier - look again
      any = drain_wake_list();
      any = poll_warm(false) || any;
    }
  }
  void mark_clean() { // everything consumed so far has been launched
    const int n = nq.load(std::memory_order_acquire);
    for (int i = 0; i <= n; ++i) {
      PQueue *Q = i == n ? &overflow : queues[i].load(std::memory_order_acquire);
      if (Q) Q->clean_head.store(Q->head, std::memory_order_release);
    }
  }
  void run() {
    // the creating thread may be pinned (OMP_PROC_BIND pins each worker to one core): inheriting that mask would
    // put the scheduler on the caller's own core. Use the mask the PROCESS had when the library was loaded
    // (taskset / numactl / cgroup limits are respected; only later per-thread pinning is undone).
    if (g_have_process_mask) (void)sched_setaffinity(0, sizeof(g_process_mask), &g_process_mask);
    (void)hipSetDevice(device);
    unsigned idle = 0;
    for (;;) {
      collect();
      if (!heap.empty()) {
        std::pop_heap(heap.begin(), heap.end(), later);
        const int qi = heap.back().second;
        heap.pop_back();
        PQueue &Q = *ring_at(qi);
        const PSlot &s = Q.ring[Q.head & PQueue::MASK];
        QEntry e;
        e.desc = s.desc;
        e.w = WorkItem{s.A, s.B, s.C, s.D, s.br};
        e.stream = s.stream;
        __builtin_prefetch(&Q.ring[(Q.head + 4) & PQueue::MASK]);
        __builtin_prefetch(&Q.ring[(Q.head + 8) & PQueue::MASK]);
        ++Q.head;
        if ((Q.head & 15) == 0) Q.head_pub.store(Q.head, std::memory_order_release);
        examine(qi, Q);
        idle = 0;
        if (!e.desc) {
          q.flush();
          mark_clean();
          ((std::atomic<int> *)e.w.C)->store(1, std::memory_order_release);
        } else {
          devmem.refresh();
          submit_item(q, devmem, e.desc, e.w, e.stream);
        }
        continue;
      }
      if (stop.load(std::memory_order_relaxed)) break;
      if (++idle < 4000) cpu_relax();
      else if (idle < 20000) sched_yield();
      else { // nothing for a long while: stop burning a core (a caller that arrives now waits one nap)
        timespec ts{0, idle < 40000 ? 50000 : 1000000}; // 50 us naps, then 1 ms naps
        nanosleep(&ts, nullptr);
        if (idle > 42000) { // ~2 s of 1 ms naps: leave, unless a producer has announced a ring meanwhile
          std::lock_guard<std::mutex> lk(life_mu); // ensure_worker() joins this thread under the same lock: decide inside it
          if (!warm.empty()) continue; // (every ring must be parked before the worker may leave)
          running.store(false, std::memory_order_seq_cst);
          if (!drain_wake_list()) return;
          running.store(true, std::memory_order_seq_cst);
          idle = 0;
        }
      }
    }
  }
};
std::atomic<Scheduler *> g_sched{nullptr};
std::mutex g_sched_mu;
Scheduler &sched() {
  Scheduler *p = g_sched.load(std::memory_order_acquire);
  if (!p) {
    std::lock_guard<std::mutex> lk(g_sched_mu);
    p = g_sched.load(std::memory_order_relaxed);
    if (!p) {
      static Scheduler the_scheduler; // destroyed (worker joined) at process exit
      p = &the_scheduler;
      g_sched.store(p, std::memory_order_release);
    }
  }
  return *p;
}
