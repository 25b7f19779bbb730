// rt_launcher.h - the LAUNCH THREAD: complete replayed groups are launched off the calling thread (round 6).
// A subsystem unit of runtime.cpp (one translation unit; include runtime.cpp's way only).
//
// In the steady state of compiled code - a recorded group replayed through the direct window - an invoke costs its caller 10-15 ns
// and the group's ONE launch costs 2.3-2.6 us of the same thread inside hipLaunchKernel: the reference's headline MLP as emitted
// (3 x 256 tile invokes per iteration) spends 8.8 us marking and 7.5 us launching, the pack / unpack scripts 4.8 + 2.5. The launch
// needs nothing of the caller: a complete replay is launched from the segment's device-resident work list with the recorded flags
// (TileQueue::issue_pending). So whoever closes such a group hands {descriptor, list, count, flags, stream} to this thread through a
// small ring and goes on with the next group's invokes; the launches leave in the order they were handed over.
//
// ORDER on a stream = order of the hand-overs + "every other launch waits for the ring to be empty first":
//   * anything else that launches, copies or synchronises - a partial group gathered at a flush, an invoke that does not go through
//     the queue, a synchronisation point, a work list being rebuilt - calls launcher_drain() first (TileQueue::flush's gathered branch,
//     Segment::ensure_list, flush_tile_queue(): every such path already started with one of the three);
//   * producers are serialised by what serialises the queue state (the inline queue's lock); the scheduler thread's own queue
//     (tile-queue mode 2) launches directly - it is off the callers' path already.
// The thread spins while launches keep coming (a hand-over is picked up in ~0.2 us), parks on a futex after ~0.1-0.2 ms without one and
// leaves after ~2 s (a library used once does not keep a thread); the next hand-over wakes / restarts it. A fatal HIP error on it
// ends the process like on any caller (die). TPP_HIP_LAUNCH_THREAD=0 / xsmm_hip_set_launch_thread(0): launches stay on the thread
// that closes the group (round 5's behaviour). Kernel choice, work lists and results are the same either way.
struct LaunchReq {
  int kind = 0;               // KIND_GEMM / KIND_UNARY / KIND_BINARY; -1: a merged tile grid (launch_gemm of `desc` on `w`); -2: 2 x 2 blocks of items (list = QuadItem[n])
  const void *desc = nullptr;
  const WorkItem *list = nullptr;
  int n = 0;
  bool vec_ok = true, out_ok = true, pair_ok = true;
  int64_t br = 0;
  WorkItem w{};
  hipStream_t stream = nullptr;
};
inline void issue_launch(const LaunchReq &r) {
  if (r.kind == -1) HIP_OK(launch_gemm(*(const GemmDesc *)r.desc, r.w.A, r.w.B, r.w.C, r.w.D, r.w.br, r.stream));
  else if (r.kind == -2) HIP_OK(launch_gemm_quads(*(const GemmDesc *)r.desc, (const QuadItem *)r.list, r.n, r.br, r.stream));
  else if (r.kind == KIND_GEMM) HIP_OK(launch_gemm_grouped(*(const GemmDesc *)r.desc, r.list, r.n, r.vec_ok, r.out_ok, r.pair_ok, r.br, r.stream));
  else if (r.kind == KIND_UNARY) HIP_OK(launch_unary_grouped(*(const UnaryDesc *)r.desc, r.list, r.n, r.stream));
  else HIP_OK(launch_binary_grouped(*(const BinaryDesc *)r.desc, r.list, r.n, r.stream));
}
struct Launcher {
  static constexpr uint64_t CAP = 64;
  LaunchReq ring[CAP];
  alignas(64) std::atomic<uint64_t> tail{0}; // hand-overs made (producers: under push_mu)
  alignas(64) std::atomic<uint64_t> head{0}; // hand-overs LAUNCHED (the worker)
  alignas(64) std::atomic<uint32_t> parked{0}; // futex word: 1 = the worker sleeps (or is about to)
  std::atomic<bool> running{false}, stop{false};
  std::atomic<int> on{1};
  std::atomic<int64_t> handed{0};
  std::mutex life_mu;
  std::thread worker;
  int device = 0;
  struct Spin {
    std::atomic<int> f{0};
    void lock() {
      while (f.exchange(1, std::memory_order_acquire)) cpu_relax();
    }
    void unlock() { f.store(0, std::memory_order_release); }
  } push_mu;
  Launcher() {
    if (const char *e = getenv("TPP_HIP_LAUNCH_THREAD")) on = atoi(e) != 0;
  }
  ~Launcher() {
    stop.store(true, std::memory_order_seq_cst);
    wake();
    std::thread w;
    {
      std::lock_guard<std::mutex> lk(life_mu);
      w = std::move(worker);
    }
    if (!w.joinable()) return;
    if (w.get_id() == std::this_thread::get_id()) w.detach(); // (a fatal error on the worker exits the process from there)
    else w.join();
  }
  void wake() {
    if (parked.load(std::memory_order_seq_cst)) {
      parked.store(0, std::memory_order_seq_cst);
      (void)syscall(SYS_futex, (uint32_t *)&parked, FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0);
    }
  }
  void ensure_worker() {
    if (running.load(std::memory_order_seq_cst)) return;
    std::lock_guard<std::mutex> lk(life_mu);
    if (running.load(std::memory_order_relaxed) || stop.load()) return;
    if (worker.joinable()) worker.join(); // the previous worker has left (it cleared `running` on its way out)
    if (hipGetDevice(&device) != hipSuccess) {
      (void)hipGetLastError();
      device = 0;
    }
    running.store(true, std::memory_order_seq_cst);
    worker = std::thread([this] { run(); });
  }
  // false: not handed over (switched off, or the process is exiting) - the caller launches itself
  bool push(const LaunchReq &r) {
    if (!on.load(std::memory_order_relaxed) || stop.load(std::memory_order_relaxed)) return false;
    std::lock_guard<Spin> lk(push_mu);
    const uint64_t t = tail.load(std::memory_order_relaxed);
    for (unsigned spins = 0; t - head.load(std::memory_order_acquire) >= CAP; ++spins) { // 64 launches behind: wait for one
      if (stop.load(std::memory_order_relaxed)) return false;
      if (spins < 4000) cpu_relax();
      else {
        ensure_worker();
        wake();
        sched_yield();
      }
    }
    ring[t % CAP] = r;
    tail.store(t + 1, std::memory_order_seq_cst); // Dekker with `parked` / `running`: the worker re-reads tail after setting either
    handed.store(handed.load(std::memory_order_relaxed) + 1, std::memory_order_relaxed);
    wake();
    ensure_worker();
    return true;
  }
  // every hand-over made before this call (by this thread, or with a happens-before to it) has been LAUNCHED on return
  void drain() {
    const uint64_t t = tail.load(std::memory_order_seq_cst);
    if (head.load(std::memory_order_acquire) >= t) return;
    for (unsigned spins = 0; head.load(std::memory_order_acquire) < t; ++spins) {
      if (stop.load(std::memory_order_relaxed) && !running.load(std::memory_order_seq_cst)) return; // process teardown: nobody will launch them
      if (spins < 20000) cpu_relax();
      else {
        ensure_worker();
        wake();
        sched_yield();
      }
    }
  }
  void run() {
    // (like the scheduler thread: the mask the PROCESS had at load time, not the one CPU of a pinned OpenMP caller that happened to
    // start this thread - rt_operands.h g_process_mask)
    if (g_have_process_mask) (void)sched_setaffinity(0, sizeof(g_process_mask), &g_process_mask);
    (void)hipSetDevice(device);
    uint64_t h = head.load(std::memory_order_relaxed);
    unsigned idle = 0, naps = 0;
    for (;;) {
      if (tail.load(std::memory_order_acquire) > h) {
        const LaunchReq r = ring[h % CAP];
        issue_launch(r);
        head.store(++h, std::memory_order_release);
        idle = 0;
        naps = 0;
        continue;
      }
      if (stop.load(std::memory_order_relaxed)) break;
      if (++idle < (1u << 12)) { // ~0.1-0.2 ms of polling behind the last launch (a timing loop hands over every 4-8 us), then the futex
        cpu_relax();
        continue;
      }
      idle = 0;
      parked.store(1, std::memory_order_seq_cst);
      if (tail.load(std::memory_order_seq_cst) > h || stop.load(std::memory_order_seq_cst)) {
        parked.store(0, std::memory_order_seq_cst);
        continue;
      }
      struct timespec ts = {0, 100 * 1000 * 1000};
      (void)syscall(SYS_futex, (uint32_t *)&parked, FUTEX_WAIT_PRIVATE, 1, &ts, nullptr, 0);
      parked.store(0, std::memory_order_seq_cst);
      if (tail.load(std::memory_order_seq_cst) > h) continue;
      if (++naps < 20) continue;
      // ~2 s without a launch: leave. Under life_mu (whoever starts a successor holds it and joins this thread first), and a Dekker
      // pair with push(): running = false, THEN tail is read again; push() publishes tail, THEN reads running.
      {
        std::lock_guard<std::mutex> lk(life_mu);
        running.store(false, std::memory_order_seq_cst);
        if (tail.load(std::memory_order_seq_cst) > h && !stop.load(std::memory_order_relaxed)) {
          running.store(true, std::memory_order_seq_cst); // a hand-over slipped in: still ours
          naps = 0;
          continue;
        }
      }
      return;
    }
    running.store(false, std::memory_order_seq_cst);
  }
};
Launcher &launcher() {
  static Launcher l;
  return l;
}
inline void launcher_drain() { launcher().drain(); }
