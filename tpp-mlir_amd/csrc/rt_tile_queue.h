// rt_tile_queue.h - the tile queue's state: footprints, trace cache (segments), direct window, the group being collected and its bookkeeping
// One of the subsystem units of runtime.cpp (round 6, VERDICT r5 next 7: the 3 000-line file split by subsystem, no behaviour
// change). The units are INCLUDED into the one translation unit runtime.cpp, in dependence order, inside its anonymous namespace:
// the per-invoke host path (14-18 ns: enqueue_item -> join_window -> Segment::mark) crosses four of them and is inlined across
// their borders - as separate objects without LTO it would pay a call per border. Not a stand-alone header: include runtime.cpp's way only.

// ---- tile queue -------------------------------------------------------------------
// The compiler's native granularity is hundreds of invokes per layer on 32x32 tiles from
// OpenMP workers; one launch per invoke would be pure launch latency on a GPU. In async
// mode with the tile queue on, invokes of ONE small-tile GEMM handle on device pointers are
// appended to a work list and run as ONE grouped launch (brgemm_grouped) when something
// forces a flush: another handle or op, a data dependence on a queued output, capacity, a
// synchronize / perf_stop_timer, or leaving async mode. Program order is preserved: a new
// invoke that reads or overwrites anything a queued invoke writes (or overwrites anything a
// queued invoke reads) flushes first, so queued invokes are always mutually independent.
// union of half-open intervals: a sorted vector of disjoint ranges (a handful in practice - queued
// operands of one layer merge into a few runs - so a contiguous array beats a node-based map; the
// enqueue path runs 9 of these operations per invoke and is the throughput limit of the tile queue)
struct IntervalSet {
  std::vector<Range> iv; // sorted by begin, disjoint and non-touching
  void clear() { iv.clear(); }
  // index of the first interval whose begin is > x
  size_t upper(uintptr_t x) const {
    size_t lo = 0, hi = iv.size();
    if (hi <= 8) { // linear scan from the back: new operands are usually at or near the last run
      while (hi > 0 && iv[hi - 1].b > x) --hi;
      return hi;
    }
    while (lo < hi) {
      const size_t mid = (lo + hi) / 2;
      if (iv[mid].b > x) hi = mid;
      else lo = mid + 1;
    }
    return lo;
  }
  bool overlaps(const Range &r) const {
    if (r.b >= r.e || iv.empty()) return false;
    const size_t i = upper(r.e - 1); // intervals [0, i) begin before r.e
    return i > 0 && iv[i - 1].e > r.b;
  }
  void insert(Range r) {
    if (r.b >= r.e) return;
    size_t i = upper(r.b); // iv[i-1].b <= r.b < iv[i].b
    if (i > 0 && iv[i - 1].e >= r.b) { // r starts inside (or right at the end of) its predecessor
      if (iv[i - 1].e >= r.e) return;  // already covered: the common case for re-read operands
      --i;
      r.b = iv[i].b;
    }
    size_t j = i; // [i, j) are swallowed by r
    while (j < iv.size() && iv[j].b <= r.e) {
      r.e = std::max(r.e, iv[j].e);
      ++j;
    }
    if (j == i) iv.insert(iv.begin() + i, r);
    else {
      iv[i] = r;
      if (j > i + 1) iv.erase(iv.begin() + i + 1, iv.begin() + j);
    }
  }
};

// Footprint of the queued invokes' reads (or writes). Flat ranges are kept exactly in an interval
// set. A 2-D tile (rows x row_bytes, pitch) is kept as a rectangle in the "plane" (allocation base,
// pitch) it lives in - exact overlap tests between tiles of one row-major buffer, which is what pack /
// unpack tiles and the C tiles of a flat layer are - with a 64-row x 256-byte cell hash so a test
// touches a handful of rectangles. Anything that does not fit a plane falls back to its bounding
// range, and tests across different planes / against flat ranges use bounding ranges: conservative
// (may flush early), never unsafe.
struct Footprint {
  struct Rect { uint32_t r0, r1, c0, c1; };
  struct Plane {
    uintptr_t anchor;
    size_t pitch;
    IntervalSet bound;
    std::unordered_multimap<uint64_t, Rect> cells;
  };
  IntervalSet flat;
  std::vector<Plane> planes;
  void clear() { flat.clear(); planes.clear(); }
  static Range bounding(const Operand &o) { return Range{(uintptr_t)o.ptr, (uintptr_t)o.ptr + o.bytes}; }
  // rectangle of o in the plane (anchor, o.pitch); false if o is flat / wraps / is too wide for the hash
  static bool to_rect(const Operand &o, uintptr_t anchor, Rect &r) {
    if (!o.rows || !anchor) return false;
    const uintptr_t off = (uintptr_t)o.ptr - anchor;
    const uintptr_t r0 = off / o.pitch, c0 = off % o.pitch;
    if (c0 + o.row_bytes > o.pitch || o.row_bytes > 2048 || o.rows > 512 || r0 + o.rows > 0xffffffffu) return false;
    r = Rect{(uint32_t)r0, (uint32_t)(r0 + o.rows), (uint32_t)c0, (uint32_t)(c0 + o.row_bytes)};
    return true;
  }
  template <typename F> static void for_cells(const Rect &r, F f) {
    for (uint32_t cr = r.r0 / 64; cr <= (r.r1 - 1) / 64; ++cr)
      for (uint32_t cc = r.c0 / 256; cc <= (r.c1 - 1) / 256; ++cc) f(((uint64_t)cr << 32) | cc);
  }
  bool overlaps(const Operand &o, uintptr_t anchor) const {
    if (!o.ptr || !o.bytes) return false;
    const Range b = bounding(o);
    if (flat.overlaps(b)) return true;
    Rect r{0, 0, 0, 0};
    const bool is_rect = to_rect(o, anchor, r);
    for (const Plane &p : planes) {
      if (!p.bound.overlaps(b)) continue;
      if (!is_rect || p.anchor != anchor || p.pitch != o.pitch) return true;
      bool hit = false;
      for_cells(r, [&](uint64_t key) {
        auto range = p.cells.equal_range(key);
        for (auto it = range.first; it != range.second && !hit; ++it) {
          const Rect &q = it->second;
          hit = q.r0 < r.r1 && r.r0 < q.r1 && q.c0 < r.c1 && r.c0 < q.c1;
        }
      });
      if (hit) return true;
    }
    return false;
  }
  void insert(const Operand &o, uintptr_t anchor) {
    if (!o.ptr || !o.bytes) return;
    Rect r;
    if (!to_rect(o, anchor, r)) {
      flat.insert(bounding(o));
      return;
    }
    Plane *pl = nullptr;
    for (Plane &p : planes)
      if (p.anchor == anchor && p.pitch == o.pitch) pl = &p;
    if (!pl) {
      planes.push_back(Plane{anchor, o.pitch, {}, {}});
      pl = &planes.back();
    }
    pl->bound.insert(bounding(o));
    for_cells(r, [&](uint64_t key) { pl->cells.emplace(key, r); });
  }
};

// tile-queue counters (xsmm_hip_tile_queue_stats): launches, invokes queued with full bookkeeping / by replay, abandoned replays
std::atomic<int64_t> g_q_launches{0}, g_q_checked{0}, g_q_replayed{0}, g_q_abandoned{0}, g_q_terminated{0};
// (bumped only by whoever owns the queue state at that moment - the inline queue's lock holder or the scheduler thread: a
// plain load + store, not a locked read-modify-write on the enqueue path)
inline void bump(std::atomic<int64_t> &c) { c.store(c.load(std::memory_order_relaxed) + 1, std::memory_order_relaxed); }

// One queued invoke as the trace cache remembers it.
struct TraceItem {
  const void *desc = nullptr;
  WorkItem w{};
  hipStream_t stream = nullptr;
  bool same(const void *d, const WorkItem &x, hipStream_t s) const {
    return desc == d && w.A == x.A && w.B == x.B && w.C == x.C && w.D == x.D && w.br == x.br && stream == s;
  }
};
// A group as it was once collected: its invokes (in the order of that collection, and as a hash set) and the invokes that have
// been seen to end it by conflicting with it.
struct Segment {
  std::vector<TraceItem> items;
  std::vector<TraceItem> terminators;
  std::vector<uint32_t> seen; // round in which items[i] was last replayed (an invoke may join a group once); marked with atomic
                              // exchanges: callers mark their own arrivals while a direct window is open (DirectWindow)
  std::vector<int32_t> table; // open addressing over items, -1 = empty
  uint32_t round = 0;
  uint32_t n_items = 0; // = items.size() of a stored segment (build()): the per-invoke fast path compares an index, no division by sizeof(TraceItem)
  bool vec_ok = true, out_ok = true, pair_ok = true;
  uint64_t last_use = 0;
  // The group's work list as the grouped kernels read it: items[i].w in recorded order, in pinned host memory, written once when
  // the group is first replayed. A replay in which EVERY member arrives launches straight from it - nobody copies a work item.
  WorkItem *list = nullptr, *list_dev = nullptr; // ... and its copy in device memory (what the launches read: no PCIe round trip at the head of every workgroup)
  size_t list_cap = 0;
  bool list_valid = false, list_used = false; // holds items[] of THIS recording / a launch may still be reading it
  hipStream_t list_stream = nullptr;           // ... on this stream
  // GRID (round 5, detect_grid below): the group's gemm invokes tile ONE flat problem - a complete replay is then ONE launch of the
  // merged problem's own kernel. 0: not looked at yet, 1: grid_desc / grid_w hold the merged problem, -1: not a grid
  int grid_state = 0;
  const GemmDesc *grid_desc = nullptr;
  WorkItem grid_w{};
  // QUADS (round 6, detect_quads in rt_rewrites.h): the group's 64x64 bf16 invokes form an R x C grid of item rows and item columns
  // (R, C even) and the tile model prefers the 128x128 tile: a complete replay is ONE launch of R C / 4 2 x 2 blocks. 0: not looked
  // at yet, 1: quad_dev holds the blocks, -1: no. The buffers travel with the segment like the work list and follow its rules.
  int quad_state = 0, n_quads = 0;
  QuadItem *quad_host = nullptr, *quad_dev = nullptr;
  size_t quad_cap = 0;
  bool quad_used = false;
  hipStream_t quad_stream = nullptr;
  // Called with the inline queue's lock held, once per RECORDING (a steady-state replay never comes here). The buffers are sized
  // for the largest group (TileQueue::CAP) the first time a segment needs them and then travel with it (store_recording swaps
  // segments, so at most NSEG + 1 sets exist per queue: allocation is a start-up cost, not a per-recording one); they live as long
  // as the process (like the pinned work-list slots: no HIP call at exit). While the stream is being CAPTURED into a graph no list
  // is built (allocation / synchronisation are not legal there): the replay then gathers its members into a pinned slot at the
  // flush like an incomplete group (TileQueue::flush) - returns false.
  bool ensure_list(hipStream_t stream, size_t cap) {
    if (list_valid) return true;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess) (void)hipGetLastError();
    else if (cs != hipStreamCaptureStatusNone) return false;
    if (list_used) { // the buffers carried another recording's items: its last launch must be done (and, first, issued)
      launcher_drain();
      HIP_OK(hipStreamSynchronize(list_stream));
    }
    list_used = false;
    if (list_cap < items.size()) {
      if (list) HIP_OK(hipHostFree(list));
      if (list_dev) HIP_OK(hipFree(list_dev));
      list_cap = items.size() < cap ? cap : items.size();
      HIP_OK(hipHostMalloc((void **)&list, sizeof(WorkItem) * list_cap, hipHostMallocDefault));
      HIP_OK(hipMalloc((void **)&list_dev, sizeof(WorkItem) * list_cap));
    }
    for (size_t i = 0; i < items.size(); ++i) list[i] = items[i].w;
    HIP_OK(hipMemcpyAsync(list_dev, list, sizeof(WorkItem) * items.size(), hipMemcpyHostToDevice, stream));
    list_used = true; // (the copy reads `list`)
    list_stream = stream;
    list_valid = true;
    return true;
  }
  // Proof that every recorded pointer is device memory, per synchronisation epoch (DeviceRanges): the (few) allocations that hold
  // them, collected the first time the group is replayed and re-verified - same base, same extent - once per epoch by whoever
  // opens the group's window, under the queue's lock. A caller whose invoke matches a recorded member while dev_epoch is the
  // current epoch skips its own four range checks (a quarter of the lock-free path); without a proof it checks as before.
  static constexpr int MAX_ALLOC = 12;
  Range alloc[MAX_ALLOC];
  int n_alloc = -1;       // -1: not collected
  uint64_t dev_epoch = 0; // written under the lock before the window opens, read inside the window
  bool prove(DeviceRanges &dm, uint64_t epoch) {
    if (dev_epoch == epoch) return true;
    dev_epoch = 0;
    if (n_alloc >= 0) {
      bool same = true;
      for (int i = 0; i < n_alloc && same; ++i) {
        const Range r = dm.is_device((const void *)alloc[i].b) ? dm.range_of((const void *)alloc[i].b) : Range{0, 0};
        same = r.b == alloc[i].b && r.e == alloc[i].e;
      }
      if (same) {
        dev_epoch = epoch;
        return true;
      }
      n_alloc = -1; // an allocation went away or changed: collect again
    }
    int n = 0, last = 0;
    for (const TraceItem &t : items) {
      const void *ptrs[4] = {t.w.A, t.w.B, t.w.C, t.w.D};
      for (const void *q : ptrs) {
        if (!q) continue;
        const uintptr_t a = (uintptr_t)q;
        if (n && a >= alloc[last].b && a < alloc[last].e) continue;
        int j = 0;
        while (j < n && !(a >= alloc[j].b && a < alloc[j].e)) ++j;
        if (j == n) {
          if (n == MAX_ALLOC || !dm.is_device(q)) return false;
          const Range r = dm.range_of(q);
          if (!r.e) return false; // (device memory without an address range: not provable, the callers keep checking)
          alloc[n++] = r;
        }
        last = j;
      }
    }
    n_alloc = n;
    dev_epoch = epoch;
    return true;
  }
  bool mark(int idx) { return __atomic_exchange_n(&seen[idx], round, __ATOMIC_RELAXED) != round; } // false: joined this round already
  bool mark_solo(int idx) { // one caller in the whole process (DirectWindow, SOLO): nobody else marks
    if (__atomic_load_n(&seen[idx], __ATOMIC_RELAXED) == round) return false;
    __atomic_store_n(&seen[idx], round, __ATOMIC_RELAXED);
    return true;
  }
  static size_t hash(const WorkItem &w) {
    uint64_t h = (uint64_t)(uintptr_t)w.C * 0x9E3779B97F4A7C15ull;
    h ^= ((uint64_t)(uintptr_t)w.A >> 4) * 0xC2B2AE3D27D4EB4Full;
    h ^= ((uint64_t)(uintptr_t)w.B >> 4) * 0x165667B19E3779F9ull;
    return (size_t)(h ^ (h >> 29));
  }
  void build() {
    // by output address: membership is all a replay needs, and with the reference's static schedules a caller's invokes then sit
    // next to each other - its arrival marks in `seen` share cache lines with its own marks only, and "the one after my last" is
    // usually the next invoke (DirectWindow::Caller::hint) without a hash lookup
    std::stable_sort(items.begin(), items.end(), [](const TraceItem &a, const TraceItem &b) { return (uintptr_t)a.w.C < (uintptr_t)b.w.C; });
    size_t cap = 16;
    while (cap < 2 * items.size()) cap *= 2;
    table.assign(cap, -1);
    for (size_t i = 0; i < items.size(); ++i) {
      size_t at = hash(items[i].w) & (cap - 1);
      while (table[at] >= 0) at = (at + 1) & (cap - 1);
      table[at] = (int32_t)i;
    }
    seen.assign(items.size(), 0);
    n_items = (uint32_t)items.size();
    round = 0;
    list_valid = false;
    n_alloc = -1;
    dev_epoch = 0;
    grid_state = 0;
    quad_state = 0;
  }
  int index_of(const void *d, const WorkItem &w, hipStream_t st) const {
    if (table.empty()) return -1;
    const size_t mask = table.size() - 1;
    for (size_t at = hash(w) & mask; table[at] >= 0; at = (at + 1) & mask)
      if (items[table[at]].same(d, w, st)) return table[at];
    return -1;
  }
  bool is_terminator(const void *d, const WorkItem &w, hipStream_t st) const {
    for (const TraceItem &t : terminators)
      if (t.same(d, w, st)) return true;
    return false;
  }
};

inline void detect_grid(Segment &S); // rt_rewrites.h (GRID MERGE)
inline void detect_quads(Segment &S, hipStream_t stream); // rt_rewrites.h (QUADS)
static bool grid_merge_on();
extern std::atomic<const char *> g_last_merged;

// DIRECT WINDOW: replayed members arrive without a lock. While the inline queue replays a recorded group, `cur` names it
// (generation << 7 | segment index + 1) and a caller whose invoke is a member marks it in the segment (Segment::mark) and counts it
// in its OWN cache line - no work item is written (the segment's pinned list already holds it) and no line is shared between
// callers except `cur`, which changes once per group. Whoever has to change the queue state - a terminator, an invoke the cache
// does not know, a flush point - holds the queue's lock, CLOSES the window (cur = 0) and waits until no caller is inside it:
//   caller: busy = cur (seq_cst); re-read cur (seq_cst); ... mark, count ...; busy = 0 (release)
//   closer: cur = 0 (seq_cst); for every caller: wait until busy == 0 (seq_cst / acquire), then read its count
// a Dekker pair per caller: either the caller sees the closed window and takes the locked path, or the closer sees it busy and waits
// for its arrival to be complete. Invokes are processed synchronously on this path (when xsmm_*_invoke returns, the invoke is in the
// group or launched), so everything that happened before an invoke is in the queue state when it arrives: program order and every
// happens-before between callers hold without time stamps. Membership was proven conflict-free when the group was recorded.
//
// SOLO: as long as ONE thread is all the queue has ever seen (tpp-run without OpenMP, the reference's default), the caller's half of
// the Dekker pair is plain stores and loads and the arrival mark a load + store: the two locked instructions (xchg for the seq_cst
// store of `busy`, xchg for the mark) are 35-40 cycles of an invoke that costs ~100. The fence moves to the side that runs ONCE: the
// first time a second thread touches the queue state (claims a caller slot, or takes the queue's lock) it sets `multi`, issues
// membarrier(PRIVATE_EXPEDITED) - a full barrier on every CPU running a thread of this process - and waits until no solo section
// is in flight (`seq` of every caller even; a section brackets itself with seq++ ... seq++, and a section that read multi = false
// before the barrier had made its seq store by then: stores are not reordered with OLDER loads' retirement, an interrupt discards
// a load that ran ahead of an unretired store). From then on, for good, the protocol above. No membarrier (seccomp): never solo.
static uintptr_t thread_token() {
  static thread_local char t;
  return (uintptr_t)&t;
}
struct DirectWindow {
  static constexpr int MAXC = 256;
  struct alignas(64) Caller {
    std::atomic<uint64_t> busy{0};
    uint64_t tag = 0;   // window the count belongs to   (written inside the busy section, read by the closer after it)
    uint32_t count = 0; // arrivals in that window
    uint32_t hint = 0;  // index after this caller's last arrival
    std::atomic<int> owned{0};
    std::atomic<uint64_t> seq{0}; // odd while the owner is inside a SOLO section (written by the owner only, relaxed)
  };
  alignas(64) std::atomic<uint64_t> cur{0};
  alignas(64) std::atomic<int> ncallers{0}; // high-water mark of claimed caller slots
  std::atomic<bool> multi{true};             // false: SOLO
  std::atomic<bool> multi_ready{true};       // the switch to multi has completed (nobody is inside a solo section any more)
  std::atomic<uintptr_t> solo_owner{0};      // thread_token() of the one thread
  Caller callers[MAXC];
  DirectWindow() {
    const char *e = getenv("TPP_HIP_QUEUE_SOLO"); // 0: the two-sided protocol from the start (A/B runs)
    if ((!e || atoi(e) != 0) && !getenv("TPP_HIP_NO_MEMBARRIER") && syscall(__NR_membarrier, MEMBARRIER_CMD_REGISTER_PRIVATE_EXPEDITED, 0) == 0) {
      multi.store(false, std::memory_order_relaxed);
      multi_ready.store(false, std::memory_order_relaxed);
    }
  }
  // every entry to the queue state that is not a solo section (claiming a slot, taking the queue's lock) says who it is
  void touch(uintptr_t me) {
    if (multi.load(std::memory_order_acquire)) {
      while (!multi_ready.load(std::memory_order_acquire)) cpu_relax(); // (another thread is switching right now)
      return;
    }
    uintptr_t o = solo_owner.load(std::memory_order_acquire);
    if (o == me) return;
    if (o == 0 && solo_owner.compare_exchange_strong(o, me, std::memory_order_seq_cst)) return;
    bool expect = false;
    if (multi.compare_exchange_strong(expect, true, std::memory_order_seq_cst)) {
      if (syscall(__NR_membarrier, MEMBARRIER_CMD_PRIVATE_EXPEDITED, 0) != 0) die("tpp-xsmm-hip: membarrier failed");
      for (int i = 0; i < MAXC; ++i)
        while (callers[i].seq.load(std::memory_order_acquire) & 1) cpu_relax();
      multi_ready.store(true, std::memory_order_release);
    } else {
      while (!multi_ready.load(std::memory_order_acquire)) cpu_relax();
    }
  }
  Caller *claim() {
    touch(thread_token());
    const int n = ncallers.load(std::memory_order_acquire);
    for (int i = 0; i < MAXC; ++i) {
      int expect = 0;
      if (callers[i].owned.load(std::memory_order_relaxed) == 0 && callers[i].owned.compare_exchange_strong(expect, 1, std::memory_order_seq_cst)) {
        int hw = n;
        while (hw < i + 1 && !ncallers.compare_exchange_weak(hw, i + 1, std::memory_order_seq_cst)) {
        }
        return &callers[i];
      }
    }
    return nullptr; // more caller threads than slots: this one always takes the locked path
  }
};

struct TileQueue {
  static constexpr int CAP = 4096, SLOTS = 32, GROUP = 8; // work-list slots; one completion event per GROUP slots
  // TRACE CACHE. Compiled code repeats itself: the same handles on the same pointers in the same order, iteration after
  // iteration (the timing loop of tpp-run, every layer of a model). Whether a group of queued invokes is conflict-free,
  // and whether the next invoke conflicts with it, is a pure function of that sequence of (descriptor, pointers, batch)
  // - the footprints follow from them - so a group that was collected once with full bookkeeping is REPLAYED the next
  // time one of its invokes shows up on an empty queue: each following invoke that is a MEMBER of the recorded group
  // (compared with the next recorded one first, else looked up in the group's hash set) and has not joined in this
  // round is appended to the work list, nothing else; an invoke that has been seen to end the group launches it.
  // Membership, not order: a group is conflict-free iff its invokes are pairwise so, in any order and for any subset -
  // which is what several OpenMP callers produce, whose interleaving changes from iteration to iteration. Any other
  // invoke rebuilds the footprints of what has been queued and drops back to the full bookkeeping: if it conflicts, it
  // is remembered as one more terminator of the group; if it joins, the new group is recorded, and the cache is left
  // alone for a growing number of groups. Replaying a flush is always safe, skipping the checks is safe because the
  // same set was proven conflict-free.
  static constexpr size_t NSEG = 64, MIN_SEG = 16; // recorded groups kept (a 20-layer model repeats ~20 groups per iteration)
  std::vector<Segment> segs;
  int replay = -1;        // index of the segment being replayed
  size_t rpos = 0;        // the item expected next (a hint: the one after the last match)
  int learn = -1;         // a replay of this segment was just abandoned: if the invoke that did it conflicts, it is a terminator
  size_t learn_n = 0;     // ... provided the group still has this many invokes
  Segment rec;            // the group being recorded (full bookkeeping path)
  bool rec_open = false;
  uint64_t use_clock = 0;
  unsigned backoff = 0, backoff_next = 2; // groups to collect without consulting the cache / after the next mismatch
  int kind = 0;               // KIND_GEMM / KIND_UNARY / KIND_BINARY of the queued invokes
  const void *desc = nullptr; // their (single) descriptor
  bool vec_ok = true, out_ok = true, pair_ok = true;
  int n = 0;
  Footprint reads, writes;
  // Work lists live in host-pinned (device-mapped) memory and every workgroup reads its 40-byte item over PCIe, once,
  // at its head. Moving the list to HBM with one hipMemcpyAsync in front of each grouped launch was built and
  // measured (profiles/r02_tile_queue_device_lists.txt): the copy costs 15-20 us of host time per flush on this
  // runtime - the reference's headline pattern (3 flushes per iteration) went from 47 to 103 us - while the PCIe
  // read is ~1 us of latency that all workgroups pay in parallel.
  // A slot is reused SLOTS flushes later, once the launch that read it has finished. One event per flush cost ~2 us of host
  // time each (two of the 25 us of the headline bf16 pattern): the slots are used in groups of GROUP, ONE event is recorded
  // behind the last launch of a group, and it is waited for when the group is entered again - 24 launches later.
  WorkItem *pinned[SLOTS] = {};
  hipEvent_t done[SLOTS / GROUP] = {};
  bool used[SLOTS / GROUP] = {};
  hipStream_t gstream[SLOTS / GROUP] = {}; // the stream the group's launches went to
  int slot = 0;
  hipStream_t stream = nullptr;
  DirectWindow *dw = nullptr; // the inline queue's window (the scheduler thread's queue has none: its callers hand over through rings)
  uint64_t dw_gen = 0;
  bool window_open = false;
  int64_t direct_groups = 0;  // groups closed with lock-free arrivals in them
  // a grouped launch that has been decided but not issued: the whole recorded group, from its segment's list. Issued after the
  // NEXT group's window has been opened, so the other callers enter that group while this thread is inside hipLaunchKernel.
  struct Pending {
    bool armed = false;
    int kind = 0;
    const void *desc = nullptr;
    int seg = -1, n = 0;
    bool vec_ok = true, out_ok = true, pair_ok = true;
    hipStream_t stream = nullptr;
  } pending;
  TileQueue() { segs.reserve(NSEG); } // callers inside a direct window hold pointers into segs: it never reallocates

  // no caller is inside the window any more on return; the lock-free arrivals are added to n
  void close_window() {
    if (!window_open) return;
    window_open = false;
    const uint64_t c = dw->cur.load(std::memory_order_relaxed);
    dw->cur.store(0, std::memory_order_seq_cst);
    const int nc = dw->ncallers.load(std::memory_order_seq_cst);
    int arrived = 0;
    for (int i = 0; i < nc; ++i) {
      DirectWindow::Caller &k = dw->callers[i];
      while (k.busy.load(std::memory_order_seq_cst) != 0) cpu_relax();
      if (k.tag == c) arrived += (int)k.count;
    }
    if (arrived) {
      n += arrived;
      ++direct_groups;
      g_q_replayed.store(g_q_replayed.load(std::memory_order_relaxed) + arrived, std::memory_order_relaxed);
    }
  }
  void open_window(int seg) {
    if (!dw) return;
    ++dw_gen;
    window_open = true;
    dw->cur.store((dw_gen << 7) | (uint64_t)(seg + 1), std::memory_order_seq_cst);
  }
  // the members of the replayed group that have arrived, as a dense work list in pinned[slot] (window closed): a replay that ends
  // before every member has joined, or is abandoned
  void materialize() {
    const Segment &S = segs[replay];
    int k = 0;
    for (size_t i = 0; i < S.items.size(); ++i)
      if (__atomic_load_n(&S.seen[i], __ATOMIC_RELAXED) == S.round) pinned[slot][k++] = S.items[i].w;
    if (k != n) die("tpp-xsmm-hip: internal error: %d members marked, %d counted in a replayed group", k, n);
  }
  void issue_pending() {
    if (!pending.armed) return;
    pending.armed = false;
    Segment &S = segs[pending.seg];
    if (pending.kind == KIND_GEMM && S.grid_state == 0) detect_grid(S);
    LaunchReq r;
    r.stream = pending.stream;
    if (pending.kind == KIND_GEMM && S.grid_state != 1 && S.quad_state == 0) detect_quads(S, pending.stream);
    if (pending.kind == KIND_GEMM && S.grid_state == 1) {
      r.kind = -1;
      r.desc = S.grid_desc;
      r.w = S.grid_w;
      g_last_merged.store(S.grid_desc->trace, std::memory_order_relaxed);
    } else if (pending.kind == KIND_GEMM && S.quad_state == 1) {
      g_last_merged.store(nullptr, std::memory_order_relaxed);
      r.kind = -2;
      r.desc = pending.desc;
      r.list = (const WorkItem *)S.quad_dev;
      r.n = S.n_quads;
      r.br = S.items[0].w.br;
      S.quad_used = true;
      S.quad_stream = pending.stream;
    } else {
      g_last_merged.store(nullptr, std::memory_order_relaxed);
      r.kind = pending.kind;
      r.desc = pending.desc;
      r.list = S.list_dev;
      r.n = pending.n;
      r.vec_ok = pending.vec_ok, r.out_ok = pending.out_ok, r.pair_ok = pending.pair_ok;
      r.br = S.items[0].w.br;
      S.list_used = true; // (from the hand-over on: whoever rebuilds the list drains the launch thread, then the stream)
      S.list_stream = pending.stream;
    }
    // the inline queue's complete replays leave through the launch thread (rt_launcher.h): by value - the segment may be re-recorded
    // before the launch is issued, its list is not rewritten before ensure_list has drained both
    if (dw && launcher().push(r)) return;
    launcher_drain(); // (switched off a moment ago, or the scheduler thread's queue: behind whatever was handed over before)
    issue_launch(r);
  }

  void ensure_slot() {
    if (n != 0) return;
    const int g = slot / GROUP;
    if (slot % GROUP == 0 && used[g]) {
      HIP_OK(hipEventSynchronize(done[g])); // every launch that read a slot of this group has finished
      used[g] = false;
    }
    if (!pinned[slot]) HIP_OK(hipHostMalloc((void **)&pinned[slot], sizeof(WorkItem) * CAP, hipHostMallocDefault));
  }
  void launched() { // a grouped launch on `stream` has been issued from pinned[slot]
    const int g = slot / GROUP;
    if (slot % GROUP != 0 && gstream[g] != stream) HIP_OK(hipStreamSynchronize(gstream[g])); // (the caller changed streams inside a group)
    gstream[g] = stream;
    if (slot % GROUP == GROUP - 1) {
      if (!done[g]) HIP_OK(hipEventCreateWithFlags(&done[g], hipEventDisableTiming));
      HIP_OK(hipEventRecord(done[g], stream));
      used[g] = true;
    }
    slot = (slot + 1) % SLOTS;
  }
  void store_recording(const TraceItem *next) {
    if (learn >= 0 && next && rec_open && rec.items.size() == learn_n) {
      // the group is exactly what was replayed from segs[learn] and `next` conflicts with it: one more way that group ends
      Segment &S = segs[learn];
      if (S.terminators.size() < 64 && !S.is_terminator(next->desc, next->w, next->stream)) S.terminators.push_back(*next);
    } else if (rec_open && rec.items.size() >= MIN_SEG) {
      rec.terminators.clear();
      if (next) rec.terminators.push_back(*next);
      rec.vec_ok = vec_ok;
      rec.out_ok = out_ok;
      rec.pair_ok = pair_ok;
      rec.last_use = ++use_clock;
      rec.build();
      size_t at = segs.size();
      for (size_t i = 0; i < segs.size(); ++i)
        if (segs[i].index_of(rec.items[0].desc, rec.items[0].w, rec.items[0].stream) >= 0) at = i; // overlapping group: the newer one wins
      if (at == segs.size() && segs.size() >= NSEG) {
        at = 0;
        for (size_t i = 1; i < segs.size(); ++i)
          if (segs[i].last_use < segs[at].last_use) at = i;
      }
      if (at == segs.size()) segs.emplace_back();
      std::swap(segs[at], rec);
    }
    learn = -1;
    rec.items.clear();
    rec_open = false;
  }
  // the most recently used recorded group that contains the invoke; its index in *item
  int find_segment(const void *d, const WorkItem &w, hipStream_t s, int *item) {
    int best = -1;
    for (size_t i = 0; i < segs.size(); ++i) {
      const int idx = segs[i].index_of(d, w, s);
      if (idx >= 0 && (best < 0 || segs[i].last_use > segs[best].last_use)) {
        best = (int)i;
        *item = idx;
      }
    }
    return best;
  }
  // next: the invoke whose conflict ends this group (nullptr: an external flush point). defer: the launch may be left pending
  // (the caller opens the next group first and then calls issue_pending()).
  void flush(const TraceItem *next = nullptr, bool defer = false) {
    issue_pending();
    close_window();
    const int rp = replay;
    // the recorded group, complete, and its device-resident list exists (not while capturing): launch from its own list
    const bool whole = rp >= 0 && n > 0 && (size_t)n == segs[rp].items.size() && segs[rp].list_valid;
    if (rp >= 0 && n > 0 && !whole) materialize();
    // (counts are exact: an arrival is counted by whoever's atomic exchange on the item's mark saw it unmarked - once per round)
    store_recording(next); // (never touches segs[rp] during a replay: nothing is being recorded)
    replay = -1;
    if (n == 0) return;
    bump(g_q_launches);
    if (whole) {
      pending = Pending{true, kind, desc, rp, n, vec_ok, out_ok, pair_ok, stream};
      if (!defer) issue_pending();
    } else {
      launcher_drain(); // (launches handed to the launch thread come first on the stream)
      if (kind == KIND_GEMM) g_last_merged.store(nullptr, std::memory_order_relaxed);
      if (kind == KIND_GEMM) HIP_OK(launch_gemm_grouped(*(const GemmDesc *)desc, pinned[slot], n, vec_ok, out_ok, pair_ok, pinned[slot][0].br, stream));
      else if (kind == KIND_UNARY) HIP_OK(launch_unary_grouped(*(const UnaryDesc *)desc, pinned[slot], n, stream));
      else HIP_OK(launch_binary_grouped(*(const BinaryDesc *)desc, pinned[slot], n, stream));
      launched();
    }
    n = 0;
    desc = nullptr;
    vec_ok = out_ok = pair_ok = true;
    reads.clear();
    writes.clear();
  }
};

// What a caller hands over to the scheduler: descriptor + pointers of one invoke - or a fence (desc == nullptr,
// w.C = the flag to raise). 56 bytes: with the slot's sequence word ONE cache line crosses from the caller's core
// to the scheduler's per invoke (a 300-byte entry with the footprints resolved on the caller's side cost five, and
// made two callers 3x slower than one); the scheduler derives the footprints itself (queued_operands).
struct QEntry {
  const void *desc = nullptr;
  WorkItem w{};
  hipStream_t stream = nullptr;
};

// does the invoke conflict with the group being collected (another handle / stream, capacity, a data dependence)?
inline bool conflicts_with_group(const TileQueue &q, int kind, const void *desc, const Operand &out, uintptr_t anchor_out,
                                 const Operand *const *in, const uintptr_t *anchor_in, int n_in, hipStream_t stream) {
  if (q.n == 0) return false;
  if (q.kind != kind || q.desc != desc || q.stream != stream || q.n >= TileQueue::CAP) return true;
  if (q.writes.overlaps(out, anchor_out) || q.reads.overlaps(out, anchor_out)) return true;
  for (int i = 0; i < n_in; ++i)
    if (q.writes.overlaps(*in[i], anchor_in[i])) return true;
  return false;
}
// appends one invoke to the group being collected (full bookkeeping; the group is recorded for the trace cache)
inline void append_to_group(TileQueue &q, int kind, const void *desc, const WorkItem &w, const Operand &out, uintptr_t anchor_out,
                            const Operand *const *in, const uintptr_t *anchor_in, int n_in, bool vec_ok, bool out_ok,
                            bool pair_ok, hipStream_t stream) {
  q.ensure_slot();
  q.kind = kind;
  q.desc = desc;
  q.stream = stream;
  q.vec_ok = q.vec_ok && vec_ok;
  q.out_ok = q.out_ok && out_ok;
  q.pair_ok = q.pair_ok && pair_ok;
  if (q.n == 0) { // a new group: record it
    q.rec.items.clear();
    q.rec_open = true;
  }
  bump(g_q_checked);
  if (q.rec_open) q.rec.items.push_back(TraceItem{desc, w, stream});
  if (q.learn >= 0) { // the invoke that ended a replay joined the group: the caller has left the recorded pattern
    q.learn = -1;
    q.backoff = q.backoff_next;
    if (q.backoff_next < 64) q.backoff_next *= 2;
  }
  q.pinned[q.slot][q.n++] = w;
  for (int i = 0; i < n_in; ++i) q.reads.insert(*in[i], anchor_in[i]);
  q.writes.insert(out, anchor_out);
}
// the first invoke of a group on an empty queue: replay the recorded group it belongs to, if there is one
inline bool try_start_replay(TileQueue &q, DeviceRanges &devmem, const void *desc, const WorkItem &w, hipStream_t stream) {
  if (q.backoff > 0) {
    --q.backoff;
    return false;
  }
  int item = 0;
  const int idx = q.find_segment(desc, w, stream, &item);
  if (idx < 0) return false;
  Segment &S = q.segs[idx];
  if (++S.round == 0) { // (wrapped: forget the marks)
    std::fill(S.seen.begin(), S.seen.end(), 0u);
    S.round = 1;
  }
  S.seen[item] = S.round;
  q.ensure_slot();
  (void)S.ensure_list(stream, (size_t)TileQueue::CAP); // (false while the stream is being captured: the flush gathers the members instead)
  q.kind = *(const int *)desc;
  q.desc = desc;
  q.stream = stream;
  q.vec_ok = S.vec_ok;
  q.out_ok = S.out_ok;
  q.pair_ok = S.pair_ok;
  q.n = 1; // members are marked and counted, not copied: the work list is S.list (all of them) or is gathered at the flush
  q.replay = idx;
  q.rpos = (size_t)item + 1;
  S.last_use = ++q.use_clock;
  bump(g_q_replayed);
  if (q.dw) (void)S.prove(devmem, devmem.epoch); // (devmem belongs to the thread that runs this and is of the current epoch)
  q.open_window(idx); // from here on the other members may arrive without the lock
  return true;
}
// bookkeeping of one queued invoke: footprints from the descriptor, allocation bases ("anchors" of the 2-D planes) from
// `devmem` - the allocation cache of the thread that runs this (every operand was seen to be device memory by the caller)
__attribute__((always_inline)) inline void process_item(TileQueue &q, DeviceRanges &devmem, const void *desc, const WorkItem &w, hipStream_t stream) {
  QueuedOps o;
  queued_operands(desc, w, o);
  const Operand *in[3] = {&o.op[0], &o.op[1], &o.op[2]};
  uintptr_t anchor_in[3] = {0, 0, 0};
  auto anchor = [&](const Operand &x) -> uintptr_t {
    if (!x.rows || !x.ptr) return 0;
    if (uintptr_t b = devmem.base_of(x.ptr)) return b;
    (void)devmem.is_device(x.ptr); // first sight of this allocation on this thread in this epoch
    return devmem.base_of(x.ptr);
  };
  for (int i = 0; i < o.n_in; ++i) anchor_in[i] = anchor(o.op[i]);
  const Operand &out = o.op[o.out];
  const uintptr_t anchor_out = anchor(out);
  const int kind = *(const int *)desc;
  // strict mode: a group holds invokes of ONE alignment class and ONE batch count - the grouped launch takes its operand path from
  // the AND of the members' alignment flags and its chunk count from the first member, so a mixed group would make a member's kernel
  // depend on its neighbours
  const bool strict_break = q.n > 0 && kind == KIND_GEMM && q.kind == KIND_GEMM && cfg().strict.load(std::memory_order_relaxed) &&
                            (o.vec_ok != q.vec_ok || o.out_ok != q.out_ok || o.pair_ok != q.pair_ok || w.br != q.pinned[q.slot][0].br);
  if (strict_break || conflicts_with_group(q, kind, desc, out, anchor_out, in, anchor_in, o.n_in, stream)) {
    const TraceItem term{desc, w, stream};
    q.flush(&term);
    if (try_start_replay(q, devmem, desc, w, stream)) return; // the group this invoke starts has been collected before
  }
  append_to_group(q, kind, desc, w, out, anchor_out, in, anchor_in, o.n_in, o.vec_ok, o.out_ok, o.pair_ok, stream);
}

// footprints of the queued invokes into the (empty) read / write sets: a replay is being abandoned
inline void rebuild_footprints(TileQueue &q, DeviceRanges &devmem) {
  for (int i = 0; i < q.n; ++i) {
    QueuedOps o;
    queued_operands(q.desc, q.pinned[q.slot][i], o);
    for (int j = 0; j <= o.out; ++j) {
      const Operand &x = o.op[j];
      uintptr_t a = 0;
      if (x.rows && x.ptr && !(a = devmem.base_of(x.ptr))) {
        (void)devmem.is_device(x.ptr);
        a = devmem.base_of(x.ptr);
      }
      if (j == o.out) q.writes.insert(x, a);
      else q.reads.insert(x, a);
    }
  }
}
// One queued invoke: replayed from the trace cache if it belongs to the recorded group being replayed, else the full bookkeeping.
inline void submit_item(TileQueue &q, DeviceRanges &devmem, const void *desc, const WorkItem &w, hipStream_t stream) {
  if (q.replay >= 0) {
    Segment &S = q.segs[q.replay];
    int idx = -1;
    if (q.rpos < S.items.size() && S.items[q.rpos].same(desc, w, stream)) idx = (int)q.rpos;
    else idx = S.index_of(desc, w, stream);
    if (idx >= 0 && S.mark(idx)) {
      ++q.n;
      q.rpos = (size_t)idx + 1;
      bump(g_q_replayed);
      return;
    }
    if (idx < 0 && S.is_terminator(desc, w, stream)) {
      bump(g_q_terminated);
      q.flush(nullptr, true); // as seen before: this invoke conflicts with the group (replay ends, the queue is empty; the launch is
      q.backoff_next = 2;     // issued once the next group is open). A whole group replayed: the caller is repeating itself
    } else if (idx < 0 && (q.close_window(), (size_t)q.n == S.items.size()) && q.find_segment(desc, w, stream, &idx) >= 0) {
      // Every member of the recorded group has arrived and this invoke belongs to ANOTHER recorded group: the group is over
      // (nothing the cache knows could still join it) - launch it and replay the invoke's own group. Flushing early is always
      // safe; what this saves is learning one terminator per distinct first arriver of the next group: with several OpenMP
      // callers the first invoke of the next layer is a different tile every iteration, and every unknown one used to cost an
      // abandoned replay plus a growing back-off (2 of 10 runs of the 8-caller benchmark spent their timed iterations learning).
      bump(g_q_terminated);
      q.flush(nullptr, true);
      q.backoff_next = 2;
    } else if (idx >= 0 && (q.close_window(), (size_t)q.n == S.items.size())) {
      // Every member has arrived and this invoke is a member AGAIN: the caller runs the same group once more - the timing loop of a
      // single-layer benchmark (benchmarks/config/matmul/*.json, fc/*.json: tpp-run calls the one-layer kernel N times; round 5:
      // every iteration used to abandon its replay here, 2017 of 2020 groups, and the rebuilt bookkeeping made the run host-bound -
      // 7.4 us per iteration of 48 invokes against 5.3 for the GPU side). Flushing early is always safe; the invoke then starts the
      // replay of its group afresh below.
      bump(g_q_terminated);
      q.flush(nullptr, true);
      q.backoff_next = 2;
    } else { // neither a member nor a known terminator: make the bookkeeping catch up with what has been queued
      bump(g_q_abandoned);
      q.close_window();
      q.materialize();
      q.learn = q.replay;
      q.learn_n = (size_t)q.n;
      q.replay = -1;
      rebuild_footprints(q, devmem);
      q.rec.items.clear();
      for (int i = 0; i < q.n; ++i) q.rec.items.push_back(TraceItem{q.desc, q.pinned[q.slot][i], q.stream});
      q.rec_open = true;
    }
  }
  if (q.n == 0 && try_start_replay(q, devmem, desc, w, stream)) {
    q.issue_pending();
    return;
  }
  q.issue_pending();
  process_item(q, devmem, desc, w, stream);
}
