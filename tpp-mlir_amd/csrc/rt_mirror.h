// rt_mirror.h - host operands: device-pointer classification, the per-invoke mirror (H2D / kernel / D2H), host residents
// One of the subsystem units of runtime.cpp (round 6, VERDICT r5 next 7: the 3 000-line file split by subsystem, no behaviour
// change). The units are INCLUDED into the one translation unit runtime.cpp, in dependence order, inside its anonymous namespace:
// the per-invoke host path (14-18 ns: enqueue_item -> join_window -> Segment::mark) crosses four of them and is inlined across
// their borders - as separate objects without LTO it would pay a call per border. Not a stand-alone header: include runtime.cpp's way only.
// (needs rt_core.h; used by rt_registry.h's neighbours through Operand / DeviceRanges)

// ---- device scratch for mirroring host operands (per thread, grow only) -------------
struct Arena {
  char *base = nullptr;
  size_t cap = 0, used = 0;
  char *alloc(size_t bytes) {
    used = (used + 255) & ~size_t(255);
    char *p = base + used;
    used += bytes;
    return p;
  }
  void reserve(size_t bytes, hipStream_t s) {
    used = 0;
    if (bytes <= cap) return;
    if (base) {
      HIP_OK(hipStreamSynchronize(s));
      HIP_OK(hipFree(base));
    }
    cap = std::max(bytes, cap * 2);
    HIP_OK(hipMalloc((void **)&base, cap));
  }
};
thread_local Arena t_arena;

bool is_device_ptr(const void *p) {
  if (!p) return true; // nothing to mirror
  hipPointerAttribute_t attr;
  memset(&attr, 0, sizeof(attr));
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void)hipGetLastError(); // plain malloc'd memory on older runtimes: invalid value
    return false;
  }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged || attr.type == hipMemoryTypeArray;
}

struct Range {
  uintptr_t b, e;
};

// Device allocations seen so far ([base, base+size) from hipMemGetAddressRange), one cache per calling thread.
// Callers issue hundreds of invokes per layer on the same few allocations, and one driver query per operand
// per invoke would dominate the host time (~1 us each). The epoch is bumped at the explicit synchronisation
// points (xsmm_hip_synchronize, perf_stop_timer): the caller may free and re-allocate buffers after those, so
// cached ranges are only trusted within one epoch (and never in synchronous mode, see stage_in).
std::atomic<uint64_t> g_devmem_epoch{1};

struct DeviceRanges {
  std::vector<Range> known;
  uint64_t epoch = 0;
  bool refresh() { // true: a new epoch began (first use on this thread since the last synchronisation point)
    const uint64_t e = g_devmem_epoch.load(std::memory_order_relaxed);
    if (e == epoch) return false;
    known.clear();
    epoch = e;
    return true;
  }
  // index of the last hit PER OPERAND POSITION (A, B, C, D of consecutive invokes each stay in their own allocation;
  // one shared index would miss on every operand and fall into the scan)
  mutable size_t mru[4] = {0, 0, 0, 0};
  bool contains(const void *p, int pos = 0) const {
    const uintptr_t a = (uintptr_t)p;
    size_t &m = mru[pos & 3];
    if (m < known.size() && a >= known[m].b && a < known[m].e) return true;
    for (size_t i = 0; i < known.size(); ++i)
      if (a >= known[i].b && a < known[i].e) {
        m = i;
        return true;
      }
    return false;
  }
  Range range_of(const void *p) const { // the allocation that holds p, {0, 0} if unknown
    const uintptr_t a = (uintptr_t)p;
    for (const Range &r : known)
      if (a >= r.b && a < r.e) return r;
    return Range{0, 0};
  }
  uintptr_t base_of(const void *p) const { // allocation base, 0 if unknown
    const uintptr_t a = (uintptr_t)p;
    for (const Range &r : known)
      if (a >= r.b && a < r.e) return r.b;
    return 0;
  }
  bool is_device(const void *p, int pos = 0) {
    if (!p || contains(p, pos)) return true;
    if (!is_device_ptr(p)) return false;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) == hipSuccess && size) {
      if (known.size() >= 64) known.erase(known.begin());
      known.push_back(Range{(uintptr_t)base, (uintptr_t)base + size});
    } else {
      (void)hipGetLastError();
    }
    return true;
  }
};

// One operand of an invoke: [ptr, ptr + bytes), read and/or written by the kernel.
struct Operand {
  void *ptr;
  size_t bytes;
  bool written;
  void *dev; // resolved device pointer
  // optional 2-D shape of the footprint (rows of row_bytes every pitch bytes); 0 = one flat range.
  // Neighbouring tiles of one row-major buffer have interleaved rows, so their bounding ranges overlap
  // although the tiles do not: the tile queue's dependence tracking and the host mirror (which must
  // copy back ONLY the bytes the kernel writes) both work on this shape.
  size_t rows = 0, row_bytes = 0, pitch = 0;
  bool read = true;   // the kernel reads it (false: pure outputs, e.g. C under BETA_0)
  bool host = false;  // set by stage_in: the operand is host memory and `dev` points into a mirror
  void shape(int64_t r, size_t rb, size_t p) {
    if (r > 1 && p > rb) rows = (size_t)r, row_bytes = rb, pitch = p;
  }
};

// ---- host residents (extension): host buffers the harness declares stable -----------------------------
// The reference's callers pass host pointers and the ABI has no allocation / free hook, so a mirror can
// never be cached behind the caller's back (a freed and re-allocated range would alias a stale copy).
// A harness that knows a host buffer is long-lived (weights, inputs of a timing loop) can say so:
// xsmm_hip_host_resident(ptr, bytes) uploads it once and keeps a device copy; invokes whose operands lie
// inside a resident range use that copy without any upload (written operands are still copied back, so
// the host view stays current); xsmm_hip_host_update(ptr) re-uploads after the host changed the buffer;
// xsmm_hip_host_release(ptr) drops it.
struct Resident {
  char *host;
  size_t bytes;
  char *dev;
};
std::mutex g_res_mu;
std::vector<Resident> g_residents;
std::atomic<int> g_n_residents{0};

char *resident_dev(const void *p, size_t bytes) {
  if (!g_n_residents.load(std::memory_order_acquire)) return nullptr;
  std::lock_guard<std::mutex> lk(g_res_mu);
  for (const Resident &r : g_residents)
    if ((const char *)p >= r.host && (const char *)p + bytes <= r.host + r.bytes) return r.dev + ((const char *)p - r.host);
  return nullptr;
}

// pinned staging for the copy-back of small strided tiles (per thread, grow only)
struct Staging {
  char *base = nullptr;
  size_t cap = 0;
  char *get(size_t bytes) {
    if (bytes > cap) {
      if (base) HIP_OK(hipHostFree(base));
      cap = std::max(bytes, cap * 2);
      HIP_OK(hipHostMalloc((void **)&base, cap, hipHostMallocDefault));
    }
    return base;
  }
};
thread_local Staging t_staging;

// Resolve every operand to a device pointer. Device memory is used in place. Host operands are mirrored:
// overlapping host ranges (in-place relu, binary with out == lhs) share one mirror allocation, every
// operand the kernel READS is uploaded with its own shape (rows x row_bytes at the host pitch - the mirror
// keeps the host layout, gaps are never touched), pure outputs are not uploaded at all (C under BETA_0).
void stage_in(std::vector<Operand *> &ops, hipStream_t s) {
  std::vector<Operand *> host_ops;
  // async mode: device allocations seen in this synchronisation epoch cost one driver query each. In the
  // (default) synchronous mode every invoke is a point after which the caller may free buffers: query each time.
  thread_local DeviceRanges devmem;
  if (cfg().async.load(std::memory_order_relaxed)) devmem.refresh();
  else devmem.known.clear();
  for (Operand *o : ops) {
    o->host = false;
    if (!o->ptr || o->bytes == 0 || devmem.is_device(o->ptr)) o->dev = o->ptr;
    else host_ops.push_back(o);
  }
  if (host_ops.empty()) return;
  struct Span {
    char *host;
    size_t bytes;
    char *dev;
  };
  std::vector<Span> spans;
  std::vector<Operand *> mirrored;
  for (Operand *o : host_ops) {
    o->host = true;
    if (char *d = resident_dev(o->ptr, o->bytes)) o->dev = d, o->read = false; // device copy is current: nothing to upload
    else mirrored.push_back(o);
  }
  std::sort(mirrored.begin(), mirrored.end(), [](Operand *a, Operand *b) { return a->ptr < b->ptr; });
  for (Operand *o : mirrored) {
    char *b = (char *)o->ptr;
    if (!spans.empty() && b < spans.back().host + spans.back().bytes)
      spans.back().bytes = std::max(spans.back().bytes, (size_t)(b + o->bytes - spans.back().host));
    else spans.push_back({b, o->bytes, nullptr});
  }
  size_t total = 0;
  for (Span &m : spans) total += m.bytes + 512;
  t_arena.reserve(total, s);
  for (Span &m : spans) // keep the host address's offset within 256 B so alignment-dependent kernel choices see the caller's real alignment
    m.dev = t_arena.alloc(m.bytes + 256) + (((uintptr_t)m.host) & 255);
  for (Operand *o : mirrored)
    for (Span &m : spans)
      if ((char *)o->ptr >= m.host && (char *)o->ptr < m.host + m.bytes) {
        o->dev = m.dev + ((char *)o->ptr - m.host);
        break;
      }
  for (Operand *o : mirrored) {
    // an operand that is written AND overlaps a read operand (in-place ops) is covered by that operand's upload
    if (!o->read) continue;
    if (o->rows && o->rows * o->row_bytes * 2 < o->bytes)
      HIP_OK(hipMemcpy2DAsync(o->dev, o->pitch, o->ptr, o->pitch, o->row_bytes, o->rows, hipMemcpyHostToDevice, s));
    else
      HIP_OK(hipMemcpyAsync(o->dev, o->ptr, o->bytes, hipMemcpyHostToDevice, s));
  }
}

// Copy back what the kernel wrote - and only that: rows x row_bytes of a strided tile, never the gap bytes
// between its rows (they belong to neighbouring tiles other threads may be writing right now). Small tiles go
// through a pinned staging buffer + row-wise memcpy on this thread; large ones through hipMemcpy2DAsync.
void finish(std::vector<Operand *> &ops, hipStream_t s) {
  bool any_host = false;
  struct Late {
    Operand *o;
    char *stage;
  };
  Late late[4];
  int n_late = 0;
  size_t stage_bytes = 0;
  for (Operand *o : ops)
    if (o->host && o->written && o->rows && o->bytes <= (1u << 20)) stage_bytes += o->bytes;
  char *stage = stage_bytes ? t_staging.get(stage_bytes) : nullptr;
  for (Operand *o : ops) {
    if (!o->host) continue;
    any_host = true;
    if (!o->written) continue;
    if (!o->rows) {
      HIP_OK(hipMemcpyAsync(o->ptr, o->dev, o->bytes, hipMemcpyDeviceToHost, s));
    } else if (o->bytes <= (1u << 20) && n_late < 4) {
      HIP_OK(hipMemcpyAsync(stage, o->dev, o->bytes, hipMemcpyDeviceToHost, s));
      late[n_late++] = Late{o, stage};
      stage += o->bytes;
    } else {
      HIP_OK(hipMemcpy2DAsync(o->ptr, o->pitch, o->dev, o->pitch, o->row_bytes, o->rows, hipMemcpyDeviceToHost, s));
    }
  }
  if (any_host || !cfg().async.load(std::memory_order_relaxed)) HIP_OK(hipStreamSynchronize(s));
  for (int i = 0; i < n_late; ++i)
    for (size_t r = 0; r < late[i].o->rows; ++r)
      memcpy((char *)late[i].o->ptr + r * late[i].o->pitch, late[i].stage + r * late[i].o->pitch, late[i].o->row_bytes);
}
