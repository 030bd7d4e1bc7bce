// Host-side context, device buffers and launch helpers behind the C ABI (include/p3r.h).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <exception>
#include <functional>
#include <map>
#include <memory>
#include <chrono>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/p3r.h"
#include "field.h"
#include "poseidon2.h"
#include "proof_layout.h"
#include "mmcs4.h"

namespace p3r {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  throw Error(code, buf);
}

// fn(begin, end) over [0, n) on a few host threads (the preparation's per-op maps: validation, preprocessed rows,
// the ALU matrix).  Chunks are contiguous and an error is reported for the LOWEST chunk that failed, so the message
// is the one the sequential loop gives whenever the first bad element is the only one in its chunk's prefix.
template <class Fn>
inline void host_parallel_for(size_t n, size_t min_chunk, Fn fn) {
  const size_t hw = std::max<size_t>(std::thread::hardware_concurrency(), 1);
  const size_t chunks = std::min<size_t>(std::min<size_t>(hw, 16), (n + min_chunk - 1) / std::max<size_t>(min_chunk, 1));
  if (chunks <= 1) { fn((size_t)0, n); return; }
  std::vector<std::exception_ptr> err(chunks);
  std::vector<std::thread> th;
  auto run = [&](size_t c) {
    try { fn(n * c / chunks, n * (c + 1) / chunks); } catch (...) { err[c] = std::current_exception(); }
  };
  for (size_t c = 1; c < chunks; ++c) th.emplace_back(run, c);
  run(0);
  for (auto& t : th) t.join();
  for (auto& e : err) if (e) std::rethrow_exception(e);
}

#define P3R_HIP(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      ::p3r::fail(e_ == hipErrorOutOfMemory ? P3R_ENOMEM : P3R_EHIP, "%s failed: %s",   \
                  #expr, hipGetErrorString(e_));                                        \
  } while (0)

// Device memory pool.  A prove allocates and drops dozens of multi-hundred-MB temporaries;
// hipMalloc/hipFree each cost a device-wide synchronisation plus page-table work, so freed
// blocks are kept in exact-size free lists and reused.  Safe without extra synchronisation
// because every user of a ctx enqueues on ONE stream: a block handed out again is only
// touched by work ordered after the work that used it before.  (Blocking copies therefore
// go through h2d_sync below, never through the NULL stream.)
struct DevPool;
// Every live pool of the process: a ctx that runs out of device memory trims the caches of ALL of them
// before giving up (several contexts on one GPU - concurrent provers, aggregation nodes - each keep
// their freed blocks; one must not fail with P3R_ENOMEM while its siblings hold unused gigabytes).
struct PoolRegistry {
  std::mutex mu;
  std::vector<DevPool*> pools;
  static PoolRegistry& get() {
    static PoolRegistry r;
    return r;
  }
};

struct DevPool {
  std::map<size_t, std::vector<void*>> free_lists;
  size_t cached_bytes = 0;
  std::mutex mu;  // a sibling pool's OOM path may trim this one from another thread
  DevPool() {
    std::lock_guard<std::mutex> g(PoolRegistry::get().mu);
    PoolRegistry::get().pools.push_back(this);
  }
  DevPool(const DevPool&) = delete;
  DevPool& operator=(const DevPool&) = delete;
  static void trim_all() {
    std::lock_guard<std::mutex> g(PoolRegistry::get().mu);
    for (DevPool* p : PoolRegistry::get().pools) p->trim();
  }
  static size_t round_up(size_t bytes) {
    const size_t g = bytes < (size_t(1) << 20) ? 256 : (size_t(1) << 20);
    return (bytes + g - 1) / g * g;
  }
  void* get(size_t bytes) {
    bytes = round_up(bytes);
    {
      std::lock_guard<std::mutex> g(mu);
      auto it = free_lists.find(bytes);
      if (it != free_lists.end() && !it->second.empty()) {
        void* p = it->second.back();
        it->second.pop_back();
        cached_bytes -= bytes;
        return p;
      }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipErrorOutOfMemory) {
      (void)hipGetLastError();
      trim_all();  // this pool's cache and every sibling's
      e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess)
      fail(e == hipErrorOutOfMemory ? P3R_ENOMEM : P3R_EHIP, "hipMalloc(%zu bytes) failed: %s", bytes,
           hipGetErrorString(e));
    return p;
  }
  // Free lists are kept sorted (lowest address at the back, handed out first): the same sequence of
  // allocations and releases then yields the same addresses proof after proof whatever the release
  // order was, which is what lets the job lists and pointer tables of a proof shape stay cached on
  // the device (p3r_ctx::const_tables) instead of being rebuilt and uploaded every time.
  void put(void* p, size_t bytes) {
    bytes = round_up(bytes);
    std::lock_guard<std::mutex> g(mu);
    auto& v = free_lists[bytes];
    v.insert(std::lower_bound(v.begin(), v.end(), p, std::greater<void*>()), p);
    cached_bytes += bytes;
  }
  // Frees every cached block.  Blocks are only cached after the work that used them was enqueued on the
  // owner's stream; hipFree synchronises the device, so freeing them here is safe from any thread.
  void trim() {
    std::lock_guard<std::mutex> g(mu);
    for (auto& kv : free_lists)
      for (void* p : kv.second) (void)hipFree(p);
    free_lists.clear();
    cached_bytes = 0;
  }
  ~DevPool() {
    {
      std::lock_guard<std::mutex> g(PoolRegistry::get().mu);
      auto& v = PoolRegistry::get().pools;
      v.erase(std::remove(v.begin(), v.end(), this), v.end());
    }
    trim();
  }
};
// One pool per ctx (a ctx owns one stream, which is what makes reuse safe).  The pool in use is
// selected per API call (thread-local); a buffer remembers the pool it came from and keeps it
// alive, so objects may outlive the call - and even the ctx - that created them.
inline std::shared_ptr<DevPool>& tls_pool() {
  thread_local std::shared_ptr<DevPool> p;
  return p;
}

// Development knobs (tile sizes, the pre-round-2 NTT passes, copy-engine fetches, Merkle partition): environment
// variables that select equality-tested alternatives of the shipped path.  They exist in the `knobs` build of the
// library only (-DP3R_TUNING_KNOBS: plonky3_recursion_amd/knobs/libp3r_hip.so, what tests/test_gpu_cpp_host.py and the
// tuning tools load); in the product build every knob reads as unset and the alternatives are dead code the
// compiler drops.
static inline const char* tuning_knob(const char* name) {
#ifdef P3R_TUNING_KNOBS
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// Host-side timeline of one proof (knobs build, P3R_HOST_TIMELINE=1): what the HOST does between a transcript round trip
// and the next launch - the part of a proof the GPU spends idle (tools/host_gaps.py shows the gaps, this shows their cause).
struct HostTimeline {
  std::vector<std::pair<const char*, int64_t>> marks;
  bool on = false;
};
inline HostTimeline& host_timeline() {
  static thread_local HostTimeline t;
  return t;
}
inline void host_mark(const char* tag) {
  HostTimeline& t = host_timeline();
  if (t.on) t.marks.emplace_back(tag, (int64_t)std::chrono::steady_clock::now().time_since_epoch().count());
}
inline void host_timeline_begin() {
  HostTimeline& t = host_timeline();
  t.on = tuning_knob("P3R_HOST_TIMELINE") != nullptr;
  t.marks.clear();
  host_mark("begin");
}
inline void host_timeline_dump() {
  HostTimeline& t = host_timeline();
  if (!t.on || t.marks.empty()) return;
  fprintf(stderr, "host timeline (us since begin, us since the mark before):\n");
  for (size_t i = 0; i < t.marks.size(); ++i)
    fprintf(stderr, "  %9.1f %8.1f  %s\n", (t.marks[i].second - t.marks[0].second) / 1e3,
            i ? (t.marks[i].second - t.marks[i - 1].second) / 1e3 : 0.0, t.marks[i].first);
  t.marks.clear();
}

// Blocking host->device / device->host copy ordered on the ctx stream.
inline hipError_t copy_sync(hipStream_t s, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, s);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(s);
}

// Fills and small copies of the proof path as kernels of our own instead of hipMemsetAsync / hipMemcpyAsync: the runtime's
// blit path costs more on the host per call than a kernel launch, which shows where the GPU has nothing queued - the first
// operation after a transcript round trip, the dozen memsets that open a circuit run.  Same-box A/B
// (profiles/r06/host_gaps.txt): headline 27.16 -> 27.07 ms, 2^16-row layer 4.09 -> 3.92 ms.  P3R_RUNTIME_COPIES=1 (knobs
// build) restores the runtime calls.
static __global__ void __launch_bounds__(256) k_fill_bytes(uint8_t* __restrict__ p, uint32_t pattern, size_t n) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nt = (size_t)gridDim.x * 256;
  const size_t mis = (16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15, head = mis < n ? mis : n;
  if (tid < head) p[tid] = (uint8_t)pattern;
  uint4* q = reinterpret_cast<uint4*>(p + head);
  const size_t nq = (n - head) >> 4;
  for (size_t i = tid; i < nq; i += nt) q[i] = make_uint4(pattern, pattern, pattern, pattern);
  const size_t done = head + (nq << 4);
  if (tid < n - done) p[done + tid] = (uint8_t)pattern;
}
// W: bytes per element (16 / 4 / 1, what the alignment of both ends allows)
template <class T>
static __global__ void __launch_bounds__(256) k_copy_small(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, size_t n) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nt = (size_t)gridDim.x * 256;
  const size_t nq = n / sizeof(T);
  T* d = reinterpret_cast<T*>(dst);
  const T* q = reinterpret_cast<const T*>(src);
  for (size_t i = tid; i < nq; i += nt) d[i] = q[i];
  const size_t done = nq * sizeof(T);
  if (tid < n - done) dst[done + tid] = src[done + tid];
}
inline bool runtime_copies() {
  static const bool on = tuning_knob("P3R_RUNTIME_COPIES") != nullptr;
  return on;
}
inline hipError_t fill_async(hipStream_t s, void* p, int byte, size_t bytes) {
  if (bytes == 0) return hipSuccess;
  host_mark("launch: fill");
  if (runtime_copies()) return hipMemsetAsync(p, byte, bytes, s);
  const uint32_t b = (uint32_t)byte & 0xFFu, pattern = b * 0x01010101u;
  const size_t blocks = std::min<size_t>((bytes / 16 + 255) / 256 + 1, 4096);
  hipLaunchKernelGGL(k_fill_bytes, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<uint8_t*>(p), pattern, bytes);
  return hipGetLastError();
}
// device-visible source (device memory, or pinned host memory by its device pointer) -> device memory
inline hipError_t copy_async_kernel(hipStream_t s, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return hipSuccess;
  host_mark("launch: copy");
  const uintptr_t both = reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src);
  const size_t w = (both & 15) == 0 ? 16 : (both & 3) == 0 ? 4 : 1;
  const size_t blocks = std::min<size_t>((bytes / w + 255) / 256 + 1, 4096);
  uint8_t* d = static_cast<uint8_t*>(dst);
  const uint8_t* q = static_cast<const uint8_t*>(src);
  if (w == 16) hipLaunchKernelGGL(k_copy_small<uint4>, dim3((unsigned)blocks), dim3(256), 0, s, d, q, bytes);
  else if (w == 4) hipLaunchKernelGGL(k_copy_small<uint32_t>, dim3((unsigned)blocks), dim3(256), 0, s, d, q, bytes);
  else hipLaunchKernelGGL(k_copy_small<uint8_t>, dim3((unsigned)blocks), dim3(256), 0, s, d, q, bytes);
  return hipGetLastError();
}

// Pinned staging ring for small host->device uploads (pointer tables, challenge powers, job
// descriptors) that must not stall the host: the source is copied into pinned memory, the
// transfer is enqueued on the ctx stream and the caller moves on.  A region is handed out again
// only after the ring wraps, and wrapping waits for the stream first.
struct HostStage {
  static constexpr size_t kBytes = size_t(4) << 20;  // the largest client is the query gather list (~0.4 MB)
  char* base = nullptr;
  char* base_dev = nullptr;  // the ring as the device sees it (the upload kernel reads it in place)
  size_t off = 0;
  HostStage() = default;
  HostStage(const HostStage&) = delete;
  HostStage& operator=(const HostStage&) = delete;
  ~HostStage() {
    if (base) (void)hipHostFree(base);
  }
  static constexpr size_t kWordBytes = 64;  // head of the ring: words() below, never reused by uploads
  hipError_t ensure() {
    if (base) return hipSuccess;
    off = kWordBytes;
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&base), kBytes, hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) return e;
    return hipHostGetDevicePointer(reinterpret_cast<void**>(&base_dev), base, 0);
  }
  // A few pinned words for small device->host results that are fetched asynchronously and read
  // after a later synchronisation of the stream (the circuit run's error word).
  hipError_t words(uint32_t** out) {
    hipError_t e = ensure();
    *out = reinterpret_cast<uint32_t*>(base);
    return e;
  }
  hipError_t upload(hipStream_t s, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return hipSuccess;
    if (bytes > kBytes / 4) return copy_sync(s, dst, src, bytes, hipMemcpyHostToDevice);
    {
      hipError_t e = ensure();
      if (e != hipSuccess) return e;
    }
    const size_t need = (bytes + 63) & ~size_t(63);
    if (off + need > kBytes) {
      hipError_t e = hipStreamSynchronize(s);
      if (e != hipSuccess) return e;
      off = kWordBytes;
    }
    std::memcpy(base + off, src, bytes);
    // (the launch publishes the host's stores: the ring is coherent memory, read over the link by one small kernel)
    hipError_t e = runtime_copies() ? hipMemcpyAsync(dst, base + off, bytes, hipMemcpyHostToDevice, s)
                                    : copy_async_kernel(s, dst, base_dev + off, bytes);
    off += need;
    return e;
  }
};

// Pinned landing area for the larger device->host results of a proof (opened values, query
// answers): the transfer goes straight into it and the host reads it in place.  A fetch waits for
// the stream; its result is valid until the next fetch.
struct HostLanding {
  char* base = nullptr;
  size_t cap = 0;
  HostLanding() = default;
  HostLanding(const HostLanding&) = delete;
  HostLanding& operator=(const HostLanding&) = delete;
  ~HostLanding() {
    if (base) (void)hipHostFree(base);
  }
  hipError_t fetch(hipStream_t s, const void* dev, size_t bytes, const uint32_t** out) {
    if (bytes > cap) {
      if (base) (void)hipHostFree(base);
      base = nullptr;
      cap = 0;
      const size_t want = std::max(bytes * 2, size_t(1) << 20);
      hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&base), want, hipHostMallocDefault);
      if (e != hipSuccess) return e;
      cap = want;
    }
    *out = reinterpret_cast<const uint32_t*>(base);
    if (bytes == 0) return hipStreamSynchronize(s);
    return copy_sync(s, base, dev, bytes, hipMemcpyDeviceToHost);
  }
};

// Small device->host results that the host transcript waits for (a commitment cap, the final
// polynomial, a proof-of-work witness): a one-workgroup kernel on the ctx stream stores the words
// straight into coherent pinned host memory and then a sequence number; the host polls that word
// instead of going through a copy engine and hipStreamSynchronize (29 us per fetch measured with
// tools/microbench/d2h_latency.py - more than the Merkle levels of a small layer's commitment).
// The stream is NOT synchronised when post() returns: only this kernel's stores are known to be done.
static __global__ void __launch_bounds__(256) k_post_small(uint32_t* host_dst, const uint32_t* __restrict__ src, uint32_t n,
                                                    uint32_t* flag, uint32_t seq) {
  for (uint32_t i = threadIdx.x; i < n; i += 256) host_dst[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
struct HostPost {
  static constexpr size_t kWords = 16384;  // 64 KB of results + the sequence word behind them
  uint32_t* host = nullptr;
  uint32_t* dev = nullptr;
  uint32_t seq = 0;
  // how long the k-th fetch of a proof waited the last time (see post()); begin_proof() resets k and, when the proof's
  // shape is another one, the predictions
  static constexpr int kSlots = 48;
  int slot = 0;
  bool in_proof = false;   // fetches outside a proof (commit / open API calls, preparation) are not predicted
  uint64_t shape = 0;
  int64_t expect_ns[kSlots] = {};
  void begin_proof(uint64_t shape_key) {
    slot = 0;
    in_proof = true;
    if (shape_key != shape) { shape = shape_key; std::fill(expect_ns, expect_ns + kSlots, 0); }
  }
  void end_proof() { in_proof = false; }
  HostPost() = default;
  HostPost(const HostPost&) = delete;
  HostPost& operator=(const HostPost&) = delete;
  ~HostPost() {
    if (host) (void)hipHostFree(host);
  }
  static bool enabled() {
    static const bool off = tuning_knob("P3R_NO_POLLED_FETCH") != nullptr;
    return !off;
  }
  // `words` <= kWords cells from `src` (device) -> *out (valid until the next post)
  hipError_t post(hipStream_t s, const uint32_t* src, size_t words, const uint32_t** out) {
    if (!host) {
      hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&host), (kWords + 16) * 4,
                                   hipHostMallocMapped | hipHostMallocCoherent);
      if (e != hipSuccess) return e;
      e = hipHostGetDevicePointer(reinterpret_cast<void**>(&dev), host, 0);
      if (e != hipSuccess) {  // never leave a half-initialised pair behind: the next call would launch with dev == NULL
        (void)hipHostFree(host);
        host = nullptr;
        dev = nullptr;
        return e;
      }
      host[kWords] = 0;
    }
    *out = host;
    ++seq;
    hipLaunchKernelGGL(k_post_small, dim3(1), dim3(256), 0, s, dev, src, (uint32_t)words, dev + kWords, seq);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    volatile uint32_t* flag = host + kWords;
    // How to wait.  The store normally lands within microseconds (small layers: the host spins).  A 2^20-row proof waits
    // 3 - 15 ms per round trip; a thread that sleeps through that in short naps is woken from an idle core when the result
    // arrives, and the GPU sits idle for the 50 - 100 us that takes - six times per proof (profiles/r06/host_gaps.txt).
    // The k-th round trip of a proof takes what it took in the proof before (same shape, same kernels): sleep through
    // most of THAT, then poll awake.  A wait that outlasts its prediction (another shape, sibling provers on the GPU)
    // falls back to naps; the first proof of a shape has no prediction and naps as before.
    static const int mode = tuning_knob("P3R_POST_MODE") ? atoi(tuning_knob("P3R_POST_MODE")) : 0;  // 1: naps only, 2: spin only
    const int sl = slot < kSlots ? slot : kSlots - 1;
    if (in_proof) ++slot;
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed_ns = [&] { return (int64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); };
    const int64_t expect = mode == 0 && in_proof ? expect_ns[sl] : 0;
    if (expect > 400000) {
      // asleep for the first three quarters (in pieces, so that a result that comes early is not slept through for long)
      const int64_t until = expect - std::max<int64_t>(expect / 4, 200000);
      for (int64_t now = elapsed_ns(); now < until && *flag != seq; now = elapsed_ns())
        std::this_thread::sleep_for(std::chrono::nanoseconds(std::min<int64_t>(until - now, 500000)));
    }
    const int64_t awake_until = expect > 400000 ? expect + expect / 2 + 500000 : 0;   // then: naps
    for (uint64_t spins = 0;; ++spins) {
      if (*flag == seq) break;
      if (mode != 2 && spins > (uint64_t(1) << 14) && (spins & 0x3F) == 0x3F) {
        if (awake_until && elapsed_ns() < awake_until) std::this_thread::yield();
        else if (spins > (uint64_t(1) << 17)) std::this_thread::sleep_for(std::chrono::microseconds(20));
        else std::this_thread::yield();
      }
      if ((spins & 0xFFF) == 0xFFF) {
        // a fault upstream would leave the flag unset for ever: ask the stream now and then
        e = hipStreamQuery(s);
        if (e != hipErrorNotReady) {
          if (e != hipSuccess) return e;
          if (*flag == seq) break;
          return hipStreamSynchronize(s);  // idle stream without the store: cannot happen; do not spin on it
        }
      }
    }
    if (in_proof) expect_ns[sl] = elapsed_ns();
    std::atomic_thread_fence(std::memory_order_acquire);
    return hipSuccess;
  }
};

// RAII device buffer of u32 cells (pooled).
struct DevBuf {
  uint32_t* p = nullptr;
  size_t n = 0;
  std::shared_ptr<DevPool> pool;
  DevBuf() = default;
  explicit DevBuf(size_t cells) { alloc(cells); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n), pool(std::move(o.pool)) { o.p = nullptr; o.n = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; pool = std::move(o.pool); o.p = nullptr; o.n = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void alloc(size_t cells) {
    release();
    if (cells) {
      pool = tls_pool();
      if (!pool) fail(P3R_EINVAL, "device allocation outside a p3r API call");
      p = static_cast<uint32_t*>(pool->get(cells * sizeof(uint32_t)));
    }
    n = cells;
  }
  void release() {
    if (p) pool->put(p, n * sizeof(uint32_t));
    p = nullptr;
    n = 0;
    pool.reset();
  }
};

inline int log2_exact(size_t x, const char* what) {
  if (x == 0 || (x & (x - 1))) fail(P3R_EINVAL, "%s (%zu) must be a power of two", what, x);
  int l = 0;
  while ((size_t(1) << l) < x) ++l;
  return l;
}

}  // namespace p3r

// Device matrix: column-major, Montgomery form. `borrowed` views do not own storage.
struct p3r_dmat {
  p3r::DevBuf buf;
  uint32_t* d = nullptr;
  size_t h = 0, w = 0;
};

// Device-resident batch of Poseidon2CircuitRow main-trace fields.
struct p3r_p2_dev {
  size_t n = 0;
  std::unique_ptr<p3r_dmat> inputs;  // n x 16
  p3r::DevBuf flags;                 // bytes: new_start[n] | merkle_path[n] | mmcs_bit[n]
  p3r::DevBuf seed;                  // mmcs_index_sum[n], Montgomery
};

struct p3r_tree {
  // matrices in COMMIT order (as passed by the caller)
  std::vector<const p3r_dmat*> mats;
  std::vector<std::unique_ptr<p3r_dmat>> owned;  // when committed from host matrices
  // MerkleTreeHidingMmcs (p3r_config.mmcs_salt_elems > 0): `mats` then holds every committed matrix FOLLOWED by its salt
  // matrix (height x salt_elems, owned) - [M0, S0, M1, S1, ..] - so that the leaf preimage of a height class, built by the
  // plain commit in tallest-first stable order, is the concatenation of [row | salt] per matrix
  // (recursion/src/pcs/mmcs.rs:375-389); an opening is (rows of the even entries, salts = rows of the odd ones, siblings).
  int salt_elems = 0;
  std::vector<std::unique_ptr<p3r_dmat>> salt_owned;
  // FRI commit-phase tree of a hiding ExtensionMmcs: the salts of its one (strided) leaf matrix, column c of row r at
  // [c * salt_stride * rows + r * salt_stride] (the layout of the strided leaf kernels)
  p3r::DevBuf phase_salts;
  size_t phase_salt_stride = 0, phase_rows = 0;
  int log_max_h = 0;
  int cap_height = 0;
  size_t total_width = 0;
  // layers[l]: digests of layer l (layer 0 = leaves), SoA [8][n_l], n_l = 2^(log_max_h - l)
  std::vector<p3r::DevBuf> layers;
  // arity-4 MMCS (p3r_config.mmcs_arity = 4; mmcs4.h): levels[l] produces layers[l + 1]; n_l = layer_n[l] (a layer
  // of 2 is padded to 4 with zero digests).  Empty for the binary tree.
  int arity = 2;
  std::vector<p3r::Mmcs4Level> levels;
  std::vector<size_t> layer_n;
};

namespace p3r {
struct ProfRec {
  const char* name;
  hipEvent_t a, b;
};
}  // namespace p3r

struct p3r_ctx {
  p3r_config cfg{};
  std::shared_ptr<p3r::DevPool> pool = std::make_shared<p3r::DevPool>();
  bool prof_enabled = false;
  std::vector<p3r::ProfRec> prof;
  // wall-clock per prove stage (only while profiling; each mark synchronises the stream)
  std::vector<std::pair<std::string, double>> stage_ms;
  std::string cur_stage;
  double cur_stage_t0 = 0;
  int partial_rounds = 0;
  hipStream_t stream = nullptr;
  // second stream of the commits (prove_impl.hip.h::lde_and_commit): the leaf hashing of one height class runs on it
  // while the main stream extends the next class; joined before the Merkle levels.  Nothing else uses it.
  hipStream_t stream2 = nullptr;
  hipStream_t stream2_low = nullptr;   // the same at the lowest priority (A/B: P3R_COMMIT_OVERLAP_MODE = 2)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int n_cus = 256;  // compute units of the device (grids of the persistent kernels)
  p3r::DevBuf rc;  // Poseidon2 constants, Montgomery
  p3r::DevBuf rc_f64;  // the same constants as canonical doubles (poseidon2_f64.hip.h)
  const double* rcd() const { return reinterpret_cast<const double*>(rc_f64.p); }
  // the FP64 table of the width-32 permutation (poseidon2_w32_f64.hip.h), after the width-16 constants
  size_t rcd_w32_at = 0;
  const double* rcd_w32() const { return rcd() + rcd_w32_at; }
  p3r::DevBuf p2_diag;  // internal-layer diagonal, Montgomery (lane-cooperative kernels)
  std::vector<uint32_t> rc_canonical;
  std::vector<uint8_t> fri_log_arities;  // copy of p3r_config.fri_log_arities (empty: the rule)
  p3r::ProofLayout proof_layout;         // from p3r_config.proof_layout (identity by default)
  std::vector<uint32_t> rc_mont_host;  // Montgomery copy on the host (the prover's out-of-domain self-check)
  std::string err;
  bool w32_diag_builtin = false;  // the width-32 diagonal takes the per-lane forms (poseidon2_w32_f64.hip.h)
  bool w32_unacknowledged = false;  // width-32 constants defaulted without P3R_EXT_UNPINNED_W32_DEFAULTS (p3r.h)
  uint64_t zk_nonce = 0;  // proofs made so far under a ZK configuration (p3r_zk_nonce; zk_rand.h)
  uint32_t zk_key[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // the key the context draws with (zk_rand.h::zk_context_key)
  std::vector<uint8_t> pending_proof;  // the proof of a prove call whose buffer was too small (p3r_take_proof)
  p3r::HostStage stage;  // small uploads that do not wait (see HostStage)
  p3r::HostLanding landing;  // device->host results read in place (see HostLanding)
  p3r::HostPost post;        // small device->host results the host polls for (see HostPost)
  // Small read-only device tables (column pointers and job lists of the row-hash kernels),
  // keyed by their content: the pool hands the same addresses to the same allocation sequence,
  // so after the first proof of a shape every table is already on the device.
  std::map<std::string, p3r::DevBuf> const_tables;

  // NTT table caches (device), keyed by log size / direction / shift.
  std::map<std::pair<int, int>, p3r::DevBuf> tw_sub;                 // (log_r, inverse)
  std::map<std::pair<int, int>, std::pair<p3r::DevBuf, p3r::DevBuf>> tw4;  // (log_n, inverse) -> (lo, hi)
  std::map<std::tuple<int, int, uint32_t>, std::pair<p3r::DevBuf, p3r::DevBuf>> pre;  // (log_n, added_bits, shift)
};

// `words` cells from device memory into `dst` (host), for results the host transcript waits on.
inline hipError_t fetch_small(p3r_ctx* ctx, const uint32_t* src, size_t words, uint32_t* dst) {
  if (words == 0) return hipSuccess;
  if (p3r::HostPost::enabled() && words <= p3r::HostPost::kWords) {
    const uint32_t* got = nullptr;
    p3r::host_mark("fetch: posted, waiting for the GPU");
    hipError_t e = ctx->post.post(ctx->stream, src, words, &got);
    p3r::host_mark("fetch: arrived");
    if (e != hipSuccess) return e;
    std::memcpy(dst, got, words * 4);
    return hipSuccess;
  }
  return p3r::copy_sync(ctx->stream, dst, src, words * 4, hipMemcpyDeviceToHost);
}

