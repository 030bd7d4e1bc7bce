// Host-side context, device buffers and launch helpers behind the C ABI (include/p3r.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/p3r.h"
#include "field.h"
#include "poseidon2.h"

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

#define P3R_HIP(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      ::p3r::fail(e_ == hipErrorOutOfMemory ? P3R_ENOMEM : P3R_EHIP, "%s failed: %s",   \
                  #expr, hipGetErrorString(e_));                                        \
  } while (0)

// RAII device buffer of u32 cells.
struct DevBuf {
  uint32_t* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  explicit DevBuf(size_t cells) { alloc(cells); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void alloc(size_t cells) {
    release();
    if (cells) P3R_HIP(hipMalloc((void**)&p, cells * sizeof(uint32_t)));
    n = cells;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
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
  int log_max_h = 0;
  int cap_height = 0;
  size_t total_width = 0;
  // layers[l]: digests of layer l (layer 0 = leaves), SoA [8][n_l], n_l = 2^(log_max_h - l)
  std::vector<p3r::DevBuf> layers;
};

namespace p3r {
struct ProfRec {
  const char* name;
  hipEvent_t a, b;
};
}  // namespace p3r

struct p3r_ctx {
  p3r_config cfg{};
  bool prof_enabled = false;
  std::vector<p3r::ProfRec> prof;
  int partial_rounds = 0;
  hipStream_t stream = nullptr;
  p3r::DevBuf rc;  // Poseidon2 constants, Montgomery
  std::vector<uint32_t> rc_canonical;
  std::string err;

  // NTT table caches (device), keyed by log size / direction / shift.
  std::map<std::pair<int, int>, p3r::DevBuf> tw_sub;                 // (log_r, inverse)
  std::map<std::pair<int, int>, std::pair<p3r::DevBuf, p3r::DevBuf>> tw4;  // (log_n, inverse) -> (lo, hi)
  std::map<std::tuple<int, int, uint32_t>, std::pair<p3r::DevBuf, p3r::DevBuf>> pre;  // (log_n, added_bits, shift)
};
