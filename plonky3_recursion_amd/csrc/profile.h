// Per-kernel-family timing with HIP events on the ctx's own stream (bench.py's
// `roofline.achieved` is derived from these; torch.cuda.Event would not see this stream).
#pragma once
#include "context.h"

namespace p3r {

inline void prof_clear(p3r_ctx* ctx) {
  for (auto& r : ctx->prof) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  ctx->prof.clear();
}

struct ProfScope {
  p3r_ctx* ctx;
  hipEvent_t a = nullptr, b = nullptr;
  const char* name;
  ProfScope(p3r_ctx* c, const char* n) : ctx(c), name(n) {
    if (!ctx->prof_enabled) return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
      a = b = nullptr;
      return;
    }
    (void)hipEventRecord(a, ctx->stream);
  }
  ~ProfScope() {
    if (!a) return;
    (void)hipEventRecord(b, ctx->stream);
    ctx->prof.push_back(ProfRec{name, a, b});
  }
};

}  // namespace p3r
