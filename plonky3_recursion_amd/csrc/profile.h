// Per-kernel-family timing with HIP events on the ctx's own stream (bench.py's
// `roofline.achieved` is derived from these; torch.cuda.Event would not see this stream).
#pragma once
#include <chrono>

#include "context.h"

namespace p3r {

inline void prof_clear(p3r_ctx* ctx) {
  for (auto& r : ctx->prof) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  ctx->prof.clear();
  ctx->stage_ms.clear();
  ctx->cur_stage.clear();
}

// Stage marks: `prof_stage(ctx, "name")` closes the previous stage and opens a new one;
// `prof_stage(ctx, nullptr)` closes the last.  No-ops unless profiling is enabled.
inline void prof_stage(p3r_ctx* ctx, const char* name) {
  p3r::host_mark(name ? name : "stage end");
  if (!ctx->prof_enabled) return;
  (void)hipStreamSynchronize(ctx->stream);
  const double now = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  if (!ctx->cur_stage.empty()) {
    bool found = false;
    for (auto& kv : ctx->stage_ms)
      if (kv.first == ctx->cur_stage) { kv.second += now - ctx->cur_stage_t0; found = true; }
    if (!found) ctx->stage_ms.emplace_back(ctx->cur_stage, now - ctx->cur_stage_t0);
  }
  ctx->cur_stage = name ? name : "";
  ctx->cur_stage_t0 = now;
}

// A counter next to the timers (read back as the profile entry "stage:count:<name>", its value in total_ms): what a family
// processed while profiling was on - e.g. the permutations the leaf-hash launches absorbed, which bench.py prices
// against the issue peak without modelling the table mix of every configuration.
inline void prof_count(p3r_ctx* ctx, const char* name, double v) {
  if (!ctx->prof_enabled) return;
  const std::string key = std::string("count:") + name;
  for (auto& kv : ctx->stage_ms)
    if (kv.first == key) { kv.second += v; return; }
  ctx->stage_ms.emplace_back(key, v);
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
