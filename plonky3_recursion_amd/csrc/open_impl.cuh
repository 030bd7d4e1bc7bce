// K9 sequencing: all openings of one proof share barycentric weight vectors (one per distinct
// (height, point)) and come back to the host in ONE transfer.  Included into p3r_core.hip.
//
// Opening points and their order: recursion/src/verifier/batch_stark.rs:645-852 (rounds),
// :1114-1276 (observation order).  Values are the unique interpolants, so any exact evaluation
// method matches upstream's `interpolate_coset`.
namespace {

template <class PP>
struct Opener {
  using F = Fp<PP>;
  using E = Fp4<PP>;
  p3r_ctx* ctx;
  DevBuf out;
  size_t used = 0;
  struct Job { size_t off; int P, w; };
  std::vector<Job> jobs;
  std::map<std::array<uint64_t, 3>, DevBuf> wcache;

  explicit Opener(p3r_ctx* c, size_t capacity_ef = 8192) : ctx(c), out(capacity_ef * 4) {}

  // L_i(z) = w^i (z^n - 1) / (n (z - w^i)) over the size-n subgroup
  const uint32_t* weights(size_t n, const E& z) {
    std::array<uint64_t, 3> key{n, ((uint64_t)z.c[0].v << 32) | z.c[1].v, ((uint64_t)z.c[2].v << 32) | z.c[3].v};
    auto it = wcache.find(key);
    if (it != wcache.end()) return it->second.p;
    const int log_n = log2_exact(n, "trace height");
    DevBuf b(4 * n);
    E scale = (z.pow(n) - E::one()) * F::from_u64(n).inv();
    ProfScope ps(ctx, "open_weights");
    hipLaunchKernelGGL(k_bary_weights<PP>, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream, n,
                       F::two_adic_generator(log_n).v, e4_store<PP>(z), e4_store<PP>(scale), b.p);
    return wcache.emplace(key, std::move(b)).first->second.p;
  }

  // `mat`: n x w evaluations over dshift*<w_n> (natural order).  Returns a job id.
  size_t open(const uint32_t* mat, size_t n, int w, F dshift, const std::vector<E>& points) {
    const int P = (int)points.size();
    if (used + (size_t)P * w > out.n / 4) fail(P3R_EINVAL, "too many opened values for the staging buffer");
    const F inv_shift = dshift.inv();
    const uint32_t* w0 = weights(n, points[0] * inv_shift);
    const uint32_t* w1 = P == 2 ? weights(n, points[1] * inv_shift) : nullptr;
    // rows per block: 8192 for tall matrices, fewer for short ones so the launch still has
    // ~1000 workgroups (a 2^16-row table would otherwise occupy a third of the chip)
    const size_t col_groups = (w + kOpenCols - 1) / kOpenCols;
    size_t rows_per_block = kOpenRows;
    // (at most 64 chunks: the final reduction walks them serially)
    while (rows_per_block > 2 * kBlock && (n + rows_per_block - 1) / rows_per_block < 64 &&
           col_groups * ((n + rows_per_block - 1) / rows_per_block) < 1024)
      rows_per_block /= 2;
    const int n_chunks = (int)((n + rows_per_block - 1) / rows_per_block);
    DevBuf partial((size_t)P * n_chunks * w * 4);
    ProfScope ps(ctx, "open_dot");
    dim3 grid((w + kOpenCols - 1) / kOpenCols, n_chunks);
    launch_open_dot<PP>(ctx->stream, grid, mat, n, w, w0, w1, partial.p, n_chunks, (int)rows_per_block);
    hipLaunchKernelGGL(k_open_reduce<PP>, dim3(blocks_for((size_t)P * w * 4)), dim3(kBlock), 0, ctx->stream,
                       partial.p, P, n_chunks, w, out.p + used * 4);
    P3R_HIP(hipGetLastError());
    jobs.push_back({used, P, w});
    used += (size_t)P * w;
    return jobs.size() - 1;
  }

  // values[job][point][col]
  std::vector<std::vector<std::vector<E>>> finish() {
    std::vector<uint32_t> raw(used * 4);
    P3R_HIP(copy_sync(ctx->stream, raw.data(), out.p, raw.size() * 4, hipMemcpyDeviceToHost));
    std::vector<std::vector<std::vector<E>>> res(jobs.size());
    for (size_t j = 0; j < jobs.size(); ++j) {
      res[j].resize(jobs[j].P);
      for (int p = 0; p < jobs[j].P; ++p) {
        res[j][p].resize(jobs[j].w);
        for (int c = 0; c < jobs[j].w; ++c)
          for (int k = 0; k < 4; ++k)
            res[j][p][c].c[k] = F::raw(raw[(jobs[j].off + (size_t)p * jobs[j].w + c) * 4 + k]);
      }
    }
    return res;
  }
};

}  // namespace
