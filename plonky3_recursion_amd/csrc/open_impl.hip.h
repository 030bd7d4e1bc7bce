// K9 sequencing: all openings of one proof share barycentric weight vectors (one per distinct
// (height, point)) and come back to the host in ONE transfer.  Included into p3r_core.hip.
//
// Opening points and their order: recursion/src/verifier/batch_stark.rs:645-852 (rounds),
// :1114-1276 (observation order).  Values are the unique interpolants, so any exact evaluation
// method matches upstream's `interpolate_coset`.
namespace {

template <class PP, int DC = 4>
struct Opener {
  using F = Fp<PP>;
  using E = typename Chal<PP, DC>::type;
  p3r_ctx* ctx;
  size_t used = 0;  // opened values so far (extension-field elements)
  struct Job { size_t off; int P, w; };
  std::vector<Job> jobs;
  std::vector<OpenJob> dot_jobs;
  std::vector<BaryJobT<DC>> bary_jobs;
  std::vector<DevBuf> keep;  // weights and partial sums, alive until finish()
  DevBuf out;                // the opened values on the device: [job][point][col][DC]
  std::map<std::array<uint64_t, 6>, const uint32_t*> wcache;
  uint32_t bary_blocks = 0, dot_blocks = 0;

  explicit Opener(p3r_ctx* c) : ctx(c) {}

  // L_i(z) = w^i (z^n - 1) / (n (z - w^i)) over the size-n subgroup
  const uint32_t* weights(size_t n, const E& z) {
    std::array<uint64_t, 6> key{n, 0, 0, 0, 0, 0};
    for (int k = 0; k < DC; ++k) key[1 + k] = z.c[k].v;
    auto it = wcache.find(key);
    if (it != wcache.end()) return it->second;
    const int log_n = log2_exact(n, "trace height");
    keep.emplace_back((size_t)DC * n);
    BaryJobT<DC> b{};
    b.out = keep.back().p;
    b.n = n;
    b.w_n = F::two_adic_generator(log_n).v;
    b.z = e4_store<PP, DC>(z);
    b.scale = e4_store<PP, DC>((z.pow(n) - E::one()) * F::from_u64(n).inv());
    b.block0 = bary_blocks;
    bary_blocks += blocks_for((n + 3) / 4);  // a lane owns four consecutive points
    bary_jobs.push_back(b);
    return wcache.emplace(key, b.out).first->second;
  }

  // `mat`: n x w evaluations over dshift*<w_n> (natural order).  Returns a job id; nothing is
  // launched before finish().
  size_t open(const uint32_t* mat, size_t n, int w, F dshift, const std::vector<E>& points) {
    const int P = (int)points.size();
    const F inv_shift = dshift.inv();
    OpenJob j{};
    j.mat = mat;
    j.wt0 = weights(n, points[0] * inv_shift);
    j.wt1 = P == 2 ? weights(n, points[1] * inv_shift) : nullptr;
    j.n = n;
    j.w = w;
    // rows per block: 8192 for tall matrices, fewer for short ones so the job still has
    // ~1000 workgroups (a 2^16-row table would otherwise occupy a third of the chip)
    j.col_groups = (w + kOpenCols - 1) / kOpenCols;
    size_t rows_per_block = kOpenRows;
    // (at most 64 chunks: the final reduction walks them serially)
    while (rows_per_block > 2 * kBlock && (n + rows_per_block - 1) / rows_per_block < 64 &&
           j.col_groups * ((n + rows_per_block - 1) / rows_per_block) < 1024)
      rows_per_block /= 2;
    j.rows_per_block = (int)rows_per_block;
    j.n_chunks = (int)((n + rows_per_block - 1) / rows_per_block);
    keep.emplace_back((size_t)P * j.n_chunks * w * DC);
    j.partial = keep.back().p;
    j.block0 = dot_blocks;
    dot_blocks += (uint32_t)(j.col_groups * j.n_chunks);
    j.out0 = (uint32_t)(used * DC);
    dot_jobs.push_back(j);
    jobs.push_back({used, P, w});
    used += (size_t)P * w;
    return jobs.size() - 1;
  }

  template <class T>
  const T* upload_jobs(const std::vector<T>& v) {
    keep.emplace_back((v.size() * sizeof(T) + 3) / 4);
    P3R_HIP(ctx->stage.upload(ctx->stream, keep.back().p, v.data(), v.size() * sizeof(T)));
    return reinterpret_cast<const T*>(keep.back().p);
  }

  // device address of the opened values of one job and point ([w][DC]), valid after finish()
  const uint32_t* values_dev(size_t job, int point) const {
    return out.p + (jobs[job].off + (size_t)point * jobs[job].w) * DC;
  }

  // values[job][point][col]
  std::vector<std::vector<std::vector<E>>> finish() {
    if (jobs.empty()) return {};
    out.alloc(used * DC);
    {
      const BaryJobT<DC>* d_bary = upload_jobs(bary_jobs);
      const OpenJob* d_jobs = upload_jobs(dot_jobs);
      {
        ProfScope ps(ctx, "open_weights");
        hipLaunchKernelGGL((k_bary_weights<PP, DC>), dim3(bary_blocks), dim3(kBlock), 0, ctx->stream, d_bary,
                           (int)bary_jobs.size());
      }
      ProfScope ps(ctx, "open_dot");
      hipLaunchKernelGGL((k_open_dot<PP, DC>), dim3(dot_blocks), dim3(kBlock), 0, ctx->stream, d_jobs, (int)dot_jobs.size());
      hipLaunchKernelGGL((k_open_reduce<PP, DC>), dim3(blocks_for(used * DC)), dim3(kBlock), 0, ctx->stream, d_jobs,
                         (int)dot_jobs.size(), (uint32_t)(used * DC), out.p);
      P3R_HIP(hipGetLastError());
    }
    // the opened values (4 800 words for a recursion layer) travel like a commitment root: posted by a one-workgroup
    // kernel and polled for - the copy engine's round trip is 30 - 40 us longer (profiles/r06/host_gaps.txt); read in
    // place, before the next post
    const uint32_t* raw = nullptr;
    if (HostPost::enabled() && used * DC <= HostPost::kWords) P3R_HIP(ctx->post.post(ctx->stream, out.p, used * DC, &raw));
    else P3R_HIP(ctx->landing.fetch(ctx->stream, out.p, used * DC * 4, &raw));
    keep.clear();  // `out` stays for the reduced openings (values_dev)
    std::vector<std::vector<std::vector<E>>> res(jobs.size());
    for (size_t j = 0; j < jobs.size(); ++j) {
      res[j].resize(jobs[j].P);
      for (int p = 0; p < jobs[j].P; ++p) {
        res[j][p].resize(jobs[j].w);
        for (int c = 0; c < jobs[j].w; ++c)
          for (int k = 0; k < DC; ++k)
            res[j][p][c].c[k] = F::raw(raw[(jobs[j].off + (size_t)p * jobs[j].w + c) * DC + k]);
      }
    }
    return res;
  }
};

}  // namespace
