// The host-only entry points of the C ABI (include/p3r.h): `verify_batch` behind verify_all_tables, the proof parsers
// and `Mmcs::verify_batch` - no device, no p3r_ctx.  A parent node of an aggregation tree runs them on bytes that
// arrived from another rank (plonky3_recursion_amd/aggregation.py), so they are also compiled, without the device
// code, under AddressSanitizer + UBSan and driven by a structure-aware mutator (tests/san/).  Included once per
// library: by p3r_core.hip in the product, by tests/san/san_host.hip in the sanitizer build.
#pragma once
#include <algorithm>
#include <chrono>

#include "poseidon2_rc_default.inc"
#include "poseidon2_w32_default.inc"
#include "verify_impl.h"

using namespace p3r;

namespace {

// The permutation constants of a configuration, canonical: the width-16 round constants, then the width-32 table
// (round constants | diagonal, poseidon2.h) - the layout of p3r_ctx::rc and of the verifier's table.  NULL pointers
// select the self-generated defaults.  `bad`: reports a length mismatch.
template <class PP, class Bad>
std::vector<uint32_t> constants_table(const p3r_config& cfg, Bad&& bad) {
  const size_t nrc = p2_num_constants<PP>(), nrcw = p2w_num_rc<PP>();
  const uint32_t* src = cfg.poseidon2_rc;
  if (src && cfg.poseidon2_rc_len != nrc) bad("poseidon2_rc_len", cfg.poseidon2_rc_len, nrc);
  if (!src) src = PP::FIELD_ID == 0 ? kDefaultRc_koala_bear : kDefaultRc_baby_bear;
  const uint32_t* w = cfg.poseidon2_w32_rc;
  if (w && cfg.poseidon2_w32_rc_len != nrcw) bad("poseidon2_w32_rc_len", cfg.poseidon2_w32_rc_len, nrcw);
  if (!w) w = PP::FIELD_ID == 0 ? kDefaultRcW32_koala_bear : kDefaultRcW32_baby_bear;
  const uint32_t* dg = cfg.poseidon2_w32_diag ? cfg.poseidon2_w32_diag : (PP::FIELD_ID == 0 ? kDefaultDiagW32_koala_bear : kDefaultDiagW32_baby_bear);
  std::vector<uint32_t> t(src, src + nrc);
  t.insert(t.end(), w, w + nrcw);
  t.insert(t.end(), dg, dg + P2W_WIDTH);
  for (size_t i = 0; i < t.size(); ++i)
    if (t[i] >= PP::P) p3r::vfail("%s constant %zu is not canonical (%u >= p)", i < nrc ? "poseidon2_rc" : "width-32 permutation", i < nrc ? i : i - nrc, t[i]);
  return t;
}

// p3r.h: P3R_EXT_UNPINNED_W32_DEFAULTS
inline bool w32_unacknowledged(const p3r_config& cfg) {
  return (!cfg.poseidon2_w32_rc || !cfg.poseidon2_w32_diag) && !(cfg.ext_choices & P3R_EXT_UNPINNED_W32_DEFAULTS);
}
constexpr const char* kW32Unpinned =
    "the width-32 permutation with poseidon2_w32_rc / poseidon2_w32_diag NULL: the built-in constants are self-generated, not "
    "upstream's - pass the caller's, or acknowledge with P3R_EXT_UNPINNED_W32_DEFAULTS";

}  // namespace

extern "C" {

int p3r_mmcs_verify(const p3r_config* cfg, const uint32_t* cap, size_t n_mats, const size_t* heights, const size_t* widths,
                    size_t index, const uint32_t* opened_values, const uint32_t* proof, size_t proof_len, char* err_buf,
                    size_t err_cap) {
  auto report = [&](const char* msg) {
    if (err_buf && err_cap) snprintf(err_buf, err_cap, "%s", msg);
  };
  try {
    if (!cfg || !cap || !heights || !widths || !opened_values || (!proof && proof_len) || !n_mats) { report("NULL argument"); return P3R_EINVAL; }
    if (cfg->abi_version != P3R_ABI_VERSION) { report("ABI version mismatch"); return P3R_EINVAL; }
    if (cfg->mmcs_arity != 0 && cfg->mmcs_arity != 2 && cfg->mmcs_arity != 4) { report("mmcs_arity must be 2 or 4"); return P3R_EINVAL; }
    if (cfg->mmcs_arity == 4 && cfg->cap_height != 0) { report("arity-4 MMCS: cap_height must be 0"); return P3R_EUNSUPPORTED; }
    if (cfg->mmcs_arity == 4 && w32_unacknowledged(*cfg)) { report(kW32Unpinned); return P3R_EINVAL; }
    // the cap is read before the tree is walked: bound it by the tallest matrix first (a shift by >= 64 is undefined, a
    // large one an allocation of that many digests and a read past `cap`)
    {
      size_t hmax = 0;
      for (size_t m = 0; m < n_mats; ++m) {
        if (!heights[m] || (heights[m] & (heights[m] - 1))) { report("matrix heights must be powers of two"); return P3R_EINVAL; }
        if (widths[m] > (size_t(1) << 24)) { report("matrix width out of range"); return P3R_EINVAL; }
        hmax = std::max(hmax, heights[m]);
      }
      if (cfg->cap_height >= 48 || (size_t(1) << cfg->cap_height) > hmax) { report("cap_height exceeds the height of the tallest matrix"); return P3R_EINVAL; }
      if (proof_len > 64 * 3) { report("opening proof longer than any tree's"); return P3R_EINVAL; }
    }
    auto run = [&](auto tag) {
      using PP = decltype(tag);
      using F = p3r::Fp<PP>;
      using Digest = std::array<F, P2_DIGEST>;
      const std::vector<uint32_t> table = constants_table<PP>(*cfg, [](const char* what, uint32_t got, size_t want) {
        p3r::vfail("%s is %u, the field needs %zu constants", what, got, want);
      });
      std::vector<uint32_t> rc(table.size());
      for (size_t i = 0; i < rc.size(); ++i) rc[i] = F::from_canonical(table[i]).v;
      auto fe = [](uint32_t v) {
        if (v >= PP::P) p3r::vfail("non-canonical field element");
        return F::from_canonical(v);
      };
      std::vector<Digest> capd(size_t(1) << cfg->cap_height), path(proof_len);
      for (auto& d : capd) for (auto& x : d) x = fe(*cap++);
      const uint32_t* pf = proof;
      for (auto& d : path) for (auto& x : d) x = fe(*pf++);
      std::vector<int> lhs;
      std::vector<std::vector<F>> rows;
      const uint32_t* ov = opened_values;
      for (size_t m = 0; m < n_mats; ++m) {
        if (!heights[m] || (heights[m] & (heights[m] - 1))) p3r::vfail("matrix heights must be powers of two");
        lhs.push_back(p3r::log2_exact(heights[m], "matrix height"));
        std::vector<F> r(widths[m]);
        for (auto& x : r) x = fe(*ov++);
        rows.push_back(std::move(r));
      }
      if (cfg->mmcs_arity == 4) p3r::mmcs_verify4<PP>(capd, (int)cfg->cap_height, lhs, rows, index, path, rc.data() + p3r::p2_num_constants<PP>(), "opening");
      else p3r::mmcs_verify<PP>(capd, (int)cfg->cap_height, lhs, rows, index, path, rc.data(), "opening");
    };
    if (cfg->field == P3R_FIELD_KOALA_BEAR) run(p3r::KoalaBearParams{});
    else if (cfg->field == P3R_FIELD_BABY_BEAR) run(p3r::BabyBearParams{});
    else { report("unknown field"); return P3R_EINVAL; }
    return P3R_OK;
  } catch (const std::exception& e) {
    report(e.what());
    return P3R_EINVAL;
  }
}

// ---- native verifier (verify_impl.h): host code, no device needed ----
// MerkleTreeHidingMmcs::verify_batch: the leaf preimage of a height class is the concatenation of [row | salt] per matrix
// (recursion/src/pcs/mmcs.rs:375-389), i.e. the plain walk over the matrices widened by cfg->mmcs_salt_elems columns.
int p3r_mmcs_verify_salted(const p3r_config* cfg, const uint32_t* cap, size_t n_mats, const size_t* heights, const size_t* widths,
                           size_t index, const uint32_t* opened_values, const uint32_t* salts, const uint32_t* proof,
                           size_t proof_len, char* err_buf, size_t err_cap) {
  if (!cfg || !widths || !opened_values || !n_mats || (cfg->mmcs_salt_elems && !salts)) {
    if (err_buf && err_cap) snprintf(err_buf, err_cap, "NULL argument");
    return P3R_EINVAL;
  }
  const size_t S = cfg->mmcs_salt_elems;
  if (S > 16) {
    if (err_buf && err_cap) snprintf(err_buf, err_cap, "mmcs_salt_elems must be in 0..16");
    return P3R_EINVAL;
  }
  std::vector<size_t> wide(widths, widths + n_mats);
  std::vector<uint32_t> rows;
  const uint32_t* ov = opened_values;
  for (size_t m = 0; m < n_mats; ++m) {
    if (widths[m] > (size_t(1) << 24)) {
      if (err_buf && err_cap) snprintf(err_buf, err_cap, "matrix width out of range");
      return P3R_EINVAL;
    }
    rows.insert(rows.end(), ov, ov + widths[m]);
    ov += widths[m];
    if (S) rows.insert(rows.end(), salts + m * S, salts + (m + 1) * S);
    wide[m] += S;
  }
  return p3r_mmcs_verify(cfg, cap, n_mats, heights, wide.data(), index, rows.data(), proof, proof_len, err_buf, err_cap);
}

int p3r_verify_batch(const p3r_config* cfg, const p3r_air_desc* airs, size_t n_airs,
                     const uint32_t* preprocessed_commitment, const uint32_t* degree_bits, const uint8_t* proof,
                     size_t proof_len, uint32_t flags, char* err_buf, size_t err_cap) {
  auto report = [&](const char* msg) {
    if (err_buf && err_cap) snprintf(err_buf, err_cap, "%s", msg);
  };
  try {
    if (!cfg || !airs || !preprocessed_commitment || !degree_bits || (!proof && proof_len)) { report("NULL argument"); return P3R_EINVAL; }
    if (cfg->abi_version != P3R_ABI_VERSION) { report("ABI version mismatch"); return P3R_EINVAL; }
    if (cfg->challenge_degree != 0 && cfg->challenge_degree != 4 && cfg->challenge_degree != 5) { report("UnsupportedChallengeDegree"); return P3R_EUNSUPPORTED; }
    const bool generic_d = p3r::ext_degree_is_binomial_generic(cfg->ext_degree);
    if (generic_d && cfg->ext_w == 0) { report("MissingWForExtension"); return P3R_EINVAL; }
    if (!generic_d && cfg->ext_degree != 1 && cfg->ext_degree != 4 && !(cfg->ext_degree == 5 && cfg->field == P3R_FIELD_KOALA_BEAR)) { report("UnsupportedDegree"); return P3R_EUNSUPPORTED; }
    p3r::VerifyParams prm{(int)cfg->log_blowup, (int)cfg->max_log_arity, (int)cfg->cap_height, (int)cfg->log_final_poly_len,
                          (int)cfg->commit_pow_bits, (int)cfg->query_pow_bits, (int)cfg->num_queries, {}};
    if (cfg->fri_log_arities) prm.fri_log_arities.assign(cfg->fri_log_arities, cfg->fri_log_arities + cfg->fri_log_arities_len);
    if (cfg->mmcs_arity != 0 && cfg->mmcs_arity != 2 && cfg->mmcs_arity != 4) { report("mmcs_arity must be 2 or 4"); return P3R_EINVAL; }
    if (cfg->mmcs_arity == 4 && cfg->cap_height != 0) { report("arity-4 MMCS: cap_height must be 0"); return P3R_EUNSUPPORTED; }
    prm.mmcs_arity = cfg->mmcs_arity == 4 ? 4 : 2;
    if (w32_unacknowledged(*cfg)) {
      bool uses_w32 = cfg->mmcs_arity == 4;
      for (size_t i = 0; i < n_airs; ++i) uses_w32 = uses_w32 || airs[i].kind == P3R_AIR_POSEIDON2_W32;
      if (uses_w32) { report(kW32Unpinned); return P3R_EINVAL; }
    }
    if (cfg->zk > 1 || cfg->num_random_codewords > 8) { report("zk must be 0 or 1, num_random_codewords at most 8"); return P3R_EINVAL; }
    if (cfg->mmcs_salt_elems > 16) { report("mmcs_salt_elems must be in 0..16"); return P3R_EINVAL; }
    prm.mmcs_salt_elems = (int)cfg->mmcs_salt_elems;
    prm.zk = (int)cfg->zk;
    prm.num_random_codewords = cfg->zk ? (cfg->num_random_codewords ? (int)cfg->num_random_codewords : 2) : 0;
    if (!prm.layout.set(cfg->proof_layout, cfg->proof_layout_len)) { report("proof_layout must be 18 bytes: three permutations"); return P3R_EINVAL; }
    std::vector<p3r::AirParams> a(n_airs);
    for (size_t i = 0; i < n_airs; ++i) {
      if (airs[i].kind > P3R_AIR_POSEIDON2_W32 || !airs[i].lanes) { report("bad AIR descriptor"); return P3R_EINVAL; }
      a[i] = {(int)airs[i].kind, (int)airs[i].lanes, (int)airs[i].horner_packed_steps, (int)airs[i].coeff_lookups,
              (cfg->ext_choices & P3R_EXT_LOOKUP_UNPACKED) ? 1 : 0, (int)cfg->ext_degree, 0u};
    }
    const bool canonical = (flags & P3R_PROVE_CANONICAL_FIELD_ENCODING) != 0;
    std::vector<uint32_t> cap(preprocessed_commitment, preprocessed_commitment + ((size_t)P2_DIGEST << cfg->cap_height));
    const std::vector<uint32_t> want_db(degree_bits, degree_bits + n_airs);
    auto run = [&](auto tag) {
      using PP = decltype(tag);
      const std::vector<uint32_t> table = constants_table<PP>(*cfg, [](const char* what, uint32_t got, size_t want) {
        p3r::vfail("%s is %u, the field needs %zu constants", what, got, want);
      });
      auto airs_pp = a;
      for (auto& x : airs_pp) x.ext_w_mont = generic_d ? p3r::Fp<PP>::from_canonical(cfg->ext_w).v : 0u;
      if (cfg->challenge_degree == 5) {
        if constexpr (p3r::kHasQuintic<PP>)
          p3r::verify_batch<PP, 5>(prm, table, airs_pp, cap, want_db, proof, proof_len, canonical);
        else
          p3r::vfail("UnsupportedChallengeDegree: the quintic challenge field is KoalaBear's");
      } else {
        p3r::verify_batch<PP>(prm, table, airs_pp, cap, want_db, proof, proof_len, canonical);
      }
    };
    if (cfg->field == P3R_FIELD_KOALA_BEAR) run(p3r::KoalaBearParams{});
    else if (cfg->field == P3R_FIELD_BABY_BEAR) run(p3r::BabyBearParams{});
    else { report("unknown field"); return P3R_EINVAL; }
    return P3R_OK;
  } catch (const p3r::VerifyFailure& e) {
    report(e.what());
    return P3R_EINVAL;
  } catch (const std::exception& e) {
    report(e.what());
    return P3R_EINVAL;
  }
}

int p3r_batch_proof_len(uint32_t field, const uint8_t* bytes, size_t len, uint32_t flags, size_t* proof_len,
                        char* err_buf, size_t err_cap) {
  return p3r_batch_proof_len_layout(field, bytes, len, flags, nullptr, proof_len, err_buf, err_cap);
}
int p3r_batch_proof_len_layout(uint32_t field, const uint8_t* bytes, size_t len, uint32_t flags,
                               const uint8_t* proof_layout, size_t* proof_len, char* err_buf, size_t err_cap) {
  try {
    if (!bytes || !proof_len) throw std::runtime_error("NULL argument");
    const bool canonical = (flags & P3R_PROVE_CANONICAL_FIELD_ENCODING) != 0;
    p3r::ProofLayout PL;
    if (!PL.set(proof_layout, 18)) throw std::runtime_error("proof_layout must be three permutations batch[5] | fri[5] | opened[8]");
    const bool zk = (flags & P3R_PROOF_ZK) != 0, salted = (flags & P3R_PROOF_SALTED) != 0;
    if (field == P3R_FIELD_KOALA_BEAR && (flags & P3R_PROOF_QUINTIC_CHALLENGE))
      (void)p3r::parse_proof<p3r::KoalaBearParams, 5>(bytes, len, canonical, proof_len, PL, zk, salted);
    else if (field == P3R_FIELD_KOALA_BEAR) (void)p3r::parse_proof<p3r::KoalaBearParams>(bytes, len, canonical, proof_len, PL, zk, salted);
    else if (field == P3R_FIELD_BABY_BEAR) (void)p3r::parse_proof<p3r::BabyBearParams>(bytes, len, canonical, proof_len, PL, zk, salted);
    else throw std::runtime_error("unknown field");
    return P3R_OK;
  } catch (const std::exception& e) {
    if (err_buf && err_cap) snprintf(err_buf, err_cap, "%s", e.what());
    return P3R_EINVAL;
  }
}

int p3r_batch_stark_proof_parse(uint32_t field, const uint8_t* bytes, size_t len, uint32_t flags,
                                const uint8_t* proof_layout, p3r_batch_stark_meta* out, char* err_buf,
                                size_t err_cap) {
  try {
    if (!bytes || !out) throw std::runtime_error("NULL argument");
    const auto t0 = std::chrono::steady_clock::now();
    const bool canonical = (flags & P3R_PROVE_CANONICAL_FIELD_ENCODING) != 0;
    p3r::ProofLayout PL;
    if (!PL.set(proof_layout, 18)) throw std::runtime_error("proof_layout must be three permutations batch[5] | fri[5] | opened[8]");
    const int dc = (flags & P3R_PROOF_QUINTIC_CHALLENGE) ? 5 : 4;
    const bool zk = (flags & P3R_PROOF_ZK) != 0, salted = (flags & P3R_PROOF_SALTED) != 0;
    if (field == P3R_FIELD_KOALA_BEAR) p3r::parse_batch_stark_meta<p3r::KoalaBearParams>(bytes, len, canonical, PL, dc, out, zk, salted);
    else if (field == P3R_FIELD_BABY_BEAR) p3r::parse_batch_stark_meta<p3r::BabyBearParams>(bytes, len, canonical, PL, dc, out, zk, salted);
    else throw std::runtime_error("unknown field");
    out->parse_ns = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    return P3R_OK;
  } catch (const std::exception& e) {
    if (err_buf && err_cap) snprintf(err_buf, err_cap, "%s", e.what());
    return P3R_EINVAL;
  }
}

}  // extern "C"
