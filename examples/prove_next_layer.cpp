// prove_next_layer from C++ through include/p3r.hpp: the shape of
// recursion/examples/recursive_fibonacci.rs (build the prep cache once, prove layers against it,
// verify) with the synthetic verifier circuit of harness/synth.cpp standing in for
// build_next_layer_circuit, which is out of scope.
//
//   prove_next_layer <field: koala-bear|baby-bear> <log_height> <proof_out_file> [layers]
//
// Prints one line per layer; writes the postcard bytes of the inner BatchProof of the last layer.
#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <fstream>
#include <string>

#include "p3r.hpp"

namespace {
struct Synth {  // harness/libp3r_synth.so
  void* lib;
  void* (*generate)(int, int, uint64_t, int, int, int, const uint32_t*, uint32_t);
  const char* (*error)(void*);
  int (*get)(void*, const char*, const uint32_t**, size_t*);
  void (*release)(void*);
  explicit Synth(const std::string& path) {
    lib = dlopen(path.c_str(), RTLD_NOW);
    if (!lib) throw std::runtime_error(dlerror());
    generate = reinterpret_cast<decltype(generate)>(dlsym(lib, "syn_generate"));
    error = reinterpret_cast<decltype(error)>(dlsym(lib, "syn_error"));
    get = reinterpret_cast<decltype(get)>(dlsym(lib, "syn_get"));
    release = reinterpret_cast<decltype(release)>(dlsym(lib, "syn_free"));
  }
};
std::vector<uint32_t> arr(const Synth& s, void* h, const char* name) {
  const uint32_t* p = nullptr;
  size_t n = 0;
  if (s.get(h, name, &p, &n) != 0) return {};
  return std::vector<uint32_t>(p, p + n);
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: %s <field> <log_height> <proof_out> [layers] [--quintic | --arity4 | --zk]\n", argv[0]); return 2; }
  try {
    const p3r::Field field = std::string(argv[1]) == "baby-bear" ? p3r::Field::BabyBear : p3r::Field::KoalaBear;
    const int log_h = std::atoi(argv[2]);
    const int layers = argc > 4 ? std::atoi(argv[4]) : 2;
    // --quintic (recursive_fibonacci.rs:515): KoalaBear, a D = 5 verifier circuit (base-mode Poseidon2 permutations,
    // both Recompose kinds) under koala_bear_quintic_params, i.e. Challenge = the quintic field as well
    const bool quintic = argc > 5 && std::string(argv[5]) == "--quintic";
    // --arity4 (recursive_aggregation.rs:902-1046 `--arity4`): the PCS commits with MyMmcsArity4 - 4-to-1 trees over the
    // width-32 permutation, the challenger stays on width 16
    const bool arity4 = argc > 5 && std::string(argv[5]) == "--arity4";
    // --zk (create_config_zk, recursion/examples/common/mod.rs:511-553): the PCS is HidingFriPcs with two random codewords.
    // Every proof draws fresh randomness (the context counts its proofs); p3r_zk_set_nonce replays one.
    const bool zk = argc > 5 && std::string(argv[5]) == "--zk";
    constexpr uint64_t kReplayNonce = 7;
    const uint32_t D = quintic ? 5 : 4;
    std::string self = argv[0];
    const std::string root = self.substr(0, self.rfind('/')) + "/..";

    p3r::FriParams fri;  // the examples' defaults: blow-up 4, 54 queries, 15 bits of query PoW
    if (arity4) { fri.mmcs_arity = 4; fri.allow_unpinned_w32_defaults = true; }   // this example has no upstream statics to pass: the built-in width-32 constants, acknowledged as unpinned
    if (zk) { fri.zk = true; fri.num_random_codewords = 2; fri.zk_key = {3, 0, 0, 0, 0, 0, 0, 0}; fri.zk_deterministic = true; }   // (the test replays this proof on the CPU oracle)
    p3r::Context ctx(field, fri, 0, {}, D, 0, quintic ? 5 : 4);
    std::vector<uint32_t> rc(p3r_poseidon2_num_constants(ctx.raw()));
    ctx.check(p3r_poseidon2_round_constants(ctx.raw(), rc.data()));

    // the verifier circuit + its inputs
    Synth synth(root + "/harness/libp3r_synth.so");
    void* w = synth.generate((int)field, log_h, 0x5EED0000, 64, 8, 20, rc.data(), quintic ? (64u /* both Recompose kinds */ | 5u << 8) : 0u);
    if (*synth.error(w)) throw std::runtime_error(synth.error(w));
    p3r::Circuit circuit;
    circuit.witness_count = arr(synth, w, "counts")[5];
    const auto ops = arr(synth, w, "ops");
    circuit.ops.resize(ops.size() / 8);
    std::memcpy(circuit.ops.data(), ops.data(), ops.size() * 4);
    circuit.ext = arr(synth, w, "ext");
    circuit.public_rows = arr(synth, w, "public_rows");
    circuit.private_input_rows = arr(synth, w, "private_rows");
    circuit.witness_rewrite = arr(synth, w, "rewrite");
    p3r::CircuitInputs inputs{arr(synth, w, "in_public_values"), arr(synth, w, "in_private_values"), arr(synth, w, "pd_op_ids"),
                              arr(synth, w, "pd_siblings")};
    synth.release(w);

    const p3r::FriRecursionBackend backend;
    p3r::ProveNextLayerParams params;
    params.table_packing = p3r::TablePacking::create(1, 3).with_fri_params(fri.log_final_poly_len, fri.log_blowup);
    const size_t n_ops = circuit.ops.size();
    const p3r::Circuit node_circuit = circuit;  // the same circuit again as a 2-to-1 aggregation node, below
    auto t0 = std::chrono::steady_clock::now();
    p3r::NextLayerPrepCache prep = p3r::build_next_layer_prep(ctx, std::move(circuit), backend, params);
    ctx.sync();
    std::printf("build_next_layer_prep: %zu ops, %zu schedule levels, %.1f ms\n", n_ops, prep.prepared_circuit->levels(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());

    p3r::RecursionInput input;
    input.circuit_inputs = &inputs;
    p3r::RecursionOutput out;
    for (int l = 0; l < layers; ++l) {
      if (zk && l == layers - 1) ctx.check(p3r_zk_set_nonce(ctx.raw(), kReplayNonce));   // the written proof is proof number 7
      t0 = std::chrono::steady_clock::now();
      out = p3r::prove_next_layer(input, ctx, backend, params, prep);
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      prep.prover->verify_all_tables(out.proof);
      std::printf("layer %d: prove_next_layer %.1f ms, proof %zu bytes (%zu with metadata), verify_all_tables ok\n", l, ms,
                  out.proof.proof.size(), out.proof.to_postcard().size());
    }
    {
      // the wire form and back (what moves between the processes of an aggregation tree): native parser, same bytes again,
      // and the parsed proof verifies from its own metadata
      const std::vector<uint8_t> wire = out.proof.to_postcard();
      const p3r::BatchStarkProof back = p3r::BatchStarkProof::from_postcard(wire, field, true, quintic ? 5 : 4, zk);
      if (back.proof != out.proof.proof || back.to_postcard() != wire) throw p3r::Error(P3R_EINVAL, "postcard round trip differs");
      p3r::verify_all_tables(ctx.config(), back);
      std::printf("BatchStarkProof postcard round trip ok (%zu bytes)\n", wire.size());
    }
    // prove_aggregation_layer (recursion.rs:656-762): the circuit's inputs arrive as the shares of the two
    // proofs it verifies; the AggregationPrepCache slot is filled by the first call and reused by the second
    {
      const size_t hp = inputs.public_values.size() / (2 * D) * D, hv = inputs.private_values.size() / (2 * D) * D;
      const size_t hs = inputs.private_data_op_ids.size() / 2;
      const uint32_t n_left = hs < inputs.private_data_op_ids.size() ? inputs.private_data_op_ids[hs] : 0;
      size_t cut = 0;
      while (cut < inputs.private_data_op_ids.size() && inputs.private_data_op_ids[cut] < n_left) ++cut;
      p3r::CircuitInputs l, r;
      l.public_values.assign(inputs.public_values.begin(), inputs.public_values.begin() + hp);
      r.public_values.assign(inputs.public_values.begin() + hp, inputs.public_values.end());
      l.private_values.assign(inputs.private_values.begin(), inputs.private_values.begin() + hv);
      r.private_values.assign(inputs.private_values.begin() + hv, inputs.private_values.end());
      l.private_data_op_ids.assign(inputs.private_data_op_ids.begin(), inputs.private_data_op_ids.begin() + cut);
      for (size_t i = cut; i < inputs.private_data_op_ids.size(); ++i) r.private_data_op_ids.push_back(inputs.private_data_op_ids[i] - n_left);
      l.private_data_siblings.assign(inputs.private_data_siblings.begin(), inputs.private_data_siblings.begin() + cut * 8);
      r.private_data_siblings.assign(inputs.private_data_siblings.begin() + cut * 8, inputs.private_data_siblings.end());
      p3r::RecursionInput li, ri;
      li.circuit_inputs = &l;
      ri.circuit_inputs = &r;
      std::unique_ptr<p3r::AggregationPrepCache> slot;
      if (zk) {   // two proofs of one input differ under ZK ...
        p3r::RecursionOutput fresh = p3r::prove_next_layer(input, ctx, backend, params, prep);
        if (fresh.proof.proof == out.proof.proof) throw std::runtime_error("two ZK proofs of one input are identical");
        prep.prover->verify_all_tables(fresh.proof);
        ctx.check(p3r_zk_set_nonce(ctx.raw(), kReplayNonce));   // ... and a replayed proof number gives the same bytes
      }
      p3r::RecursionOutput a1 = p3r::prove_aggregation_layer(li, ri, node_circuit, ctx, backend, params, &slot, n_left);
      const p3r::AggregationPrepCache* filled = slot.get();
      if (zk) ctx.check(p3r_zk_set_nonce(ctx.raw(), kReplayNonce));
      t0 = std::chrono::steady_clock::now();
      p3r::RecursionOutput a2 = p3r::prove_aggregation_layer(li, ri, node_circuit, ctx, backend, params, &slot, n_left);
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (!filled || slot.get() != filled) throw std::runtime_error("AggregationPrepCache was not reused");
      if (a1.proof.proof != out.proof.proof || a2.proof.proof != out.proof.proof)
        throw std::runtime_error("prove_aggregation_layer bytes differ from prove_next_layer of the same circuit");
      slot->prover->verify_all_tables(a2.proof);
      std::printf("prove_aggregation_layer (cached prep) %.1f ms, same bytes, verify_all_tables ok\n", ms);
    }
    // negative checks: tampered bytes and tampered metadata are refused
    p3r::BatchStarkProof bad = out.proof;
    bad.proof[bad.proof.size() / 2] ^= 1;
    bool rejected = false;
    try { prep.prover->verify_all_tables(bad); } catch (const p3r::Error&) { rejected = true; }
    if (!rejected) throw std::runtime_error("a tampered proof was accepted");
    bad = out.proof;
    bad.table_packing.horner_packed_steps = 1;
    rejected = false;
    try { prep.prover->verify_all_tables(bad); } catch (const p3r::Error& e) { rejected = std::string(e.what()).find("BadHornerPackedSteps") != std::string::npos; }
    if (!rejected) throw std::runtime_error("tampered metadata was accepted");
    std::ofstream(argv[3], std::ios::binary).write(reinterpret_cast<const char*>(out.proof.proof.data()), (std::streamsize)out.proof.proof.size());
    std::printf("ok\n");
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
}
