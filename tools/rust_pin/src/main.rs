//! Parity-pinning kit: runs UPSTREAM Plonky3 (p3-* 0.6) and the reference's circuit-prover on the
//! inputs this repo's golden files hold, and writes the results in the schema tests/ already loads.
//!
//!   tests/golden/primitives.json  (inputs)  ->  tests/golden/rust_primitives.json
//!       per field: upstream round constants (flat layout of include/p3r.h), Poseidon2 permutation
//!       KATs, PaddingFreeSponge / TruncatedPermutation KATs, DuplexChallenger transcript,
//!       extension-field product and inverse, two-adic generators, coset_lde_batch of an 8 x 3 matrix.
//!   tests/golden/rust_fibonacci_layer_<field>.json
//!       the Fibonacci(n = 100) circuit of recursion/examples/recursive_fibonacci.rs:315-337 over
//!       the degree-4 extension, proved with BatchStarkProver::prove_all_tables under the FRI
//!       parameters below: postcard bytes of the whole BatchStarkProof and of its inner BatchProof,
//!       the preprocessed commitment, degree bits.  tests/test_rust_pins.py verifies these bytes with
//!       the native verifier and compares them with the oracle's and the HIP prover's bytes.
//!
//! Written against the API the reference itself uses (recursion/examples/common/mod.rs:192-207,
//! 464-486; circuit-prover/src/batch_stark_prover/tests.rs:1031-1099).  It has NOT been compiled in
//! this repo's build image (no cargo there): expect to fix an import or two on first use.
use std::fs;

use p3_challenger::{CanObserve, CanSample, CanSampleBits, DuplexChallenger, FieldChallenger};
use p3_circuit::CircuitBuilder;
use p3_circuit_prover::batch_stark_prover::BatchStarkProof;
use p3_circuit_prover::common::get_airs_and_degrees_with_prep;
use p3_circuit_prover::{BatchStarkProver, CircuitProverData, ConstraintProfile, TablePacking};
use p3_batch_stark::ProverData;
use p3_commit::{ExtensionMmcs, Mmcs};
use p3_dft::{Radix2DitParallel, TwoAdicSubgroupDft};
use p3_field::extension::BinomialExtensionField;
use p3_field::{BasedVectorSpace, Field, PrimeCharacteristicRing, PrimeField32, TwoAdicField};
use p3_fri::{FriParameters, TwoAdicFriPcs};
use p3_matrix::Matrix;
use p3_matrix::bitrev::BitReversibleMatrix;
use p3_matrix::dense::RowMajorMatrix;
use p3_merkle_tree::MerkleTreeMmcs;
use p3_symmetric::{CryptographicHasher, PaddingFreeSponge, Permutation, PseudoCompressionFunction, TruncatedPermutation};
use p3_uni_stark::StarkConfig;
use serde_json::{Value, json};

const WIDTH: usize = 16;
const RATE: usize = 8;
const DIGEST: usize = 8;

// FRI parameters of the layer fixture (small, so that the oracle proves it in milliseconds)
const LOG_BLOWUP: usize = 2;
const MAX_LOG_ARITY: usize = 2;
const CAP_HEIGHT: usize = 0;
const LOG_FINAL_POLY_LEN: usize = 2;
const COMMIT_POW_BITS: usize = 0;
const QUERY_POW_BITS: usize = 6;
const NUM_QUERIES: usize = 8;
const FIB_N: usize = 100;

fn u32s<F: PrimeField32>(xs: &[F]) -> Vec<u32> {
    xs.iter().map(|x| x.as_canonical_u32()).collect()
}
fn from_json<F: PrimeCharacteristicRing>(v: &Value) -> Vec<F> {
    v.as_array().unwrap().iter().map(|x| F::from_u64(x.as_u64().unwrap())).collect()
}

macro_rules! field_module {
    ($modname:ident, $F:ty, $Perm:ty, $default_perm:path, $rc_ei:path, $rc_int:path, $rc_ef:path, $key:literal) => {
        mod $modname {
            use super::*;
            pub type F = $F;
            pub type Challenge = BinomialExtensionField<F, 4>;
            pub type Perm = $Perm;
            pub type MyHash = PaddingFreeSponge<Perm, WIDTH, RATE, DIGEST>;
            pub type MyCompress = TruncatedPermutation<Perm, 2, DIGEST, WIDTH>;
            pub type MyMmcs = MerkleTreeMmcs<<F as Field>::Packing, <F as Field>::Packing, MyHash, MyCompress, 2, DIGEST>;
            pub type ChallengeMmcs = ExtensionMmcs<F, Challenge, MyMmcs>;
            pub type Challenger = DuplexChallenger<F, Perm, WIDTH, RATE>;
            pub type Dft = Radix2DitParallel<F>;
            pub type MyPcs = TwoAdicFriPcs<F, Dft, MyMmcs, ChallengeMmcs>;
            pub type MyConfig = StarkConfig<MyPcs, Challenge, Challenger>;

            /// the upstream statics in the flat layout of p3r_config.poseidon2_rc:
            /// [4][16] external-initial | [partial] internal | [4][16] external-final
            pub fn round_constants() -> Vec<u32> {
                let mut out = Vec::new();
                for r in $rc_ei.iter() { out.extend(u32s(r)); }
                out.extend(u32s(&$rc_int[..]));
                for r in $rc_ef.iter() { out.extend(u32s(r)); }
                out
            }

            pub fn primitives(inp: &Value) -> Value {
                let perm: Perm = $default_perm();
                // permutation KATs on the committed inputs
                let permute: Vec<Value> = inp["permute"].as_array().unwrap().iter().map(|k| {
                    let xs: Vec<F> = from_json(&k["in"]);
                    let mut s: [F; WIDTH] = xs.try_into().unwrap();
                    perm.permute_mut(&mut s);
                    json!({"in": k["in"], "out": u32s(&s)})
                }).collect();
                // PaddingFreeSponge (ragged widths) and TruncatedPermutation
                let hash = MyHash::new(perm.clone());
                let compress = MyCompress::new(perm.clone());
                let sponge: Vec<Value> = inp["sponge"].as_array().unwrap().iter().map(|k| {
                    let xs: Vec<F> = from_json(&k["in"]);
                    let d: [F; DIGEST] = hash.hash_iter(xs);
                    json!({"in": k["in"], "out": u32s(&d)})
                }).collect();
                let l: [F; DIGEST] = from_json::<F>(&inp["compress"]["left"]).try_into().unwrap();
                let r: [F; DIGEST] = from_json::<F>(&inp["compress"]["right"]).try_into().unwrap();
                let c = compress.compress([l, r]);
                // DuplexChallenger script: 0 observe(arg), 1 sample, 2 sample extension element (4 outputs),
                // 3 sample_bits(arg)
                let mut ch = Challenger::new(perm.clone());
                let mut outs: Vec<u32> = Vec::new();
                let ops = inp["challenger"]["ops"].as_array().unwrap();
                let args = inp["challenger"]["args"].as_array().unwrap();
                for (op, arg) in ops.iter().zip(args.iter()) {
                    let a = arg.as_u64().unwrap();
                    match op.as_u64().unwrap() {
                        0 => ch.observe(F::from_u64(a)),
                        1 => { let x: F = ch.sample(); outs.push(x.as_canonical_u32()); }
                        2 => {
                            let e: Challenge = ch.sample_algebra_element();
                            outs.extend(u32s(e.as_basis_coefficients_slice()));
                        }
                        3 => outs.push(ch.sample_bits(a as usize) as u32),
                        _ => unreachable!(),
                    }
                }
                // extension field
                let ea = Challenge::from_basis_coefficients_slice(&from_json::<F>(&inp["ext"]["a"])).unwrap();
                let eb = Challenge::from_basis_coefficients_slice(&from_json::<F>(&inp["ext"]["b"])).unwrap();
                // two-adic generators
                let mut gens = serde_json::Map::new();
                for (k, _) in inp["two_adic_generators"].as_object().unwrap() {
                    let bits: usize = k.parse().unwrap();
                    gens.insert(k.clone(), json!(F::two_adic_generator(bits).as_canonical_u32()));
                }
                // coset_lde_batch(..).bit_reverse_rows(): the committed LDE order
                let l = &inp["lde"];
                let (h, w) = (l["h"].as_u64().unwrap() as usize, l["w"].as_u64().unwrap() as usize);
                let added = l["added_bits"].as_u64().unwrap() as usize;
                let shift = F::from_u64(l["shift"].as_u64().unwrap());
                let mut vals: Vec<F> = Vec::with_capacity(h * w);
                for row in l["evals"].as_array().unwrap() { vals.extend(from_json::<F>(row)); }
                let lde = Dft::default().coset_lde_batch(RowMajorMatrix::new(vals, w), added, shift).bit_reverse_rows().to_row_major_matrix();
                let lde_rows: Vec<Vec<u32>> = (0..lde.height()).map(|i| u32s(&lde.row_slice(i).unwrap())).collect();
                json!({
                    "rc": round_constants(),
                    "permute": permute,
                    "sponge": sponge,
                    "compress": {"left": inp["compress"]["left"], "right": inp["compress"]["right"], "out": u32s(&c)},
                    "challenger": {"ops": inp["challenger"]["ops"], "args": inp["challenger"]["args"], "out": outs},
                    "ext": {"a": inp["ext"]["a"], "b": inp["ext"]["b"],
                            "mul": u32s((ea * eb).as_basis_coefficients_slice()),
                            "inv_a": u32s(ea.inverse().as_basis_coefficients_slice())},
                    "two_adic_generators": gens,
                    "lde": {"h": h, "w": w, "added_bits": added, "shift": l["shift"], "evals": l["evals"], "lde": lde_rows},
                })
            }

            fn config() -> MyConfig {
                let perm: Perm = $default_perm();
                let hash = MyHash::new(perm.clone());
                let compress = MyCompress::new(perm.clone());
                let val_mmcs = MyMmcs::new(hash, compress, CAP_HEIGHT);
                let challenge_mmcs = ChallengeMmcs::new(val_mmcs.clone());
                let fri_params = FriParameters {
                    max_log_arity: MAX_LOG_ARITY,
                    log_blowup: LOG_BLOWUP,
                    log_final_poly_len: LOG_FINAL_POLY_LEN,
                    num_queries: NUM_QUERIES,
                    commit_proof_of_work_bits: COMMIT_POW_BITS,
                    query_proof_of_work_bits: QUERY_POW_BITS,
                    mmcs: challenge_mmcs,
                };
                let pcs = MyPcs::new(Dft::default(), val_mmcs, fri_params);
                MyConfig::new(pcs, Challenger::new(perm))
            }

            /// Fibonacci over the extension field (tests/fib_lib.py builds the same op list): const 0,
            /// public expected_result, const 1, n - 1 additions, connect(last, expected_result).
            pub fn fibonacci_layer() -> Value {
                let mut builder = CircuitBuilder::<Challenge>::new();
                let expected = builder.alloc_public_input("expected_result");
                let mut a = builder.alloc_const(Challenge::ZERO, "F(0)");
                let mut b = builder.alloc_const(Challenge::ONE, "F(1)");
                for _ in 2..=FIB_N {
                    let next = builder.add(a, b);
                    a = b;
                    b = next;
                }
                builder.connect(b, expected);
                let circuit = builder.build().unwrap();
                let (mut fa, mut fb) = (F::ZERO, F::ONE);
                for _ in 2..=FIB_N { let t = fa + fb; fa = fb; fb = t; }
                let packing = TablePacking::new(1, 1).with_fri_params(LOG_FINAL_POLY_LEN, LOG_BLOWUP);
                let cfg = config();
                let (airs_degrees, primitive_columns, non_primitive_columns) =
                    get_airs_and_degrees_with_prep::<MyConfig, Challenge, 4>(&circuit, &packing, &[], &[], ConstraintProfile::Standard).unwrap();
                let (airs, log_degrees): (Vec<_>, Vec<usize>) = airs_degrees.into_iter().unzip();
                let prover_data = ProverData::from_airs_and_degrees(&cfg, &airs, &log_degrees);
                let cpd = CircuitProverData::new(prover_data, primitive_columns, non_primitive_columns);
                let mut runner = circuit.runner();
                runner.set_public_inputs(&[Challenge::from(fb)]).unwrap();
                let traces = runner.run().unwrap();
                let prover = BatchStarkProver::new(cfg).with_table_packing(packing);
                let proof: BatchStarkProof<MyConfig> = prover.prove_all_tables(&traces, &cpd).unwrap();
                prover.verify_all_tables::<Challenge>(&proof).unwrap();  // EF = the expected trace element field (D = 4)
                let outer = postcard::to_allocvec(&proof).unwrap();
                let inner = postcard::to_allocvec(&proof.proof).unwrap();
                json!({
                    "field": $key, "n": FIB_N, "fib": fb.as_canonical_u32(),
                    "fri": {"log_blowup": LOG_BLOWUP, "max_log_arity": MAX_LOG_ARITY, "cap_height": CAP_HEIGHT,
                            "log_final_poly_len": LOG_FINAL_POLY_LEN, "commit_pow_bits": COMMIT_POW_BITS,
                            "query_pow_bits": QUERY_POW_BITS, "num_queries": NUM_QUERIES},
                    "packing": {"public_lanes": 1, "alu_lanes": 1},
                    "rc": round_constants(),
                    "degree_bits": log_degrees,
                    "batch_stark_proof_postcard_hex": hex(&outer),
                    "batch_proof_postcard_hex": hex(&inner),
                })
            }
        }
    };
}

fn hex(b: &[u8]) -> String {
    b.iter().map(|x| format!("{x:02x}")).collect()
}

field_module!(koala, p3_koala_bear::KoalaBear, p3_koala_bear::Poseidon2KoalaBear<16>, p3_koala_bear::default_koalabear_poseidon2_16,
              p3_koala_bear::KOALABEAR_POSEIDON2_RC_16_EXTERNAL_INITIAL, p3_koala_bear::KOALABEAR_POSEIDON2_RC_16_INTERNAL,
              p3_koala_bear::KOALABEAR_POSEIDON2_RC_16_EXTERNAL_FINAL, "koala_bear");
field_module!(baby, p3_baby_bear::BabyBear, p3_baby_bear::Poseidon2BabyBear<16>, p3_baby_bear::default_babybear_poseidon2_16,
              p3_baby_bear::BABYBEAR_POSEIDON2_RC_16_EXTERNAL_INITIAL, p3_baby_bear::BABYBEAR_POSEIDON2_RC_16_INTERNAL,
              p3_baby_bear::BABYBEAR_POSEIDON2_RC_16_EXTERNAL_FINAL, "baby_bear");

fn main() {
    let golden = concat!(env!("CARGO_MANIFEST_DIR"), "/../../tests/golden");
    let inp: Value = serde_json::from_str(&fs::read_to_string(format!("{golden}/primitives.json")).unwrap()).unwrap();
    let out = json!({
        "provenance": "tools/rust_pin: upstream p3-* 0.6 + the reference's circuit-prover, run on the inputs of primitives.json",
        "fields": {
            "koala_bear": koala::primitives(&inp["fields"]["koala_bear"]),
            "baby_bear": baby::primitives(&inp["fields"]["baby_bear"]),
        }
    });
    fs::write(format!("{golden}/rust_primitives.json"), serde_json::to_string(&out).unwrap()).unwrap();
    fs::write(format!("{golden}/rust_fibonacci_layer_koala_bear.json"), serde_json::to_string(&koala::fibonacci_layer()).unwrap()).unwrap();
    fs::write(format!("{golden}/rust_fibonacci_layer_baby_bear.json"), serde_json::to_string(&baby::fibonacci_layer()).unwrap()).unwrap();
    println!("wrote rust_primitives.json and rust_fibonacci_layer_*.json under {golden}");
}
