/* p3r.h - C ABI of the MI355X-native batch-STARK prover for the p3-recursion
 * `prove_next_layer` hot path.
 *
 * The reference (Plonky3/Plonky3-recursion) is pure Rust and exposes NO FFI for a device
 * backend; the seam this library replaces is
 *
 *   BatchStarkProver::prove_all_tables      circuit-prover/src/batch_stark_prover.rs:1203-1222
 *     -> prove::<EF,D>                       circuit-prover/src/batch_stark_prover.rs:1275-1642
 *       -> p3_batch_stark::prove_batch       call site batch_stark_prover.rs:1595
 *   ProverData::from_airs_and_degrees        call sites recursion/src/recursion.rs:376,487,737,859
 *
 * plus the finer unit seams a Rust integrator can bind one at a time:
 *
 *   TableProver::batch_instance_d4 (Poseidon2 table)   batch_stark_prover/dynamic_air.rs:324-419
 *     -> Poseidon2CircuitAir::generate_trace_rows       poseidon2-circuit-air/src/air.rs:280-520
 *   Mmcs::commit / open_batch (MerkleTreeMmcs)          circuit-prover/src/config.rs:56-63,129
 *   TwoAdicSubgroupDft::coset_lde_batch                 circuit-prover/src/config.rs:55,131
 *   CryptographicPermutation<[F;16]>::permute           circuit-prover/src/config.rs:126-136
 *
 * Conventions (mirroring the reference's Result<_, String> stringification at
 * recursion/src/recursion.rs:34-36):
 *   - every int-returning call returns 0 on success and a negative P3R_E* code on failure;
 *     p3r_last_error(ctx) then returns a human-readable message (ctx may be NULL for
 *     failures of p3r_create itself);
 *   - one p3r_ctx per GPU, NOT thread-safe, one call in flight per ctx (RecursionOutput is
 *     !Send in the reference: recursion/src/recursion.rs:117-139);
 *   - every field element crossing this ABI is a CANONICAL u32 (< p); matrices are
 *     ROW-MAJOR exactly like p3_matrix::dense::RowMajorMatrix; extension-field elements are
 *     4 consecutive base coefficients (basis 1,x,x^2,x^3);
 *   - the caller owns all host buffers; the ctx owns all device memory;
 *   - there is NO CPU fallback: without a usable gfx950 device p3r_create fails.
 */
#ifndef P3R_H
#define P3R_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P3R_ABI_VERSION 8

enum {
  P3R_OK = 0,
  P3R_EINVAL = -1,   /* bad argument / shape (reference: InvalidProofShape-style errors) */
  P3R_ENODEV = -2,   /* no usable HIP device */
  P3R_EHIP = -3,     /* HIP runtime error */
  P3R_ENOMEM = -4,
  P3R_EUNSUPPORTED = -5, /* reference: BatchStarkProverError::UnsupportedDegree */
  P3R_EBUFFER = -6   /* caller buffer too small */
};

enum { P3R_FIELD_KOALA_BEAR = 0, P3R_FIELD_BABY_BEAR = 1 };

/* Mirrors the FRI/PCS parameters the examples build
 * (recursion/examples/common/mod.rs:464-486, recursive_fibonacci.rs:71-147) plus the
 * field selection of circuit-prover/src/config.rs:180-183. */
typedef struct p3r_config {
  uint32_t abi_version;        /* P3R_ABI_VERSION */
  uint32_t field;              /* P3R_FIELD_* */
  uint32_t ext_degree;         /* circuit extension degree D of the traces: 1 (base-field circuits, CircuitBuilder<F>: the
                                * base proof of recursive_fibonacci.rs:315-331; bus tuples (idx, v)), 4 (binomial
                                * x^4 = W), 2 / 6 / 8 (binomial x^D = ext_w, below), or 5 over KoalaBear
                                * (quintic trinomial x^5 + x^2 - 1: QuinticTrinomialExtensionField, proved under the
                                * same D = 4 STARK configuration as in circuit-prover/src/batch_stark_prover/
                                * tests.rs:844-1029).  Under D = 1 / D = 5 values are n x D / n x 4D, witness indices in
                                * the preprocessed columns are scaled by D and the Poseidon2 table is the compact-D1
                                * one (p3r_layer_desc).  The circuit boundary (p3r_circuit_create ...) runs such
                                * circuits too: constants carry D coefficients, inputs are n x D, Poseidon2 ops are
                                * base-mode permutations (p3r_op_kind). */
  uint32_t log_blowup;
  uint32_t max_log_arity;
  uint32_t cap_height;
  uint32_t log_final_poly_len;
  uint32_t commit_pow_bits;
  uint32_t query_pow_bits;
  uint32_t num_queries;
  int32_t device;              /* HIP device ordinal */
  /* Poseidon2 width-16 round constants, canonical, flat:
   *   [4][16] external-initial | [partial_rounds] internal | [4][16] external-final
   * (the p3_{koala,baby}_bear::*_POSEIDON2_RC_16_* statics a Rust caller passes through,
   * poseidon2-circuit-air/src/public_types.rs:48-54,220-226).  NULL selects the library's
   * built-in table, which is self-generated and NOT pinned to upstream. */
  const uint32_t* poseidon2_rc;
  uint32_t poseidon2_rc_len;
  /* Protocol details the in-tree reference does not pin (they live in un-vendored p3-* crates, DESIGN.md
   * section 4) are selectable, so that a run of tools/rust_pin against upstream is a configuration
   * change and not a code change:
   *   ext_choices      P3R_EXT_* bits below
   *   fri_log_arities  optional explicit FRI folding schedule (one log2 arity per commit phase, tallest
   *                    first); NULL selects the rule min(max_log_arity, distance to the final height,
   *                    distance to the next roll-in height).  Must reach every input height and the
   *                    final height exactly, and no entry may exceed max_log_arity (the verifier's
   *                    bound on a step's arity).  Prover and verifier must be given the same schedule. */
  uint32_t ext_choices;
  const uint8_t* fri_log_arities;
  uint32_t fri_log_arities_len;
  /*   proof_layout     optional order of the fields of the postcard-serialised structs: 18 bytes, three
   *                    permutations batch[5] | fri[5] | opened[8] (csrc/host_transcript.h::ProofLayout
   *                    lists the fields); NULL = the order read off the in-tree destructuring patterns. */
  const uint8_t* proof_layout;
  uint32_t proof_layout_len;
  /* ABI version 4.  W of the binomial extension x^D = W for ext_degree 2, 6, 8 (canonical; BinomiallyExtendable<D>::W
   * of the caller's field crate - the value the proof carries as w_binomial).  Such layers hold the primitive tables
   * and Recompose (batch_stark_prover/tests.rs:486: test_koalabear_batch_stark_extension_field_d8) and enter at the
   * prove_all_tables boundary.  0 for the other degrees: 4 uses the field's W (3 / 11), 1 and 5 have none. */
  uint32_t ext_w;
  /* ABI version 4.  Degree of the STARK's CHALLENGE field (SC::Challenge): 0 / 4 = the quartic binomial extension
   * (every BASELINE configuration); 5 = KoalaBear's quintic trinomial extension, the configuration of
   * koala_bear_quintic_params (test-utils/src/lib.rs:414-460; recursive_fibonacci --quintic;
   * recursion/tests/fibonacci_batch_stark_prover_quintic.rs).  Extension elements of the proof then hold five words. */
  uint32_t challenge_degree;
  /* ABI version 6.  Constants of the WIDTH-32 Poseidon2 permutation (Poseidon2{Koala,Baby}Bear<32>, the permutation of
   * the arity-4 MMCS: Poseidon2Config::{KOALA,BABY}_BEAR_D4_W32, circuit/src/ops/poseidon2_perm/config.rs:88-100,
   * :164-172; 31 / 30 partial rounds).  poseidon2_w32_rc: [4][32] external-initial | [partial] internal | [4][32]
   * external-final, canonical (p3_{koala,baby}_bear::*_POSEIDON2_RC_32_*, poseidon2-circuit-air/src/public_types.rs:
   * 179-187,396-404); poseidon2_w32_diag: the 32 entries of the internal layer's diagonal
   * (GenericPoseidon2LinearLayers<32> of the field crate).  Both live in un-vendored crates, so - like the width-16
   * round constants - they are the caller's DATA; NULL selects self-generated defaults (tools/
   * gen_poseidon2_constants.py: Grain LFSR constants, a diagonal of small integers and inverse powers of two), which
   * are NOT pinned to upstream.  Used by the width-32 Poseidon2 table (P3R_AIR_POSEIDON2_W32). */
  const uint32_t* poseidon2_w32_rc;
  uint32_t poseidon2_w32_rc_len;
  const uint32_t* poseidon2_w32_diag;   /* 32 canonical values, or NULL */
  /* ABI version 6.  Arity of the prover's own MMCS: 0 / 2 = binary trees over the width-16 permutation
   * (PaddingFreeSponge<Perm16, 16, 8, 8>, TruncatedPermutation<Perm16, 2, 8, 16>: every BASELINE configuration);
   * 4 = the arity-4 MMCS of `recursive_aggregation --arity4` (recursion/examples/recursive_aggregation.rs:1024-1046:
   * MerkleTreeMmcs<.., 4, 8> with PaddingFreeSponge<Perm32, 32, 24, 8> leaves and TruncatedPermutation<Perm32, 4, 8, 32>
   * levels over the width-32 permutation above; the challenger keeps the width-16 permutation).  Trace, quotient and
   * FRI commit-phase trees, their opening proofs (step - 1 sibling digests per level, recursion/src/pcs/mmcs.rs:
   * 866-1316) and both verifiers follow.  cap_height must be 0 with arity 4. */
  uint32_t mmcs_arity;
  /* ABI version 7.  ZK: the PCS is HidingFriPcs (create_config_zk, recursion/examples/common/mod.rs:511-553:
   * `MyPcsZk::new(dft, val_mmcs, fri_params, 2, SmallRng::seed_from_u64(rng_seed))` over the SAME non-hiding MMCS;
   * `recursive_fibonacci --zk`, `recursive_aggregation --zk`, recursion/tests/fibonacci_batch_stark_prover_zk.rs).
   *   zk = 1: every committed matrix (preprocessed, main, permutation, quotient chunks) is committed over the EXTENDED
   *     trace domain of twice the height - original rows interleaved with random rows - with `num_random_codewords`
   *     random columns appended; a random round (Challenge::DIMENSION + num_random_codewords columns per instance) is
   *     committed and opened at zeta; the quotient has 2^(log_chunks + 1) chunks, each masked by a random multiple of
   *     its vanishing polynomial; degree_bits in the proof and in the preprocessed metadata are the extended ones
   *     (recursion.rs:374); the opening proof is HidingFriPcs's tuple (random opened values, FriProof).  What the
   *     verifiers enforce is recursion/src/verifier/batch_stark.rs:424-428,487-490,536,623-661,701-735,855-864,1116-1260;
   *     the prover side of HidingFriPcs lives in the un-vendored p3-fri crate and its proofs are randomised, so byte
   *     parity with upstream is undefined by construction - the construction is written from those acceptance
   *     conditions.  The preprocessed round is padded with zeros, not random values: its commitment
   *     depends on the circuit shape only.
   *   num_random_codewords: 0 selects 2 (the examples' value); 1..8.
   *   zk_key (ABI version 8; versions up to 7 had a 64-bit `zk_seed` here): 256-bit key of the random values - the
   *     counterpart of HidingFriPcs's `rng: R`, which is the caller's to choose (common/mod.rs:536-542).  The values come
   *     from a keyed counter-based generator, ChaCha with 8 rounds over (key, proofs made so far by the ctx, round,
   *     matrix, cell), field elements by rejection sampling (csrc/zk_rand.h).  By default p3r_create mixes 128 bits of
   *     operating-system entropy (getrandom) into the key: two contexts, or two runs, never mask different witnesses with
   *     the same values, whatever the caller passes - an all-zero key is fine.  P3R_EXT_ZK_DETERMINISTIC in ext_choices
   *     takes the key as it is and lets p3r_zk_set_nonce move the proof counter: reproducible proofs for tests and
   *     replay, and NO hiding if a (key, nonce) pair is ever used for two witnesses. */
  uint32_t zk;
  uint32_t num_random_codewords;
  uint32_t zk_key[8];
  /* ABI version 8.  MerkleTreeHidingMmcs for the input MMCS and (through ExtensionMmcs) the FRI commit-phase MMCS -
   * "the upstream-recommended ZK setup" of recursion/tests/zk_hiding_mmcs.rs (SALT_ELEMS = 4 there): every committed
   * matrix gets mmcs_salt_elems random elements per row, a leaf preimage is the concatenation of [row | salt] over the
   * matrices of its height class (recursion/src/pcs/mmcs.rs:315-413, :430-510), and an MMCS opening proof is the tuple
   * (per-matrix salts, sibling digests) (:763-790).  0: the plain MerkleTreeMmcs; 1..16.  The salts come from the
   * generator of zk_key (which is therefore keyed also when zk = 0); both arities; independent of zk, as the MMCS type is
   * independent of the PCS type upstream.  The preprocessed commitment is salted too (it is made once per circuit). */
  uint32_t mmcs_salt_elems;
} p3r_config;
/* LogUp: one auxiliary column per interaction instead of packing same-bus interactions greedily up to
 * the degree budget 2^log_chunks + 1 (batch_stark_prover.rs:925-941 `pack_same_bus`). */
#define P3R_EXT_LOOKUP_UNPACKED 1u
/* ABI version 7.  The width-32 permutation - the one of the arity-4 MMCS (mmcs_arity = 4) and of the width-32 Poseidon2
 * table - has its constants in un-vendored crates, and the library's built-in defaults for them are SELF-GENERATED: a
 * proof made with them verifies nowhere else.  Using that permutation with poseidon2_w32_rc or poseidon2_w32_diag NULL
 * therefore needs this acknowledgement in ext_choices; without it p3r_create (mmcs_arity = 4), the width-32 entry points,
 * p3r_prep_create / p3r_layer_create of a batch holding P3R_AIR_POSEIDON2_W32, p3r_verify_batch and p3r_mmcs_verify
 * refuse with P3R_EINVAL.  (The width-16 defaults are believed to be upstream's; the width-32 ones are known not to be.) */
#define P3R_EXT_UNPINNED_W32_DEFAULTS 2u
/* ABI version 8.  ZK: use p3r_config.zk_key as it is (no operating-system entropy mixed in) and allow p3r_zk_set_nonce:
 * the proofs of a context are then a function of (key, nonce, inputs) - what the parity tests against the CPU oracle
 * need, and what a deployment must NOT set (see zk_key). */
#define P3R_EXT_ZK_DETERMINISTIC 4u

typedef struct p3r_ctx p3r_ctx;
typedef struct p3r_dmat p3r_dmat; /* device-resident matrix (power-of-two height) */
typedef struct p3r_tree p3r_tree; /* device-resident MMCS prover data */

/* ---- context ---- */
p3r_ctx* p3r_create(const p3r_config* cfg);
void p3r_destroy(p3r_ctx* ctx);
const char* p3r_last_error(const p3r_ctx* ctx);
/* Width of the Poseidon2 circuit-table main trace (166 KoalaBear / 300 BabyBear), i.e.
 * BaseAir::width of Poseidon2CircuitAir (poseidon2-circuit-air/src/air.rs:561-585). */
uint32_t p3r_poseidon2_trace_width(const p3r_ctx* ctx);
/* Number of round constants the configured field expects (148 / 141). */
uint32_t p3r_poseidon2_num_constants(const p3r_ctx* ctx);
/* The round constants in use (canonical, the flat layout of p3r_config.poseidon2_rc); `out` holds
 * p3r_poseidon2_num_constants values. */
int p3r_poseidon2_round_constants(const p3r_ctx* ctx, uint32_t* out);
/* ABI version 7.  Proofs made so far under a ZK configuration (the state of the hiding PCS's RNG): read it, or - under
 * P3R_EXT_ZK_DETERMINISTIC only (ABI version 8; P3R_EINVAL otherwise: replaying a nonce repeats the masks) - set it to
 * replay / skip ahead (parity tests set it so that the CPU oracle can be given the same value). */
uint64_t p3r_zk_nonce(const p3r_ctx* ctx);
int p3r_zk_set_nonce(p3r_ctx* ctx, uint64_t nonce);
/* Blocks until all work queued on the ctx's stream has completed. */
int p3r_sync(p3r_ctx* ctx);
/* Device memory a ctx has released stays in its pool for reuse (a 2^20-row prove keeps several GB
 * cached between proofs; context.h::DevPool).  p3r_trim returns the cached blocks of this ctx to the
 * driver and reports how many bytes that was; an allocation that fails with out-of-memory trims the
 * pools of EVERY ctx of the process before it is reported as P3R_ENOMEM. */
int p3r_trim(p3r_ctx* ctx, uint64_t* freed_bytes);

/* ---- device matrices (inputs stay resident in HBM between calls) ---- */
p3r_dmat* p3r_dmat_upload(p3r_ctx* ctx, const uint32_t* rowmajor, size_t height, size_t width);
p3r_dmat* p3r_dmat_alloc(p3r_ctx* ctx, size_t height, size_t width);
int p3r_dmat_download(p3r_ctx* ctx, const p3r_dmat* m, uint32_t* rowmajor_out);
size_t p3r_dmat_height(const p3r_dmat* m);
size_t p3r_dmat_width(const p3r_dmat* m);
void p3r_dmat_free(p3r_ctx* ctx, p3r_dmat* m);

/* ---- K3: Poseidon2 (CryptographicPermutation + circuit-table trace fill) ---- */

/* n independent width-16 permutations; in/out are n x 16 row-major. */
int p3r_poseidon2_permute_batch(p3r_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n);
/* Device-resident form: states is an n x 16 matrix permuted in place. */
int p3r_poseidon2_permute_dmat(p3r_ctx* ctx, p3r_dmat* states);

/* Flattened Vec<Poseidon2CircuitRow<F>> (circuit/src/ops/poseidon2_perm/trace.rs:94-125),
 * already padded to a power of two by the caller exactly as Poseidon2Prover does
 * (circuit-prover/src/batch_stark_prover/poseidon2.rs:1125-1140). Only the fields the
 * main trace depends on are carried; the CTL fields feed the preprocessed trace. */
typedef struct p3r_p2_rows {
  size_t n;                       /* power of two */
  const uint32_t* input_values;   /* n x 16 row-major */
  const uint8_t* new_start;       /* n */
  const uint8_t* merkle_path;     /* n */
  const uint8_t* mmcs_bit;        /* n */
  const uint32_t* mmcs_index_sum; /* n */
} p3r_p2_rows;
/* ABI version 6.  Rows of the WIDTH-32 table (arity-4 compression shape, 4 * CAPACITY_EXT == WIDTH_EXT:
 * Poseidon2CircuitRow with mmcs_bit2; poseidon2-circuit-air/src/air.rs:370-432).  The main trace is
 * [Poseidon2Cols<32> | mmcs_bit | mmcs_bit2 | mmcs_bit * mmcs_bit2 | mmcs_index_sum]; the index accumulator of a
 * Merkle continuation row is 4 * previous + mmcs_bit + 2 * mmcs_bit2. */
typedef struct p3r_p2w_rows {
  size_t n;                       /* un-padded row count (0: the layer has no width-32 table) */
  const uint32_t* input_values;   /* n x 32 row-major */
  const uint8_t* new_start;       /* n */
  const uint8_t* merkle_path;     /* n */
  const uint8_t* mmcs_bit;        /* n: low bit of the position pos = mmcs_bit + 2 * mmcs_bit2 */
  const uint8_t* mmcs_bit2;       /* n: high bit */
  const uint32_t* mmcs_index_sum; /* n */
} p3r_p2w_rows;

/* Unit seams of the width-32 permutation and its table (ABI 6): Poseidon2{Koala,Baby}Bear<32> on n row-major
 * canonical states; generate_trace_rows of the arity-4 layout (trace_out is n x p3r_poseidon2_w32_trace_width(),
 * n a power of two). */
int p3r_poseidon2_w32_permute_batch(p3r_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n);
int p3r_poseidon2_w32_trace_fill(p3r_ctx* ctx, const p3r_p2w_rows* rows, uint32_t* trace_out);
uint32_t p3r_poseidon2_w32_trace_width(const p3r_ctx* ctx);

/* Poseidon2CircuitAir::generate_trace_rows: trace_out is n x p3r_poseidon2_trace_width(). */
int p3r_poseidon2_trace_fill(p3r_ctx* ctx, const p3r_p2_rows* rows, uint32_t* trace_out);
/* Same, leaving the trace in HBM. */
p3r_dmat* p3r_poseidon2_trace_fill_dmat(p3r_ctx* ctx, const p3r_p2_rows* rows);
/* Rows kept resident in HBM (so a prove can start from device-resident inputs). */
typedef struct p3r_p2_dev p3r_p2_dev;
p3r_p2_dev* p3r_p2_rows_upload(p3r_ctx* ctx, const p3r_p2_rows* rows);
void p3r_p2_rows_free(p3r_ctx* ctx, p3r_p2_dev* rows);
p3r_dmat* p3r_poseidon2_trace_fill_dev(p3r_ctx* ctx, const p3r_p2_dev* rows);

/* ---- K5: coset low-degree extension (TwoAdicSubgroupDft::coset_lde_batch followed by
 * bit_reverse_rows, as TwoAdicFriPcs::commit applies it) ----
 * evals: h x w evaluations over the size-h subgroup (natural order).
 * out  : (h << added_bits) x w, row i = evaluation at shift * w_{h<<added_bits}^{bitrev(i)}. */
int p3r_coset_lde(p3r_ctx* ctx, const uint32_t* evals, size_t h, size_t w, uint32_t added_bits,
                  uint32_t shift, uint32_t* out);
p3r_dmat* p3r_coset_lde_dmat(p3r_ctx* ctx, const p3r_dmat* evals, uint32_t added_bits,
                             uint32_t shift);

/* ---- K6: MMCS (MerkleTreeMmcs<PaddingFreeSponge<Perm,16,8,8>, TruncatedPermutation<Perm,2,8,16>>) ---- */

typedef struct p3r_matrix {
  const uint32_t* values; /* height x width row-major */
  size_t height;          /* power of two */
  size_t width;
} p3r_matrix;

/* Mmcs::commit over host matrices of mixed heights; cap_out receives (8 << cap_height)
 * elements; *tree_out (optional, may be NULL) receives the prover data for p3r_mmcs_open. */
int p3r_mmcs_commit(p3r_ctx* ctx, const p3r_matrix* mats, size_t n_mats, uint32_t* cap_out,
                    p3r_tree** tree_out);
/* Same over device-resident matrices; the tree borrows (does not own) the matrices. */
int p3r_mmcs_commit_dmat(p3r_ctx* ctx, const p3r_dmat* const* mats, size_t n_mats,
                         uint32_t* cap_out, p3r_tree** tree_out);
/* Mmcs::open_batch(index): writes, for each committed matrix in commit order, its row
 * (index >> (log_max_height - log_height)) into opened_values (concatenated, sum of widths
 * elements), and the sibling digests bottom-up into proof_out
 * (p3r_tree_proof_len(tree) x 8 elements: log_max_height - cap_height digests for a binary tree; for an arity-4 tree
 * step - 1 digests per level in ascending position, the opened node's own left out). */
int p3r_mmcs_open(p3r_ctx* ctx, const p3r_tree* tree, size_t index, uint32_t* opened_values,
                  uint32_t* proof_out);
/* Digests of one opening proof of this tree (`Mmcs::Proof = Vec<[F; 8]>`). */
size_t p3r_tree_proof_len(const p3r_tree* tree);
/* Mmcs::verify_batch on the host, no context: heights / widths of the committed matrices in commit order, the opened
 * rows concatenated in that order, `proof` = proof_len digests.  Honours cfg->mmcs_arity (and, for arity 4, the
 * width-32 constants of cfg).  Returns P3R_OK when the opening is accepted, P3R_EINVAL with the reason in err_buf
 * otherwise. */
int p3r_mmcs_verify(const p3r_config* cfg, const uint32_t* cap, size_t n_mats, const size_t* heights, const size_t* widths,
                    size_t index, const uint32_t* opened_values, const uint32_t* proof, size_t proof_len, char* err_buf,
                    size_t err_cap);
/* ABI version 8.  The same for a hiding MMCS (cfg->mmcs_salt_elems > 0: MerkleTreeHidingMmcs::verify_batch,
 * recursion/src/pcs/mmcs.rs:315-413): `salts` = n_mats x mmcs_salt_elems, the first half of the opening proof
 * `(salts, siblings)`; every leaf preimage is the concatenation of [row | salt] per matrix of its height class. */
int p3r_mmcs_verify_salted(const p3r_config* cfg, const uint32_t* cap, size_t n_mats, const size_t* heights, const size_t* widths,
                           size_t index, const uint32_t* opened_values, const uint32_t* salts, const uint32_t* proof,
                           size_t proof_len, char* err_buf, size_t err_cap);
size_t p3r_tree_log_max_height(const p3r_tree* tree);
size_t p3r_tree_total_width(const p3r_tree* tree);
void p3r_tree_free(p3r_ctx* ctx, p3r_tree* tree);

/* ---- batch-STARK proving (the chosen drop-in seam) ----
 *
 * p3r_prep_create  == ProverData::from_airs_and_degrees + CircuitProverData::new as called by
 *                     build_next_layer_prep (recursion/src/recursion.rs:342-394): LDE and one
 *                     global MMCS commitment of every AIR's preprocessed trace, kept in HBM.
 * p3r_prove_batch  == p3_batch_stark::prove_batch(&config, &instances, &prover_data)
 *                     (circuit-prover/src/batch_stark_prover.rs:1543-1595): takes the per-table
 *                     main traces in instance order [Const, Public, Alu, dynamic...]
 *                     (batch_stark_prover.rs:1493-1519) and writes the postcard bytes of
 *                     BatchProof (recursion/src/types/proof.rs:403-409).
 */
/* P3R_AIR_POSEIDON2_W32 (ABI 6): Poseidon2CircuitAir{Koala,Baby}BearD4Width32, the table of the arity-4 MMCS rows -
 * eval_arity4 (poseidon2-circuit-air/src/air.rs:1178-1342), 48 preprocessed columns, 16 bus interactions. */
enum { P3R_AIR_CONST = 0, P3R_AIR_PUBLIC = 1, P3R_AIR_ALU = 2, P3R_AIR_POSEIDON2 = 3, P3R_AIR_RECOMPOSE = 4, P3R_AIR_POSEIDON2_W32 = 5 };

/* One CircuitTableAir (circuit-prover/src/common.rs:90-100) of extension degree D = 4. */
typedef struct p3r_air_desc {
  uint32_t kind;                /* P3R_AIR_* */
  uint32_t lanes;               /* TablePacking lanes of this table (packing.rs:10-27) */
  uint32_t horner_packed_steps; /* ALU only: TablePacking::horner_packed_steps, 2..8 */
  uint32_t coeff_lookups;       /* Recompose only: challenger.d() != D (backend/fri.rs:693-721) */
} p3r_air_desc;

typedef struct p3r_prep p3r_prep;

/* prep_mats[i] = BaseAir::preprocessed_trace() of instance i (row-major, padded height).
 * commit_out receives the global preprocessed commitment (8 << cap_height elements). */
p3r_prep* p3r_prep_create(p3r_ctx* ctx, const p3r_air_desc* airs, const p3r_matrix* prep_mats,
                          size_t n_instances, uint32_t* commit_out);
void p3r_prep_free(p3r_ctx* ctx, p3r_prep* prep);

#define P3R_PROOF_QUINTIC_CHALLENGE 2u /* flags of the proof PARSERS (p3r_batch_proof_len*, p3r_batch_stark_proof_parse):
                                        * extension elements are five words (p3r_config.challenge_degree = 5) */
#define P3R_PROOF_ZK 4u /* flags of the proof PARSERS: the proof is a hiding PCS's (p3r_config.zk = 1): its opening proof
                        * is the tuple (random opened values, FriProof) - the proof TYPE, which the bytes do not announce */
#define P3R_PROOF_SALTED 8u /* flags of the proof PARSERS (ABI 8): the MMCSs are hiding ones (p3r_config.mmcs_salt_elems > 0):
                            * every MMCS opening proof is the tuple (per-matrix salts, sibling digests) */
#define P3R_PROVE_CANONICAL_FIELD_ENCODING 1u /* flags: write canonical u32 instead of the
                                               * Montgomery word p3-monty-31's serde emits */

/* Main traces resident in HBM (n_instances device matrices, instance order). On success
 * *proof_len bytes of proof_buf hold the serialized BatchProof; P3R_EBUFFER reports the
 * needed size through *proof_len. P3R_EINVAL with "do not satisfy the constraints" is the
 * counterpart of prove_batch's internal-inconsistency panic. */
int p3r_prove_batch(p3r_ctx* ctx, const p3r_prep* prep, const p3r_dmat* const* main_traces,
                    size_t n_instances, uint32_t flags, uint8_t* proof_buf, size_t proof_cap,
                    size_t* proof_len);
/* The proof of the last prove call on this context that returned P3R_EBUFFER (any of p3r_prove_batch[_host],
 * p3r_prove_all_tables[_resident], p3r_prove_next_layer[_resident]): the bytes are KEPT, not recomputed - a second prove
 * call costs a second proof, and under zk = 1 or a hiding MMCS it IS another proof, of another length, so retrying with
 * the reported size can fail again.  The binding of a growable byte vector (Vec<u8>): call, on P3R_EBUFFER resize to
 * *proof_len and take.  The bytes are dropped by the next prove call and by a successful take; P3R_EINVAL when none is
 * waiting, P3R_EBUFFER (size in *proof_len) when proof_cap is still too small. */
int p3r_take_proof(p3r_ctx* ctx, uint8_t* proof_buf, size_t proof_cap, size_t* proof_len);
/* Same from host matrices (row-major canonical). */
int p3r_prove_batch_host(p3r_ctx* ctx, const p3r_prep* prep, const p3r_matrix* main_traces,
                         size_t n_instances, uint32_t flags, uint8_t* proof_buf, size_t proof_cap,
                         size_t* proof_len);

/* ---- prove_all_tables: the recursion layer as the reference sees it ----
 *
 * p3r_layer_create == build_next_layer_prep (recursion/src/recursion.rs:342-394): everything
 *   that depends only on the verifier-circuit shape - get_airs_and_degrees_with_prep's
 *   preprocessed op lists (circuit-prover/src/common.rs:127-390), the ALU lane schedule
 *   (circuit-prover/src/air/alu_air.rs:349-463), the preprocessed traces and their commitment.
 *   It is the object NextLayerPrepCache caches (recursion.rs:295-298).
 * p3r_prove_all_tables == BatchStarkProver::prove_all_tables(&traces, &circuit_prover_data)
 *   (circuit-prover/src/batch_stark_prover.rs:1203-1222): builds the five main traces in
 *   instance order [Const, Public, Alu, Poseidon2, Recompose] (K1-K3) and proves them.
 *   It returns the inner BatchProof bytes; the Rust shim wraps them with the metadata fields
 *   of BatchStarkProof (batch_stark_prover.rs:1631-1641), which it already owns.
 */
typedef struct p3r_layer_desc_counts {
  size_t n_const, n_public, n_alu, n_p2, n_recompose; /* ops / rows before padding */
  /* ABI version 5.  Rows of the SECOND Recompose table, `recompose/coeff`, of a layer that holds both kinds: a
   * backend created with coefficient lookups registers the two table provers side by side
   * (recompose_table_provers(lanes, true), batch_stark_prover.rs:1914-1932: [`recompose`, `recompose/coeff`]) and a
   * verifier circuit uses both - plain recomposition in the challenger and the MMCS gadgets
   * (recursion/src/challenger/circuit.rs:206-384, pcs/mmcs.rs:106-140), the coefficient kind for decomposition links
   * (circuit_builder.rs:1438-1477).  0: one Recompose table, of the kind recompose_coeff_lookups names. */
  size_t n_recompose_coeff;
  /* ABI version 6.  Rows of the width-32 Poseidon2 table (`poseidon2_perm/<field>_d4_w32`), proved right after the
   * width-16 one - the order a mixed-config verifier circuit enables them in (W16 challenger, W32 MMCS:
   * recursion/examples/recursive_aggregation.rs:902-1000).  D = 4 circuits; 0: no such table. */
  size_t n_p2w;
} p3r_layer_desc_counts;

typedef struct p3r_layer_desc {
  p3r_layer_desc_counts counts;
  /* TablePacking (circuit-prover/src/batch_stark_prover/packing.rs:10-27) */
  uint32_t public_lanes, alu_lanes, horner_packed_steps, recompose_lanes, min_trace_height;
  /* per-op preprocessed data exactly as get_airs_and_degrees_with_prep leaves it */
  const uint32_t* const_prep;     /* n_const x 2: [ext_mult, D*witness_idx]   (common.rs:353-368) */
  const uint32_t* public_prep;    /* n_public x 2                              (common.rs:324-351) */
  const uint32_t* alu_prep13;     /* n_alu x 13: AluPrepLaneCols               (common.rs:198-323) */
  const uint32_t* recompose_prep; /* n_recompose x 2: [D*output_idx, out_mult]; with recompose_coeff_lookups (below)
                                   * n_recompose x (2 + 2D): ... then (D*coeff_idx_i, coeff_mult_i) for i < D */
  /* Poseidon2CircuitRow CTL fields after poseidon_preprocess_for_prover
   * (circuit-prover/src/batch_stark_prover.rs:97-246); witness ids are NOT yet D-scaled */
  const uint8_t* p2_new_start;        /* n_p2 */
  const uint8_t* p2_merkle_path;      /* n_p2 */
  const uint8_t* p2_mmcs_ctl_enabled; /* n_p2 */
  const uint8_t* p2_in_ctl;           /* n_p2 x IL */
  const uint32_t* p2_input_indices;   /* n_p2 x IL */
  const uint32_t* p2_out_ctl;         /* n_p2 x OL, multiplicity as a canonical field element */
  const uint32_t* p2_output_indices;  /* n_p2 x OL */
  const uint32_t* p2_mmcs_index_sum_idx; /* n_p2 */
  /* IL x OL = 4 x 2 limbs of four base elements for ext_degree 4 (the D4 width-16 table).  Under ext_degree 5 the
   * table is the compact-D1 one, KOALA_BEAR_D1_W16 on the 5-slot witness bus (poseidon2-circuit-air/src/air.rs:730-763,
   * circuit-prover/src/batch_stark_prover/poseidon2.rs:1244-1285): IL x OL = 16 x 8, one witness per state element,
   * and p2_absorb_len (n_p2, NULL = zeros) is the prefix-free sponge length tag the executor writes into the header
   * (circuit/src/ops/poseidon_perm/executor.rs:720-741).  Since ABI version 4. */
  const uint8_t* p2_absorb_len;
  /* 1: the Recompose table is the "recompose/coeff" variant (NpoTypeId::recompose_with_coeff_lookups, circuit/src/ops/
   * npo.rs:53-60; circuit-prover/src/air/recompose_air.rs:196-226): each coefficient is also a bus tuple
   * (D*coeff_idx, c, 0, ..) with the multiplicity batch_stark_prover/recompose.rs:341-352 computes.  It is what a backend
   * registers when the permutation's degree differs from the circuit's (backend/fri.rs:693-721, :741-852: always under
   * ext_degree 5, where the permutation is the D1 one).  Since ABI version 4. */
  uint32_t recompose_coeff_lookups;
  /* ABI version 5.  n_recompose_coeff x (2 + 2D): the rows of the second Recompose table (then recompose_prep holds
   * the plain kind and recompose_coeff_lookups must be 0).  The batch lists `recompose` before `recompose/coeff`. */
  const uint32_t* recompose_coeff_prep;
  /* ABI version 6.  n_p2w x 48: the ASSEMBLED preprocessed rows of the width-32 table, Poseidon2PreprocessedRow<8, 6>
   * as extract_preprocessed_from_operations + the prover's multiplicity pass leave them (poseidon-circuit-cols/src/
   * preprocessed.rs; witness indices already D-scaled, canonical): 8 x {idx, in_ctl, normal_chain_sel,
   * merkle_chain_sel} | 6 x {idx, out_ctl} | witness idx of mmcs_bit | witness idx of mmcs_bit2 | new_start |
   * merkle_path.  Padding rows are added here (air.rs:613-649). */
  const uint32_t* p2w_prep;
} p3r_layer_desc;

/* Flattened Traces<EF> (circuit/src/tables/mod.rs:49-62), canonical; D = p3r_config.ext_degree. */
typedef struct p3r_traces {
  size_t n_const;     const uint32_t* const_values;     /* n x D (D = p3r_config.ext_degree) */
  size_t n_public;    const uint32_t* public_values;    /* n x D */
  size_t n_alu;       const uint32_t* alu_values;       /* n x 4D: AluTrace.values [a,b,c,out] */
  p3r_p2_rows p2;     /* n = un-padded Poseidon2 row count (any n, padding is done here) */
  size_t n_recompose; const uint32_t* recompose_values; /* n x D */
  size_t n_recompose_coeff; const uint32_t* recompose_coeff_values; /* n x D: rows of the second Recompose table (ABI 5) */
  p3r_p2w_rows p2w;   /* rows of the width-32 Poseidon2 table (ABI 6; n = 0 without one) */
} p3r_traces;

typedef struct p3r_layer p3r_layer;
typedef struct p3r_dtraces p3r_dtraces;

p3r_layer* p3r_layer_create(p3r_ctx* ctx, const p3r_layer_desc* desc, uint32_t* commit_out);
void p3r_layer_free(p3r_ctx* ctx, p3r_layer* layer);
/* Padded heights of [Const, Public, ALU, Poseidon2, Recompose]; 0 = the table is not part of the
 * batch: a non-primitive table with no rows is left out, as `batch_instance_*` returning None does
 * (batch_stark_prover/poseidon2.rs:1089-1092, recompose.rs:77-80). */
int p3r_layer_table_heights(const p3r_layer* layer, size_t heights_out[5]);
/* Padded height of the width-32 Poseidon2 table (ABI version 6); 0 = absent. */
int p3r_layer_p2w_height(const p3r_layer* layer, size_t* height_out);
/* Padded height of the second Recompose table (`recompose/coeff` next to `recompose`, ABI version 5); 0 = absent. */
int p3r_layer_recompose_coeff_height(const p3r_layer* layer, size_t* height_out);
/* 1: the table at position 4 is the `recompose/coeff` kind (p3r_layer_desc.recompose_coeff_lookups, or a circuit whose
 * Recompose ops all carry aux = 1) - the name the proof's metadata gives it. */
int p3r_layer_recompose_kind(const p3r_layer* layer, uint32_t* coeff_lookups_out);
/* The packing the proof was made with: Public / ALU lanes fall back to 1 when the table holds at
 * most the dummy op (reduce_lanes_if_dummy, batch_stark_prover.rs:1305-1318); BatchStarkProof
 * stores this effective packing (:1617-1622). */
int p3r_layer_effective_lanes(const p3r_layer* layer, uint32_t* public_lanes, uint32_t* alu_lanes);

/* Per-proof inputs made resident in HBM once; a prove can then be repeated without PCIe. */
p3r_dtraces* p3r_traces_upload(p3r_ctx* ctx, const p3r_layer* layer, const p3r_traces* traces);
void p3r_traces_free(p3r_ctx* ctx, p3r_dtraces* traces);

int p3r_prove_all_tables(p3r_ctx* ctx, const p3r_layer* layer, const p3r_traces* traces, uint32_t flags,
                         uint8_t* proof_buf, size_t proof_cap, size_t* proof_len);
int p3r_prove_all_tables_resident(p3r_ctx* ctx, const p3r_layer* layer, const p3r_dtraces* traces,
                                  uint32_t flags, uint8_t* proof_buf, size_t proof_cap, size_t* proof_len);
/* K1-K3 only: the main trace of table `table` (0..4; 5 = the second Recompose table) as a device matrix (parity tests). */
p3r_dmat* p3r_layer_build_main_trace(p3r_ctx* ctx, const p3r_layer* layer, const p3r_dtraces* traces,
                                     uint32_t table);

/* ---- verification -------------------------------------------------------------------------------
 * `p3_batch_stark::verify_batch(&config, &airs, &proof, &public_values, &common)` as
 * `BatchStarkProver::verify_all_tables` calls it (circuit-prover/src/batch_stark_prover.rs:1230-1268,
 * 1649-1727), for proofs made with the same p3r_config.  `airs` are the proved tables in instance
 * order, `preprocessed_commitment` the (1 << cap_height) x 8 canonical digests of the
 * CircuitProverData, `degree_bits[i]` the log2 trace height the verifier's preprocessed metadata
 * holds for instance i (CommonData.preprocessed.instances[i].degree_bits): a proof declaring any
 * other degree is rejected as recursion/src/verifier/batch_stark.rs:793 does (InvalidProofShape) -
 * the prover must not choose the domains.  Host code, no device and no p3r_ctx needed.  Returns P3R_OK when the proof
 * is accepted; otherwise an error code and the reason in err_buf (the analogue of
 * BatchStarkProverError::Verify(String)). */
int p3r_verify_batch(const p3r_config* cfg, const p3r_air_desc* airs, size_t n_airs,
                     const uint32_t* preprocessed_commitment, const uint32_t* degree_bits, const uint8_t* proof,
                     size_t proof_len, uint32_t flags, char* err_buf, size_t err_cap);

/* Length of the postcard-encoded `BatchProof` at the head of `bytes` (what p3r_prove_* return; the
 * serialised `BatchStarkProof` continues with its metadata, batch_stark_prover.rs:610-636). */
int p3r_batch_proof_len(uint32_t field, const uint8_t* bytes, size_t len, uint32_t flags, size_t* proof_len,
                        char* err_buf, size_t err_cap);
/* Same for proofs written with a non-default p3r_config.proof_layout (18 bytes, or NULL). */
int p3r_batch_proof_len_layout(uint32_t field, const uint8_t* bytes, size_t len, uint32_t flags,
                               const uint8_t* proof_layout, size_t* proof_len, char* err_buf, size_t err_cap);

/* The whole serialised `BatchStarkProof` (what `to_postcard` / the Rust `postcard::to_allocvec(&proof)` emits):
 * the inner `BatchProof` followed by the metadata fields of batch_stark_prover.rs:610-636 (TablePacking
 * packing.rs:9-27, RowCounts :459-460, NonPrimitiveTableEntry :272-290, SerializedStarkCommon :505-511).
 * p3r_batch_stark_proof_parse walks the bytes once without building containers (framing, every field
 * element in range), decodes the metadata into `out` (field elements canonical) and applies the structural
 * rules a `#[derive(Deserialize)]` bypasses (batch_stark_prover.rs:666-681, packing.rs:140-161).  It is what
 * a parent node of the aggregation tree runs on each child before proving (recursion.rs:656-762 receives the
 * children already deserialised): host code, no device, no p3r_ctx. */
#define P3R_META_MAX_NPO 8
#define P3R_META_MAX_INSTANCES 16
#define P3R_META_MAX_CAP 64
typedef struct p3r_npo_table_entry {
  char op_type[64];        /* NpoTypeId(String), NUL-terminated */
  uint64_t rows;
  uint32_t lanes;
  uint32_t air_variant;    /* 0 Baseline, 1 Optimized */
  uint32_t n_public_values;
  uint32_t public_values[8];
} p3r_npo_table_entry;
typedef struct p3r_batch_stark_meta {
  uint64_t proof_len;      /* length of the inner BatchProof at the head of the bytes */
  uint64_t parse_ns;       /* time this call spent (steady clock): what a parent node pays per child */
  uint32_t public_lanes, alu_lanes, min_trace_height, horner_packed_steps;
  uint32_t n_npo_lanes;    /* TablePacking.npo_lanes: Vec<(NpoTypeId, usize)> */
  struct { char op_type[64]; uint32_t lanes; } npo_lanes[P3R_META_MAX_NPO];
  uint64_t rows[3];        /* RowCounts([const, public, alu]) */
  uint32_t alu_variant, ext_degree;
  uint32_t has_w_binomial, w_binomial, alu_quintic_trinomial;
  uint32_t n_non_primitives;
  p3r_npo_table_entry non_primitives[P3R_META_MAX_NPO];
  uint32_t has_stark_common;
  uint32_t cap_len;        /* digests of the preprocessed commitment */
  uint32_t commitment[8 * P3R_META_MAX_CAP];
  uint32_t n_instances;    /* SerializedStarkCommon.instances (Some entries, in order) */
  uint32_t preprocessed_widths[P3R_META_MAX_INSTANCES];
  uint32_t degree_bits[P3R_META_MAX_INSTANCES];
} p3r_batch_stark_meta;
int p3r_batch_stark_proof_parse(uint32_t field, const uint8_t* bytes, size_t len, uint32_t flags,
                                const uint8_t* proof_layout, p3r_batch_stark_meta* out, char* err_buf,
                                size_t err_cap);

/* ---- the caller side of prove_next_layer: the circuit itself ------------------------------------
 * `prove_next_layer` (recursion/src/recursion.rs:401-502) receives a `Circuit<EF>`, sets its public
 * inputs and the Merkle-sibling private data, RUNS it (`CircuitRunner::run`,
 * circuit/src/tables/runner.rs:195-253) and proves the resulting `Traces`.  The entry points below
 * take the flattened `Circuit<EF>` (circuit/src/circuit.rs:152-181) instead of pre-computed
 * `Traces` + preprocessed columns:
 *   p3r_circuit_create  == Circuit::generate_preprocessed_columns::<4> (circuit.rs:237-510)
 *                          + get_airs_and_degrees_with_prep (circuit-prover/src/common.rs:127-390)
 *                          + poseidon_preprocess_for_prover (batch_stark_prover.rs:97-246)
 *                          + recompose_preprocess_for_op (batch_stark_prover/recompose.rs:294-358)
 *                          + ProverData::from_airs_and_degrees            (build_next_layer_prep)
 *   p3r_circuit_run     == set_public_inputs / set_private_inputs / set_private_data + run()
 *                          (runner.rs:83-253) with the witness table and the Traces kept in HBM
 *   p3r_prove_next_layer== run + prove_all_tables
 * D = 4 (extension-field witnesses), Poseidon2 D4 width 16, Recompose without coefficient lookups:
 * the tables FriRecursionBackend registers (recursion/src/backend/fri.rs:693-721) - and, since ABI version 8, the
 * width-32 Poseidon2 table of a mixed-config verifier circuit (P3R_OP_POSEIDON2_W32_PERM). */
#define P3R_NO_WITNESS 0xFFFFFFFFu

enum p3r_op_kind {               /* circuit/src/ops/op.rs `Op`, AluOpKind */
  P3R_OP_CONST = 0,              /* out; ext[ext_off..+4] = value coefficients (canonical) */
  P3R_OP_PUBLIC = 1,             /* out; aux = public_pos */
  P3R_OP_ALU_ADD = 2,            /* a, b, out                    (c = aux = P3R_NO_WITNESS) */
  P3R_OP_ALU_MUL = 3,
  P3R_OP_ALU_BOOL_CHECK = 4,     /* a, b, out */
  P3R_OP_ALU_MUL_ADD = 5,        /* a, b, c or NO_WITNESS, out; aux = intermediate_out or NO_WITNESS */
  P3R_OP_ALU_HORNER_ACC = 6,     /* a, b, c, out; aux = acc (intermediate_out) */
  P3R_OP_HINT_EXT_DECOMPOSITION = 7,    /* a = input; ext = 4 output witnesses
                                           (circuit/src/builder/circuit_builder.rs:1659-1728) */
  P3R_OP_HINT_BINARY_DECOMPOSITION = 8, /* a = input; ext = ext_len output witnesses (:1750-1810) */
  P3R_OP_POSEIDON2_PERM = 9,     /* a = NonPrimitiveOpId; aux = flags (bit 0 new_start, bit 1 merkle_path);
                                    ext = [in0..in3, mmcs_index_sum, mmcs_bit, n_out (2 or 4), out0..];
                                    empty slots are P3R_NO_WITNESS (poseidon_perm/executor.rs:921-972).
                                    Under ext_degree 1 / 5 the op is a base-mode permutation (KOALA_BEAR_D1_W16 /
                                    BABY_BEAR_D1_W16, one witness per state element, executor.rs:600-700):
                                    ext = [in0..in15, mmcs_index_sum, mmcs_bit, n_out (8 or 16), out0..],
                                    b = absorb_len (the sponge length tag) */
  P3R_OP_RECOMPOSE = 10,         /* a = NonPrimitiveOpId; out; ext = D coefficient witnesses
                                    (circuit/src/ops/recompose.rs:115-170); aux = 0 (or P3R_NO_WITNESS): the `recompose` table,
                                    aux = 1: `recompose/coeff` (NpoTypeId::recompose_with_coeff_lookups,
                                    ops/npo.rs:48-60: every coefficient is a bus tuple too; a coefficient that is a
                                    hint output is created by this row with its read count, any other is named with
                                    multiplicity 0 - batch_stark_prover/recompose.rs:341-352).  A circuit may hold both kinds:
                                    the layer then proves two Recompose tables, `recompose` before `recompose/coeff`
                                    (p3r_layer_desc_counts.n_recompose_coeff). */
  P3R_OP_POSEIDON2_W32_PERM = 11 /* ABI version 8.  A permutation of the width-32 table (`poseidon2_perm/<field>_d4_w32`, the arity-4
                                    compression shape 4 * CAPACITY_EXT == WIDTH_EXT: eight input limbs, rate six, digests of two
                                    limbs) in a D = 4 circuit - the MMCS rows of a verifier circuit built under `--arity4`
                                    (recursion/examples/recursive_aggregation.rs:902-1046; W16 challenger rows stay
                                    P3R_OP_POSEIDON2_PERM).  a = NonPrimitiveOpId; aux = flags (bit 0 new_start, bit 1
                                    merkle_path); ext = [in0..in7, mmcs_index_sum (must be P3R_NO_WITNESS: the arity-4 table has
                                    no index accumulator bus), mmcs_bit, mmcs_bit2, n_out (6 or 8), out0..].  Semantics of
                                    PoseidonPermExecutor::execute for `is_arity4()` (circuit/src/ops/poseidon_perm/
                                    executor.rs:92-235,290-303,493-561,947-966): a sponge row chains the whole previous normal
                                    output and mirrors its own output into the Merkle chain state; a Merkle row places the
                                    previous Merkle-state digest (output limbs 0, 1) into chunk pos = mmcs_bit + 2 * mmcs_bit2,
                                    fills the other three chunks from its private data in ascending order, then overwrites
                                    every limb that names a witness.  Both direction bits are required on a Merkle row.  The
                                    preprocessed rows are Poseidon2PreprocessedRow<8, 6> (executor.rs:777-884: every named input
                                    limb is a bus read, Merkle rows included; the accumulator slots carry the two bit
                                    witnesses, both read). */
};

typedef struct p3r_op {
  uint32_t kind;
  uint32_t a, b, c, out, aux;
  uint32_t ext_off, ext_len; /* slice of p3r_circuit_desc.ext */
} p3r_op;

typedef struct p3r_circuit_desc {
  uint32_t witness_count;
  size_t n_ops;     const p3r_op* ops;                     /* execution order */
  size_t n_ext;     const uint32_t* ext;
  size_t n_public;  const uint32_t* public_rows;           /* witness of public input i */
  size_t n_private; const uint32_t* private_input_rows;
  size_t n_rewrite; const uint32_t* witness_rewrite;       /* pairs (duplicate, canonical) */
  uint32_t public_lanes, alu_lanes, horner_packed_steps, recompose_lanes, min_trace_height;
} p3r_circuit_desc;

typedef struct p3r_circuit_inputs {
  const uint32_t* public_values;   /* n_public x 4, canonical */
  const uint32_t* private_values;  /* n_private x 4 */
  size_t n_private_data;                 /* set_private_data: Poseidon2PermPrivateData { sibling } */
  const uint32_t* private_data_op_ids;   /* n_private_data NonPrimitiveOpIds */
  const uint32_t* private_data_siblings; /* n_private_data x 8: two extension limbs */
  /* ABI version 8.  Private data of P3R_OP_POSEIDON2_W32_PERM Merkle rows: the three sibling digests of an arity-4
   * level, chunk by chunk in ascending chunk order with the running-hash chunk skipped (fill_sibling_data,
   * executor.rs:166-201).  An op id of the other width in either list is an error. */
  size_t n_private_data_w32;
  const uint32_t* private_data_w32_op_ids;   /* n_private_data_w32 NonPrimitiveOpIds */
  const uint32_t* private_data_w32_siblings; /* n_private_data_w32 x 24: three chunks of two extension limbs */
} p3r_circuit_inputs;

typedef struct p3r_circuit p3r_circuit;

/* Fails (NULL + p3r_last_error) on a malformed circuit, an unclaimed private input
 * (circuit.rs:497-503) or an op the three-table backend has no table for. */
p3r_circuit* p3r_circuit_create(p3r_ctx* ctx, const p3r_circuit_desc* desc, uint32_t* commit_out);
void p3r_circuit_free(p3r_ctx* ctx, p3r_circuit* circuit);
/* The CircuitProverData the circuit was prepared into (owned by the circuit). */
const p3r_layer* p3r_circuit_layer(const p3r_circuit* circuit);
/* Op counts of the tables (five, or six with a second Recompose table), i.e. the sizes of the arrays p3r_dtraces_get returns. */
int p3r_circuit_counts(const p3r_circuit* circuit, p3r_layer_desc_counts* out);
/* Levels of the execution schedule (ops of one level have no dependencies on each other). */
int p3r_circuit_levels(const p3r_circuit* circuit, size_t* n_levels);
/* 1: the circuit was prepared on the device (preprocessed columns, ALU lane schedule and execution schedule built
 * in HBM from the uploaded op list); 0: by the host restatement of the same steps (P3R_PREP_HOST=1, or a circuit
 * the device pass handed over).  Both give the same commitment, schedule and proofs. */
int p3r_circuit_prepared_on_device(const p3r_circuit* circuit);

/* CircuitRunner::run on the device.  Errors mirror CircuitError: a witness conflict
 * (runner.rs:473-510), DivisionByZero (:378), a non-boolean mmcs_bit
 * (poseidon_perm/executor.rs:305-335), a witness that is never set (:218-221). */
p3r_dtraces* p3r_circuit_run(p3r_ctx* ctx, const p3r_circuit* circuit, const p3r_circuit_inputs* inputs);
/* run + prove_all_tables in one call.  A CircuitError of the run is what the caller gets (same code
 * and text as p3r_circuit_run), never an error the prover derived from the failed run's traces. */
int p3r_prove_next_layer(p3r_ctx* ctx, const p3r_circuit* circuit, const p3r_circuit_inputs* inputs,
                         uint32_t flags, uint8_t* proof_buf, size_t proof_cap, size_t* proof_len);

/* Inputs made resident in HBM once (set_public_inputs / set_private_inputs / set_private_data with
 * their checks, runner.rs:83-176); a run or a prove can then be repeated without PCIe. */
typedef struct p3r_dinputs p3r_dinputs;
p3r_dinputs* p3r_circuit_inputs_upload(p3r_ctx* ctx, const p3r_circuit* circuit, const p3r_circuit_inputs* inputs);
void p3r_circuit_inputs_free(p3r_ctx* ctx, p3r_dinputs* inputs);
p3r_dtraces* p3r_circuit_run_resident(p3r_ctx* ctx, const p3r_circuit* circuit, const p3r_dinputs* inputs);
int p3r_prove_next_layer_resident(p3r_ctx* ctx, const p3r_circuit* circuit, const p3r_dinputs* inputs,
                                  uint32_t flags, uint8_t* proof_buf, size_t proof_cap, size_t* proof_len);

/* Read back one array of device-resident Traces (canonical), for parity tests / inspection. */
enum p3r_traces_array {
  P3R_TRACES_CONST_VALUES = 0,    /* n_const x 4 */
  P3R_TRACES_PUBLIC_VALUES = 1,   /* n_public x 4 */
  P3R_TRACES_ALU_VALUES = 2,      /* n_alu x 16 */
  P3R_TRACES_P2_INPUT_VALUES = 3, /* n_p2 x 16 */
  P3R_TRACES_P2_FLAGS = 4,        /* n_p2 x 3: new_start, merkle_path, mmcs_bit */
  P3R_TRACES_P2_MMCS_INDEX_SUM = 5, /* n_p2 */
  P3R_TRACES_RECOMPOSE_VALUES = 6, /* n_recompose x D */
  P3R_TRACES_RECOMPOSE_COEFF_VALUES = 7, /* n_recompose_coeff x D: the second Recompose table */
  P3R_TRACES_P2W_INPUT_VALUES = 8,   /* n_p2w x 32 (ABI 8: rows of the width-32 Poseidon2 table) */
  P3R_TRACES_P2W_FLAGS = 9,          /* n_p2w x 4: new_start, merkle_path, mmcs_bit, mmcs_bit2 */
  P3R_TRACES_P2W_MMCS_INDEX_SUM = 10 /* n_p2w */
};
int p3r_dtraces_get(p3r_ctx* ctx, const p3r_layer* layer, const p3r_dtraces* traces, uint32_t which,
                    uint32_t* out, size_t out_len);

/* ---- measurement support (bench.py): run `iters` back-to-back launches of one kernel
 * family on resident data and return the mean per-launch time measured with HIP events
 * on the ctx's own stream. ---- */
int p3r_time_permute_dmat(p3r_ctx* ctx, p3r_dmat* states, int iters, double* ms_per_launch);

/* Per-kernel-family timers: while enabled, every launch of a hot kernel is bracketed by HIP
 * events on the ctx's stream; p3r_profile_read sums them per family since the last enable. */
typedef struct p3r_profile_entry {
  char name[32];
  double total_ms;
  uint64_t launches;
} p3r_profile_entry;
int p3r_profile_enable(p3r_ctx* ctx, int on);
int p3r_profile_read(p3r_ctx* ctx, p3r_profile_entry* out, size_t cap, size_t* n_out);

#ifdef __cplusplus
}
#endif
#endif /* P3R_H */
