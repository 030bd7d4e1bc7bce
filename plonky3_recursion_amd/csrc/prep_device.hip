// Device-side preparation of a verifier circuit (see prep_device.h).  Second translation unit of
// libp3r_hip.so: kernels + their sequencing; p3r_core.hip turns the result into a p3r_circuit.
//
// Everything the sequential host restatement decides while walking the ops in order - which op DEFINES a
// witness on the WitnessChecks bus and which ones read it (circuit.rs:237-510), which op SETS a witness at
// run time and which ones only compare against it (runner.rs:473-510), at which level of the dependency
// graph an op can run - is a function of "the first op that touches a witness in a given role".  So the
// walk becomes: atomicMin of the op index per witness (plus a short fixed-point loop for the two roles that
// depend on another witness's state), then maps over the ops.  Ranks inside a table, offsets into the
// packed arrays and the level-sorted order of the schedule are scans and one stable radix sort (device_prims.hip.h).
#include "prep_device.h"
#include "profile.h"

#include "device_prims.hip.h"

#include <algorithm>
#include <unordered_map>

namespace p3r {
namespace {

constexpr int kB = 256;
constexpr uint32_t kUnset = 0xFFFFFFFFu;
inline unsigned nblk(size_t n) { return (unsigned)((n + kB - 1) / kB); }

// The circuit's extension degree D and the layout of its Poseidon2 ops (circuit_impl.hip.h::P2Shape): D = 4 -> four
// input limbs, two CTL-exposed output limbs; otherwise base mode -> sixteen one-element slots, eight exposed outputs.
// ext = [in[il].., mmcs_index_sum, mmcs_bit, n_out, out..].
struct Shape {
  uint32_t D, il, ol, ol_full;
};
inline Shape shape_of(uint32_t D) { return D == 4 ? Shape{4, 4, 2, 4} : Shape{D, 16, 8, 16}; }

// witness flag bits
enum : uint32_t { WF_PRIVATE = 1, WF_CP = 2, WF_HINT = 4, WF_DUP_P2 = 8, WF_DUP_REC = 16, WF_DUP_REC_COEFF = 32, WF_DUP_P2W = 64 };
// static per-op flags (beyond RUN_*): bits 12..
enum : uint32_t { OF_READY = 1u << 12, OF_LINK = 1u << 13, OF_LIGHT = 1u << 14 };
// why the device pass gives up (any bit set: the host path reports)
enum : uint32_t { BAD_OP = 1, BAD_ROW = 2, BAD_UNCLAIMED = 4, BAD_DEFERRED = 8, BAD_DUP_ID = 16 };

struct Op {
  uint32_t kind, a, b, c, out, aux, ext_off, ext_len;
};
__device__ __forceinline__ Op load_op(const uint32_t* __restrict__ ops, size_t i) {
  const uint4 x = reinterpret_cast<const uint4*>(ops)[2 * i], y = reinterpret_cast<const uint4*>(ops)[2 * i + 1];
  return Op{x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
}
__device__ __forceinline__ bool is_alu(uint32_t k) { return k >= P3R_OP_ALU_ADD && k <= P3R_OP_ALU_HORNER_ACC; }
__device__ __forceinline__ bool is_hint(uint32_t k) {
  return k == P3R_OP_HINT_EXT_DECOMPOSITION || k == P3R_OP_HINT_BINARY_DECOMPOSITION;
}

// ------------------------------------------------------------------------------------------------ validation
// validate_circuit (circuit_impl.hip.h) + the canonical check of constants, as a yes / no per op
template <class PP>
__global__ void __launch_bounds__(kB) k_validate(const uint32_t* __restrict__ ops, size_t n_ops, const uint32_t* __restrict__ ext,
                                                 size_t n_ext, uint32_t nw, Shape sh, uint32_t* __restrict__ bad) {
  const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
  if (i >= n_ops) return;
  const Op op = load_op(ops, i);
  bool ok = (size_t)op.ext_off + op.ext_len <= n_ext;
  const uint32_t* e = ext + op.ext_off;
  auto wid = [&](uint32_t w) { return w < nw; };
  auto opt = [&](uint32_t w) { return w == kNoW || w < nw; };
  if (ok) switch (op.kind) {
    case P3R_OP_CONST:
      ok = wid(op.out) && op.ext_len == sh.D;
      if (ok) for (uint32_t k = 0; k < sh.D; ++k) ok = ok && e[k] < PP::P;
      break;
    case P3R_OP_PUBLIC: ok = wid(op.out); break;
    case P3R_OP_ALU_ADD: case P3R_OP_ALU_MUL: case P3R_OP_ALU_BOOL_CHECK: case P3R_OP_ALU_MUL_ADD: case P3R_OP_ALU_HORNER_ACC:
      ok = wid(op.a) && wid(op.b) && wid(op.out) && opt(op.c) && opt(op.aux);
      if (op.kind == P3R_OP_ALU_HORNER_ACC && (op.c == kNoW || op.aux == kNoW)) ok = false;
      break;
    case P3R_OP_HINT_EXT_DECOMPOSITION:
      ok = wid(op.a) && op.ext_len == sh.D;
      if (ok) for (uint32_t k = 0; k < sh.D; ++k) ok = ok && wid(e[k]);
      break;
    case P3R_OP_HINT_BINARY_DECOMPOSITION:
      ok = wid(op.a) && op.ext_len <= 31 * sh.D;
      if (ok) for (uint32_t k = 0; k < op.ext_len; ++k) ok = ok && wid(e[k]);
      break;
    case P3R_OP_POSEIDON2_PERM: {
      const uint32_t hdr = sh.il + 3;
      ok = op.ext_len >= hdr && (e[hdr - 1] == sh.ol || e[hdr - 1] == sh.ol_full) && op.ext_len == hdr + e[hdr - 1];
      if (ok) {
        for (uint32_t k = 0; k < sh.il + 2; ++k) ok = ok && opt(e[k]);
        for (uint32_t k = 0; k < e[hdr - 1]; ++k) ok = ok && opt(e[hdr + k]);
        if ((op.aux & 2) && e[sh.il + 1] == kNoW) ok = false;
        if (sh.D != 4 && !(op.aux & 2))   // compact D = 1 sponge rows: the capacity slots stay empty (executor.rs:712-725)
          for (uint32_t k = sh.ol; k < sh.il; ++k) ok = ok && e[k] == kNoW;
        if (sh.D != 4 && op.b > 255) ok = false;   // absorb_len is the length tag
        if (op.a >= n_ops) ok = false;
      }
      break;
    }
    case P3R_OP_POSEIDON2_W32_PERM:
      // the arity-4 shape (circuit_host.h::validate_circuit): D = 4 circuits only, no index accumulator, both direction
      // bits on a Merkle row and none on a sponge row
      ok = sh.D == 4 && op.ext_len >= kW32Hdr && (e[kW32NOutSlot] == kW32Rate || e[kW32NOutSlot] == kW32In) &&
           op.ext_len == kW32Hdr + e[kW32NOutSlot];
      if (ok) {
        for (uint32_t k = 0; k < kW32NOutSlot; ++k) ok = ok && opt(e[k]);
        for (uint32_t k = 0; k < e[kW32NOutSlot]; ++k) ok = ok && opt(e[kW32Hdr + k]);
        if (e[kW32IdxSlot] != kNoW) ok = false;
        const bool b1 = e[kW32BitSlot] != kNoW, b2 = e[kW32Bit2Slot] != kNoW;
        if ((op.aux & 2) ? !(b1 && b2) : (b1 || b2)) ok = false;
        if (op.a >= n_ops) ok = false;
      }
      break;
    case P3R_OP_RECOMPOSE:
      // aux: 0 / P3R_NO_WITNESS = `recompose`, 1 = `recompose/coeff`; anything else the host path rejects
      ok = wid(op.out) && op.a < n_ops && op.ext_len == sh.D && (op.aux == 0u || op.aux == 1u || op.aux == kNoW);
      if (ok) for (uint32_t k = 0; k < sh.D; ++k) ok = ok && wid(e[k]);
      break;
    default: ok = false;
  }
  if (!ok) atomicOr(bad, BAD_OP);
}

// public / private rows: inputs are SET before the run (runner.rs:83-122); private ones are claimed on the bus later
__global__ void __launch_bounds__(kB) k_mark_rows(const uint32_t* __restrict__ rows, size_t n, uint32_t nw, uint32_t flag,
                                                  uint32_t* __restrict__ wflags, uint32_t* __restrict__ stime,
                                                  uint32_t* __restrict__ bad) {
  const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  const uint32_t w = rows[i];
  if (w >= nw) { atomicOr(bad, BAD_ROW); return; }
  if (flag) atomicOr(&wflags[w], flag);
  stime[w] = 0;
}

// ------------------------------------------------------------------------------------------------ ranks
// Exclusive counts of the ops before op i, per class, packed two to a 64-bit sum.
struct PairConstPublic {
  const uint32_t* ops;
  __device__ uint64_t operator()(size_t i) const {
    const uint32_t k = ops[8 * i];
    return (uint64_t)(k == P3R_OP_CONST) | ((uint64_t)(k == P3R_OP_PUBLIC) << 32);
  }
};
struct PairAluP2 {
  const uint32_t* ops;
  __device__ uint64_t operator()(size_t i) const {
    const uint32_t k = ops[8 * i];
    return (uint64_t)is_alu(k) | ((uint64_t)(k == P3R_OP_POSEIDON2_PERM) << 32);
  }
};
struct PairRecExt {  // plain recompose rows | cells this op appends to the device ext array
  const uint32_t* ops;
  uint32_t D;
  __device__ uint64_t operator()(size_t i) const {
    const uint32_t k = ops[8 * i];
    const uint32_t cells = (k == P3R_OP_CONST || k == P3R_OP_RECOMPOSE) ? D : is_hint(k) ? ops[8 * i + 7] : 0u;
    return (uint64_t)(k == P3R_OP_RECOMPOSE && ops[8 * i + 5] != 1u) | ((uint64_t)cells << 32);
  }
};
struct FlagRecCoeff {  // rows of the `recompose/coeff` kind (aux = 1)
  const uint32_t* ops;
  __device__ uint32_t operator()(size_t i) const { return (ops[8 * i] == P3R_OP_RECOMPOSE && ops[8 * i + 5] == 1u) ? 1u : 0u; }
};

struct FlagP2W {  // rows of the width-32 Poseidon2 table
  const uint32_t* ops;
  __device__ uint32_t operator()(size_t i) const { return ops[8 * i] == P3R_OP_POSEIDON2_W32_PERM ? 1u : 0u; }
};

template <class Fn>
void scan_pairs(p3r_ctx* ctx, Fn fn, size_t n, DevBuf& out /* n + 1 u64: exclusive sums, total at [n] */) {
  out.alloc(2 * (n + 1));
  prims::exclusive_sum<uint64_t>(ctx->stream, fn, n, reinterpret_cast<uint64_t*>(out.p));   // the total: k_scan_total
}
// total = excl[n-1] + fn(n-1)
template <class Fn>
__global__ void k_scan_total(Fn fn, size_t n, uint64_t* __restrict__ excl) {
  if (threadIdx.x == 0 && blockIdx.x == 0) excl[n] = n ? excl[n - 1] + fn(n - 1) : 0;
}

void scan_u32(p3r_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n) {
  prims::exclusive_sum<uint32_t>(ctx->stream, prims::LoadU32{in}, n, out);
}

// op -> its row in its table; the tables' op lists
__global__ void __launch_bounds__(kB) k_table_lists(const uint32_t* __restrict__ ops, size_t n_ops, const uint64_t* __restrict__ s_cp,
                                                    const uint64_t* __restrict__ s_ap, const uint64_t* __restrict__ s_re,
                                                    uint32_t* __restrict__ const_ops, uint32_t* __restrict__ public_ops,
                                                    uint32_t* __restrict__ alu_ops, uint32_t* __restrict__ p2_ops,
                                                    uint32_t* __restrict__ rec_ops, const uint32_t* __restrict__ s_rc,
                                                    uint32_t* __restrict__ rec_coeff_ops) {
  const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
  if (i >= n_ops) return;
  const uint32_t k = ops[8 * i];
  if (k == P3R_OP_CONST) const_ops[(uint32_t)s_cp[i]] = (uint32_t)i;
  else if (k == P3R_OP_PUBLIC) public_ops[(uint32_t)(s_cp[i] >> 32)] = (uint32_t)i;
  else if (is_alu(k)) alu_ops[(uint32_t)s_ap[i]] = (uint32_t)i;
  else if (k == P3R_OP_POSEIDON2_PERM) p2_ops[(uint32_t)(s_ap[i] >> 32)] = (uint32_t)i;
  else if (k == P3R_OP_RECOMPOSE) {
    if (ops[8 * i + 5] == 1u) rec_coeff_ops[s_rc[i]] = (uint32_t)i;
    else rec_ops[(uint32_t)s_re[i]] = (uint32_t)i;
  }
}

// ------------------------------------------------------------------------------------------------ first touches
__global__ void __launch_bounds__(kB) k_mark_cp(const uint32_t* __restrict__ ops, size_t n_ops, uint32_t* __restrict__ wflags) {
  const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
  if (i >= n_ops) return;
  const uint32_t k = ops[8 * i];
  if (k == P3R_OP_CONST || k == P3R_OP_PUBLIC) atomicOr(&wflags[ops[8 * i + 4]], WF_CP);
}
// hint outputs not also produced by a Const / Public op (circuit.rs:263-284)
__global__ void __launch_bounds__(kB) k_mark_hint(const uint32_t* __restrict__ ops, size_t n_ops, const uint32_t* __restrict__ ext,
                                                  uint32_t* __restrict__ wflags) {
  const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
  if (i >= n_ops) return;
  const Op op = load_op(ops, i);
  if (!is_hint(op.kind)) return;
  for (uint32_t k = 0; k < op.ext_len; ++k) {
    const uint32_t w = ext[op.ext_off + k];
    if (!(wflags[w] & WF_CP)) atomicOr(&wflags[w], WF_HINT);
  }
}

// tdef[w]: time (op index + 1) of the op that DEFINES w on the bus; stime[w]: time of the op that SETS w at run time.
// This pass takes every role that defines / sets unconditionally; k_times_fix adds the two conditional roles.
__global__ void __launch_bounds__(kB) k_times(const uint32_t* __restrict__ ops, size_t n_ops, const uint32_t* __restrict__ ext,
                                              const uint32_t* __restrict__ wflags, Shape sh, uint32_t* __restrict__ tdef,
                                              uint32_t* __restrict__ stime) {
  const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
  if (i >= n_ops) return;
  const Op op = load_op(ops, i);
  const uint32_t t = (uint32_t)i + 1;
  const uint32_t* e = ext + op.ext_off;
  switch (op.kind) {
    case P3R_OP_CONST: atomicMin(&tdef[op.out], t); atomicMin(&stime[op.out], t); break;
    case P3R_OP_PUBLIC: atomicMin(&tdef[op.out], t); break;
    case P3R_OP_POSEIDON2_PERM:
      for (uint32_t l = 0; l < e[sh.il + 2]; ++l) {
        const uint32_t w = e[sh.il + 3 + l];
        if (w == kNoW) continue;
        if (l < sh.ol) atomicMin(&tdef[w], t);   // the CTL-exposed outputs (circuit.rs:464-491)
        atomicMin(&stime[w], t);
      }
      break;
    case P3R_OP_POSEIDON2_W32_PERM:
      for (uint32_t l = 0; l < e[kW32NOutSlot]; ++l) {
        const uint32_t w = e[kW32Hdr + l];
        if (w == kNoW) continue;
        if (l < kW32Rate) atomicMin(&tdef[w], t);   // the six rate limbs are on the bus
        atomicMin(&stime[w], t);
      }
      break;
    case P3R_OP_RECOMPOSE: atomicMin(&tdef[op.out], t); atomicMin(&stime[op.out], t); break;
    case P3R_OP_HINT_EXT_DECOMPOSITION: case P3R_OP_HINT_BINARY_DECOMPOSITION:
      for (uint32_t k = 0; k < op.ext_len; ++k) atomicMin(&stime[e[k]], t);
      break;
    default: {  // ALU
      atomicMin(&tdef[op.out], t);  // defined by now: earlier, or created here
      const uint32_t pa = wflags[op.a] & (WF_PRIVATE | WF_HINT);
      if (pa && op.a != op.out) atomicMin(&tdef[op.a], t);
      if (op.c != kNoW && (wflags[op.c] & (WF_PRIVATE | WF_HINT)) && op.c != op.out) atomicMin(&tdef[op.c], t);
      if ((wflags[op.b] & WF_PRIVATE) || (wflags[op.out] & WF_HINT)) atomicMin(&tdef[op.b], t);
      // run time: Add / Mul write `out`, or solve for `b` when it is not set yet (runner.rs:341-385): either way
      // b is set once this op has run; `out` only when b was set before (k_times_fix)
      if (op.kind == P3R_OP_ALU_ADD || op.kind == P3R_OP_ALU_MUL) atomicMin(&stime[op.b], t);
      else {
        atomicMin(&stime[op.out], t);
        if (op.kind == P3R_OP_ALU_MUL_ADD && op.aux != kNoW) atomicMin(&stime[op.aux], t);
      }
    }
  }
}
// conditional roles: b is created on the bus when `out` was defined earlier (circuit.rs:360-379);
// an Add / Mul sets `out` when b was set earlier.  Both only ever LOWER a time, and a lower time only ever
// enables more of them: iterating to the fixed point gives the sequential walk's answer.
__global__ void __launch_bounds__(kB) k_times_fix(const uint32_t* __restrict__ ops, size_t n_ops, const uint32_t* __restrict__ wflags,
                                                  uint32_t* __restrict__ tdef, uint32_t* __restrict__ stime,
                                                  uint32_t* __restrict__ changed) {
  const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
  if (i >= n_ops) return;
  const uint32_t k = ops[8 * i];
  if (!is_alu(k)) return;
  const uint32_t b = ops[8 * i + 2], out = ops[8 * i + 4], t = (uint32_t)i + 1;
  bool ch = false;
  if (!(wflags[b] & WF_PRIVATE) && !(wflags[out] & WF_HINT) && tdef[out] < t && tdef[b] > t) ch |= atomicMin(&tdef[b], t) > t;
  if ((k == P3R_OP_ALU_ADD || k == P3R_OP_ALU_MUL) && stime[b] < t && stime[out] > t) ch |= atomicMin(&stime[out], t) > t;
  if (ch) *changed = 1;
}

// ------------------------------------------------------------------------------------------------ bus roles + read counts
// generate_preprocessed_columns pass 1 (circuit_impl.hip.h::circuit_tables): who reads what
__global__ void __launch_bounds__(kB) k_roles(const uint32_t* __restrict__ ops, size_t n_ops, const uint32_t* __restrict__ ext,
                                              uint32_t* __restrict__ wflags, const uint32_t* __restrict__ tdef, Shape sh,
                                              const uint64_t* __restrict__ s_ap, uint32_t* __restrict__ reads,
                                              uint32_t* __restrict__ roles /* per ALU row: a_state | c_state<<8 | b_creator<<16 | out_creator<<24 */) {
  const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
  if (i >= n_ops) return;
  const Op op = load_op(ops, i);
  const uint32_t t = (uint32_t)i + 1;
  const uint32_t* e = ext + op.ext_off;
  auto def = [&](uint32_t w) { return tdef[w] < t; };
  switch (op.kind) {
    case P3R_OP_CONST: case P3R_OP_PUBLIC: case P3R_OP_HINT_EXT_DECOMPOSITION: case P3R_OP_HINT_BINARY_DECOMPOSITION: break;
    case P3R_OP_POSEIDON2_PERM: {
      const bool merkle = op.aux & 2;
      for (uint32_t l = 0; l < sh.il; ++l)
        if (e[l] != kNoW && !merkle) atomicAdd(&reads[e[l]], 1u);  // Merkle rows name the limb without a bus read
      for (uint32_t l = 0; l < sh.ol; ++l) {
        const uint32_t w = e[sh.il + 3 + l];
        if (w == kNoW) continue;
        bool earlier = false;   // an earlier output of this same row already defined it
        for (uint32_t j = 0; j < l; ++j) earlier |= e[sh.il + 3 + j] == w;
        if (def(w) || earlier) { atomicOr(&wflags[w], WF_DUP_P2); atomicAdd(&reads[w], 1u); }
      }
      break;
    }
    case P3R_OP_POSEIDON2_W32_PERM: {
      // every named input limb is a bus read - Merkle rows too - and a Merkle row reads its two direction bits
      // (circuit_host.h::circuit_tables; executor.rs:777-793,880-893)
      for (uint32_t l = 0; l < kW32In; ++l)
        if (e[l] != kNoW) atomicAdd(&reads[e[l]], 1u);
      for (uint32_t l = 0; l < kW32Rate; ++l) {
        const uint32_t w = e[kW32Hdr + l];
        if (w == kNoW) continue;
        bool earlier = false;
        for (uint32_t j = 0; j < l; ++j) earlier |= e[kW32Hdr + j] == w;
        if (def(w) || earlier) { atomicOr(&wflags[w], WF_DUP_P2W); atomicAdd(&reads[w], 1u); }
      }
      if (op.aux & 2) { atomicAdd(&reads[e[kW32BitSlot]], 1u); atomicAdd(&reads[e[kW32Bit2Slot]], 1u); }
      break;
    }
    case P3R_OP_RECOMPOSE:
      // dup_npo_outputs is kept per op type: one flag per Recompose kind
      if (def(op.out)) { atomicOr(&wflags[op.out], op.aux == 1u ? WF_DUP_REC_COEFF : WF_DUP_REC); atomicAdd(&reads[op.out], 1u); }
      break;
    default: {
      const bool out_def = def(op.out), b_def = def(op.b);
      auto state_of = [&](uint32_t w) -> uint32_t {
        if (def(w)) return 1;
        return ((wflags[w] & (WF_PRIVATE | WF_HINT)) && !(!out_def && w == op.out)) ? 2 : 0;
      };
      const uint32_t a_state = state_of(op.a), c_state = op.c != kNoW ? state_of(op.c) : 0;
      const bool out_backward = out_def || (wflags[op.out] & WF_HINT);
      const uint32_t out_creator = !out_def;
      const uint32_t b_creator = (!b_def && (wflags[op.b] & WF_PRIVATE)) || (out_backward && !b_def);
      if (!b_creator) atomicAdd(&reads[op.b], 1u);
      if (!out_creator) atomicAdd(&reads[op.out], 1u);
      if (a_state == 1) atomicAdd(&reads[op.a], 1u);
      if (c_state == 1) atomicAdd(&reads[op.c], 1u);
      roles[(uint32_t)s_ap[i]] = a_state | (c_state << 8) | (b_creator << 16) | (out_creator << 24);
    }
  }
}
// the accumulator of a Merkle chain is read when the row is followed by a chain boundary; the first padding row
// counts as one (batch_stark_prover.rs:149-176)
__global__ void __launch_bounds__(kB) k_acc_reads(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ ext,
                                                  const uint32_t* __restrict__ p2_ops, size_t n, size_t h, Shape sh,
                                                  uint32_t* __restrict__ reads) {
  const size_t r = (size_t)blockIdx.x * kB + threadIdx.x;
  if (r >= n) return;
  const Op op = load_op(ops, p2_ops[r]);
  const uint32_t acc = ext[op.ext_off + sh.il];
  if (acc == kNoW || !(op.aux & 2)) return;
  const bool next_ns = r + 1 < n ? (ops[8 * (size_t)p2_ops[r + 1] + 5] & 1) : (h > n ? true : (ops[8 * (size_t)p2_ops[0] + 5] & 1));
  if (next_ns) atomicAdd(&reads[acc], 1u);
}
__global__ void __launch_bounds__(kB) k_unclaimed(const uint32_t* __restrict__ rows, size_t n, const uint32_t* __restrict__ tdef,
                                                  uint32_t* __restrict__ bad) {
  const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
  if (i < n && tdef[rows[i]] == kUnset) atomicOr(bad, BAD_UNCLAIMED);
}

// ------------------------------------------------------------------------------------------------ preprocessed traces
template <class PP>
struct Cells {  // canonical -> the Montgomery cells the traces hold
  using F = Fp<PP>;
  static __device__ __forceinline__ uint32_t mont(uint32_t canonical) { return F::from_canonical(canonical).v; }
  static __device__ __forceinline__ uint32_t scaled(uint32_t w, uint32_t D) { return mont((uint32_t)(((uint64_t)w * D) % PP::P)); }
  static __device__ __forceinline__ uint32_t mult(const uint32_t* reads, uint32_t w) { return mont(reads[w] % PP::P); }
  static __device__ __forceinline__ uint32_t neg_mult(const uint32_t* reads, uint32_t w) {
    const uint32_t m = reads[w] % PP::P;
    return mont(m ? PP::P - m : 0);
  }
  static __device__ __forceinline__ uint32_t neg1() { return mont(PP::P - 1); }
  static __device__ __forceinline__ uint32_t one() { return F::one().v; }
};

// Const [ext_mult, D*idx] / Public [mult, idx] per lane / Recompose [D*idx, mult]: op j sits in row j / lanes, lane j % lanes
template <class PP>
__global__ void __launch_bounds__(kB) k_prep_simple(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ list, size_t n, int lanes,
                                                    int recompose, const uint32_t* __restrict__ reads, const uint32_t* __restrict__ wflags,
                                                    uint32_t D, size_t h, uint32_t* __restrict__ out) {
  const size_t j = (size_t)blockIdx.x * kB + threadIdx.x;
  if (j >= n) return;
  using C = Cells<PP>;
  const uint32_t w = ops[8 * (size_t)list[j] + 4];
  const size_t row = j / lanes, lane = j % lanes;
  uint32_t m = C::mult(reads, w);
  if (recompose && (wflags[w] & WF_DUP_REC)) m = C::neg1();
  const uint32_t idx = C::scaled(w, D);
  out[(lane * 2 + 0) * h + row] = recompose ? idx : m;
  out[(lane * 2 + 1) * h + row] = recompose ? m : idx;
}
// Recompose rows of the coefficient-lookup kind (recompose.rs:293-356): [D * out, mult, (D * coeff_k, mult_k) x D] per lane -
// a coefficient that is a hint output is created by this row with its read count, any other is named with multiplicity 0
template <class PP>
__global__ void __launch_bounds__(kB) k_prep_rec_coeff(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ ext,
                                                       const uint32_t* __restrict__ list, size_t n, int lanes,
                                                       const uint32_t* __restrict__ reads, const uint32_t* __restrict__ wflags, uint32_t D,
                                                       size_t h, uint32_t* __restrict__ out) {
  const size_t j = (size_t)blockIdx.x * kB + threadIdx.x;
  if (j >= n) return;
  using C = Cells<PP>;
  const Op op = load_op(ops, list[j]);
  const size_t row = j / lanes, lane = j % lanes, plw = 2 + 2 * (size_t)D;
  auto put = [&](size_t c, uint32_t v) { out[(lane * plw + c) * h + row] = v; };
  put(0, C::scaled(op.out, D));
  put(1, (wflags[op.out] & WF_DUP_REC_COEFF) ? C::neg1() : C::mult(reads, op.out));
  for (uint32_t k = 0; k < D; ++k) {
    const uint32_t w = ext[op.ext_off + k];
    put(2 + 2 * k, C::scaled(w, D));
    put(3 + 2 * k, (wflags[w] & WF_HINT) ? C::mult(reads, w) : 0u);
  }
}

// Poseidon2 preprocessed rows (poseidon2-circuit-air/src/air.rs:697-794, non-compact D = 4 layout) + padding (:613-649)
template <class PP>
__global__ void __launch_bounds__(kB) k_prep_p2(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ ext,
                                                const uint32_t* __restrict__ p2_ops, size_t n, const uint32_t* __restrict__ reads,
                                                const uint32_t* __restrict__ wflags, size_t h, uint32_t* __restrict__ out) {
  const size_t r = (size_t)blockIdx.x * kB + threadIdx.x;
  if (r >= h) return;
  using C = Cells<PP>;
  if (r >= n) {
    for (int c = 0; c < 24; ++c) out[(size_t)c * h + r] = (c == 22 && r == n) ? C::one() : 0u;
    return;
  }
  const Op op = load_op(ops, p2_ops[r]);
  const uint32_t* e = ext + op.ext_off;
  const bool ns = op.aux & 1, mp = op.aux & 2, en = e[4] != kNoW;
  auto put = [&](int c, uint32_t v) { out[(size_t)c * h + r] = v; };
  const uint32_t one = C::one();
  for (int l = 0; l < 4; ++l) {
    const bool ctl = e[l] != kNoW;
    put(l * 4, C::scaled(ctl ? e[l] : 0, 4));
    put(l * 4 + 1, ctl ? one : 0);
    put(l * 4 + 2, (!ns && !mp && !ctl) ? one : 0);
    put(l * 4 + 3, (!ns && mp && !ctl) ? one : 0);
  }
  for (int l = 0; l < 2; ++l) {
    const uint32_t w = e[7 + l];
    put(16 + 2 * l, C::scaled(w != kNoW ? w : 0, 4));
    put(17 + 2 * l, w == kNoW ? 0u : (wflags[w] & WF_DUP_P2) ? C::neg1() : C::mult(reads, w));
  }
  put(20, C::scaled(en ? e[4] : 0, 4));
  put(21, (en && mp) ? one : 0);
  put(22, ns ? one : 0);
  put(23, mp ? one : 0);
}
// Rows of the width-32 table, Poseidon2PreprocessedRow<8, 6> (circuit_host.h::circuit_tables; executor.rs:770-893,
// batch_stark_prover.rs:177-243): [idx, in_ctl, normal_chain_sel, merkle_chain_sel] x 8, [idx, out_ctl] x 6, the two bit
// witnesses of a Merkle row in the accumulator slots, new_start, merkle_path; padding as above
template <class PP>
__global__ void __launch_bounds__(kB) k_prep_p2w(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ ext,
                                                 const uint32_t* __restrict__ pw_ops, size_t n, const uint32_t* __restrict__ reads,
                                                 const uint32_t* __restrict__ wflags, size_t h, uint32_t* __restrict__ out) {
  const size_t r = (size_t)blockIdx.x * kB + threadIdx.x;
  if (r >= h) return;
  using C = Cells<PP>;
  if (r >= n) {
    for (int c = 0; c < kP2WPrepCols; ++c) out[(size_t)c * h + r] = (c == kP2WPrepCols - 2 && r == n) ? C::one() : 0u;
    return;
  }
  const Op op = load_op(ops, pw_ops[r]);
  const uint32_t* e = ext + op.ext_off;
  const bool ns = op.aux & 1, mp = op.aux & 2;
  auto put = [&](int c, uint32_t v) { out[(size_t)c * h + r] = v; };
  const uint32_t one = C::one();
  for (int l = 0; l < (int)kW32In; ++l) {
    const bool named = e[l] != kNoW;
    put(4 * l, C::scaled(named ? e[l] : 0, 4));
    put(4 * l + 1, named ? one : 0);
    put(4 * l + 2, (!ns && !mp && !named) ? one : 0);
    put(4 * l + 3, (!ns && mp && !named) ? one : 0);
  }
  for (int l = 0; l < (int)kW32Rate; ++l) {
    const uint32_t w = e[kW32Hdr + l];
    put(32 + 2 * l, C::scaled(w != kNoW ? w : 0, 4));
    put(33 + 2 * l, w == kNoW ? 0u : (wflags[w] & WF_DUP_P2W) ? C::neg1() : C::mult(reads, w));
  }
  put(44, mp ? C::scaled(e[kW32BitSlot], 4) : 0u);
  put(45, mp ? C::scaled(e[kW32Bit2Slot], 4) : 0u);
  put(46, ns ? one : 0);
  put(47, mp ? one : 0);
}
// the compact D = 1 rows of circuits of degree 1 / 5 (air.rs:730-763, executor.rs:720-741; layer_impl.hip.h::layer_create):
// [in_ctl x 8, length tag, cap_chain_enable, 8 + 8 chain selectors | 16 + 8 indices, 8 out_ctl | index_sum idx, 3 flags]
template <class PP>
__global__ void __launch_bounds__(kB) k_prep_p2_d1(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ ext,
                                                   const uint32_t* __restrict__ p2_ops, size_t n, const uint32_t* __restrict__ reads,
                                                   const uint32_t* __restrict__ wflags, uint32_t D, size_t h, uint32_t* __restrict__ out) {
  const size_t r = (size_t)blockIdx.x * kB + threadIdx.x;
  if (r >= h) return;
  using C = Cells<PP>;
  if (r >= n) {
    for (int c = 0; c < kP2D1PrepWidth; ++c) out[(size_t)c * h + r] = (c == kP2D1Tail + 2 && r == n) ? C::one() : 0u;
    return;
  }
  const Op op = load_op(ops, p2_ops[r]);
  const uint32_t* e = ext + op.ext_off;
  const bool ns = op.aux & 1, mp = op.aux & 2, en = e[16] != kNoW;
  auto put = [&](int c, uint32_t v) { out[(size_t)c * h + r] = v; };
  const uint32_t one = C::one();
  for (int l = 0; l < 8; ++l) {
    const bool ctl = e[l] != kNoW;
    put(l, ctl ? one : 0);
    put(10 + l, (!ns && !mp && !ctl) ? one : 0);
    put(18 + l, (!ns && mp && !ctl) ? one : 0);
  }
  put(8, C::mont(op.b));   // absorb_len (<= 255: validated)
  put(9, !ns ? one : 0);
  for (int l = 0; l < 16; ++l) put(kP2D1Hdr + l, C::scaled(e[l] != kNoW ? e[l] : 0, D));
  for (int l = 0; l < 8; ++l) {
    const uint32_t w = e[19 + l];
    put(kP2D1Hdr + 16 + l, C::scaled(w != kNoW ? w : 0, D));
    put(kP2D1Hdr + 24 + l, w == kNoW ? 0u : (wflags[w] & WF_DUP_P2) ? C::neg1() : C::mult(reads, w));
  }
  put(kP2D1Tail, C::scaled(en ? e[16] : 0, D));
  put(kP2D1Tail + 1, (en && mp) ? one : 0);
  put(kP2D1Tail + 2, ns ? one : 0);
  put(kP2D1Tail + 3, mp ? one : 0);
}

// AluPrepLaneCols of ALU row j (common.rs:198-323), Montgomery
template <class PP>
__device__ __forceinline__ void alu_prep13(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ alu_ops,
                                           const uint32_t* __restrict__ roles, const uint32_t* __restrict__ reads, uint32_t j,
                                           uint32_t D, uint32_t row[13]) {
  using C = Cells<PP>;
  const Op op = load_op(ops, alu_ops[j]);
  const uint32_t r = roles[j];
  const uint32_t a_state = r & 0xFF, c_state = (r >> 8) & 0xFF, b_creator = (r >> 16) & 0xFF, out_creator = r >> 24;
  const uint32_t c_w = op.c != kNoW ? op.c : 0;
  const uint32_t one = C::one();
  auto reader_col = [&](uint32_t st, uint32_t w) { return st == 1 ? one : st == 2 ? C::neg_mult(reads, w) : 0u; };
  row[0] = C::neg1();
  row[1] = op.kind == P3R_OP_ALU_ADD ? one : 0;
  row[2] = op.kind == P3R_OP_ALU_BOOL_CHECK ? one : 0;
  row[3] = op.kind == P3R_OP_ALU_MUL_ADD ? one : 0;
  row[4] = op.kind == P3R_OP_ALU_HORNER_ACC ? one : 0;
  row[5] = C::scaled(op.a, D);
  row[6] = C::scaled(op.b, D);
  row[7] = C::scaled(c_w, D);
  row[8] = C::scaled(op.out, D);
  row[9] = b_creator ? C::mult(reads, op.b) : C::neg1();
  row[10] = out_creator ? C::mult(reads, op.out) : C::neg1();
  row[11] = reader_col(a_state, op.a);
  row[12] = reader_col(c_state, c_w);
}

// ---- AluAir::compute_schedule (alu_air.rs:349-463) without the walk.  HornerAcc ops that are consecutive in the
// ALU table form a chain; inside a chain, a maximal run of steps with one multiplier b packs greedily into groups
// of up to K (the last group of a run takes what is left; a group of one is an ordinary op).  Every group starts a
// row at lane 0; the row before the first chain and the row between two chains start with a separator; the free
// lanes of all those rows take the non-chain ops in order, and what is left of them fills rows of its own.
__global__ void __launch_bounds__(kB) k_alu_marks(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ alu_ops, size_t n,
                                                  int pack_k, uint32_t* __restrict__ nonchain /* 1: not a HornerAcc */,
                                                  uint32_t* __restrict__ chain_start, uint32_t* __restrict__ heads /* groups of the run starting here */) {
  const size_t j = (size_t)blockIdx.x * kB + threadIdx.x;
  if (j >= n) return;
  auto kind = [&](size_t x) { return ops[8 * (size_t)alu_ops[x]]; };
  auto bw = [&](size_t x) { return ops[8 * (size_t)alu_ops[x] + 2]; };
  const bool h = kind(j) == P3R_OP_ALU_HORNER_ACC;
  nonchain[j] = !h;
  const bool prev_h = j > 0 && kind(j - 1) == P3R_OP_ALU_HORNER_ACC;
  chain_start[j] = h && !prev_h;
  uint32_t groups = 0;
  if (h && (!prev_h || bw(j - 1) != bw(j))) {  // first step of a same-b run: its length decides its groups
    size_t len = 1;
    while (j + len < n && kind(j + len) == P3R_OP_ALU_HORNER_ACC && bw(j + len) == bw(j)) ++len;
    groups = (uint32_t)((len + pack_k - 1) / pack_k);
  }
  heads[j] = groups;
}
// plan entries of the chain rows (one thread per same-b run) and of the non-chain ops
__global__ void __launch_bounds__(kB) k_alu_plan(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ alu_ops, size_t n, int lanes,
                                                 int pack_k, const uint32_t* __restrict__ nonchain_rank, const uint32_t* __restrict__ chain_start,
                                                 const uint32_t* __restrict__ chain_rank, const uint32_t* __restrict__ heads,
                                                 const uint32_t* __restrict__ heads_rank,
                                                 size_t chain_rows, AluPlanEntry* __restrict__ plan) {
  const size_t j = (size_t)blockIdx.x * kB + threadIdx.x;
  if (j >= n) return;
  const bool h = ops[8 * (size_t)alu_ops[j]] == P3R_OP_ALU_HORNER_ACC;
  if (!h) {
    const size_t t = nonchain_rank[j], fill = chain_rows * (size_t)(lanes - 1);
    size_t row, lane;
    if (t < fill) { row = t / (lanes - 1); lane = 1 + t % (lanes - 1); }
    else { row = chain_rows + (t - fill) / lanes; lane = (t - fill) % lanes; }
    plan[row * lanes + lane] = AluPlanEntry{(uint32_t)j, PLAN_OP, 1, 0};
    return;
  }
  const uint32_t groups = heads[j];
  if (!groups) return;
  size_t len = 1;
  while (j + len < n && ops[8 * (size_t)alu_ops[j + len]] == P3R_OP_ALU_HORNER_ACC &&
         ops[8 * (size_t)alu_ops[j + len] + 2] == ops[8 * (size_t)alu_ops[j] + 2])
    ++len;
  // rows before this run's first group: the leading separator row, one row per earlier group, one separator per
  // earlier chain (chain_rank counts the chain starts BEFORE j; a run that opens its chain has not counted it yet)
  const size_t chain_idx = chain_rank[j] + chain_start[j] - 1;
  const size_t row = 1 + heads_rank[j] + chain_idx;
  for (uint32_t g = 0; g < groups; ++g) {
    const size_t first = j + (size_t)g * pack_k, size = std::min<size_t>(pack_k, len - (size_t)g * pack_k);
    plan[(row + g) * lanes] = size >= 2 ? AluPlanEntry{(uint32_t)first, PLAN_PACKED, (uint8_t)size, 0} : AluPlanEntry{(uint32_t)first, PLAN_OP, 1, 0};
  }
}
// the unscheduled table (no HornerAcc at all): ops in order, `lanes` per row
__global__ void __launch_bounds__(kB) k_alu_plan_flat(size_t n, AluPlanEntry* __restrict__ plan) {
  const size_t j = (size_t)blockIdx.x * kB + threadIdx.x;
  if (j < n) plan[j] = AluPlanEntry{(uint32_t)j, PLAN_OP, 1, 0};
}
__global__ void __launch_bounds__(kB) k_plan_init(size_t n, AluPlanEntry* __restrict__ plan) {
  const size_t j = (size_t)blockIdx.x * kB + threadIdx.x;
  if (j < n) plan[j] = AluPlanEntry{0, PLAN_SEP, 1, 0};
}
// previous lane-0 output feeding each row's packed-Horner accumulator (alu_air.rs:513-589)
__global__ void __launch_bounds__(kB) k_alu_prev(const AluPlanEntry* __restrict__ plan, size_t rows, int lanes, int any,
                                                 uint32_t* __restrict__ prev_src) {
  const size_t r = (size_t)blockIdx.x * kB + threadIdx.x;
  if (r >= rows) return;
  uint32_t prev = kUnset;
  if (any && r > 0) {
    const AluPlanEntry e = plan[(r - 1) * lanes];
    if (e.kind == PLAN_OP) prev = e.first;
    else if (e.kind == PLAN_PACKED) prev = e.first + e.k - 1;
  }
  prev_src[r] = prev;
}
// scheduled preprocessed trace (alu_air.rs:613-677): one thread per (row, lane)
template <class PP>
__global__ void __launch_bounds__(kB) k_prep_alu(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ alu_ops,
                                                 const uint32_t* __restrict__ roles, const uint32_t* __restrict__ reads,
                                                 const AluPlanEntry* __restrict__ plan, size_t rows, int lanes, int k_max, uint32_t D,
                                                 size_t h, uint32_t* __restrict__ out) {
  const size_t s = (size_t)blockIdx.x * kB + threadIdx.x;
  if (s >= rows * (size_t)lanes) return;
  using F = Fp<PP>;
  const size_t row = s / lanes, lane = s % lanes;
  const AluPlanEntry en = plan[s];
  auto put = [&](size_t c, uint32_t v) { out[c * h + row] = v; };
  uint32_t p[13];
  if (en.kind == PLAN_OP) {
    alu_prep13<PP>(ops, alu_ops, roles, reads, en.first, D, p);
    for (int c = 0; c < 13; ++c) put(lane * 13 + c, p[c]);
  } else if (en.kind == PLAN_PACKED && lane == 0) {
    const int k = en.k;
    uint32_t last[13];
    alu_prep13<PP>(ops, alu_ops, roles, reads, en.first, D, p);
    alu_prep13<PP>(ops, alu_ops, roles, reads, en.first + k - 1, D, last);
    p[8] = last[8];
    p[10] = last[10];
    p[9] = (F::raw(p[9]) * F::from_canonical((uint32_t)k)).v;
    for (int c = 0; c < 13; ++c) put(c, p[c]);
    const F mult_a = F::raw(p[0]);
    const size_t extra = (size_t)lanes * 13;
    put(extra + (k - 2), F::one().v);
    for (int t = 1; t < k; ++t) {
      uint32_t st[13];
      alu_prep13<PP>(ops, alu_ops, roles, reads, en.first + t, D, st);
      const size_t q = extra + (k_max - 1) + 6 * (t - 1);
      put(q, st[5]); put(q + 1, st[7]); put(q + 2, st[11]); put(q + 3, st[12]);
      put(q + 4, (mult_a * F::raw(st[11])).v);
      put(q + 5, (mult_a * F::raw(st[12])).v);
    }
  }
}

inline size_t padded_h(size_t rows, size_t min_height) {
  size_t h = 1;
  while (h < std::max<size_t>(rows, 1)) h <<= 1;
  size_t mh = 1;
  while (mh < min_height) mh <<= 1;
  return std::max(h, mh);
}
inline std::unique_ptr<p3r_dmat> zero_mat(p3r_ctx* ctx, size_t h, size_t w) {
  auto m = std::make_unique<p3r_dmat>();
  m->buf.alloc(h * w);
  m->d = m->buf.p;
  m->h = h;
  m->w = w;
  P3R_HIP(hipMemsetAsync(m->d, 0, h * w * 4, ctx->stream));
  return m;
}

// ------------------------------------------------------------------------------------------------ execution schedule
// static facts of every op given stime (what the sequential builder reads off its `set` array): direction of an
// Add / Mul, writes that are comparisons, HornerAcc steps that can join a scan, reads of witnesses nobody sets
__global__ void __launch_bounds__(kB) k_sched_static(const uint32_t* __restrict__ ops, size_t n_ops, const uint32_t* __restrict__ ext,
                                                     const uint32_t* __restrict__ stime, Shape sh, uint32_t* __restrict__ oflags,
                                                     uint32_t* __restrict__ bad) {
  const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
  if (i >= n_ops) return;
  const Op op = load_op(ops, i);
  const uint32_t t = (uint32_t)i + 1;
  const uint32_t* e = ext + op.ext_off;
  auto set = [&](uint32_t w) { return stime[w] < t; };
  bool missing = false;
  auto need = [&](uint32_t w) { if (!set(w)) missing = true; };
  uint32_t f = 0;
  switch (op.kind) {
    case P3R_OP_CONST: f = OF_LIGHT | (set(op.out) ? RUN_CHECK_OUT : 0); break;
    case P3R_OP_PUBLIC: need(op.out); break;
    case P3R_OP_ALU_ADD: case P3R_OP_ALU_MUL:
      f = OF_LIGHT;
      need(op.a);
      if (set(op.b)) { if (set(op.out)) f |= RUN_CHECK_OUT; }
      else { need(op.out); f |= RUN_BACKWARD; }
      break;
    case P3R_OP_ALU_BOOL_CHECK: f = OF_LIGHT | (set(op.out) ? RUN_CHECK_OUT : 0); need(op.a); break;
    case P3R_OP_ALU_MUL_ADD:
      f = OF_LIGHT;
      need(op.a); need(op.b);
      if (op.aux != kNoW && set(op.aux)) f |= RUN_CHECK_AUX;
      if (op.c != kNoW && op.c != op.aux) need(op.c);
      if (op.aux != kNoW && op.out == op.aux) f |= RUN_CHECK_OUT;
      else if (set(op.out)) f |= RUN_CHECK_OUT;
      break;
    case P3R_OP_ALU_HORNER_ACC: {
      const bool ready = set(op.aux) && set(op.a) && set(op.b) && set(op.c) && !set(op.out) && op.out != op.a &&
                         op.out != op.b && op.out != op.c && op.out != op.aux;
      if (ready) f = OF_READY;
      else {
        f = OF_LIGHT | (set(op.out) ? RUN_CHECK_OUT : 0);
        need(op.aux); need(op.a); need(op.b); need(op.c);
      }
      break;
    }
    case P3R_OP_HINT_EXT_DECOMPOSITION: case P3R_OP_HINT_BINARY_DECOMPOSITION: f = OF_LIGHT; need(op.a); break;
    case P3R_OP_RECOMPOSE:
      f = OF_LIGHT | (set(op.out) ? RUN_CHECK_OUT : 0);
      for (uint32_t k = 0; k < sh.D; ++k) need(e[k]);
      break;
    case P3R_OP_POSEIDON2_PERM:
      for (uint32_t l = 0; l < sh.il + 2; ++l) if (e[l] != kNoW) need(e[l]);
      break;
    case P3R_OP_POSEIDON2_W32_PERM:
      for (uint32_t l = 0; l < kW32NOutSlot; ++l) if (e[l] != kNoW) need(e[l]);
      break;
    default: break;
  }
  oflags[i] = f;
  if (missing) atomicOr(bad, BAD_DEFERRED);
}
// a ready HornerAcc continues the scan of the op before it: same multiplier, accumulator = that op's output
__global__ void __launch_bounds__(kB) k_chain_link(const uint32_t* __restrict__ ops, size_t n_ops, uint32_t* __restrict__ oflags) {
  const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
  if (i >= n_ops || i == 0) return;
  if (!(oflags[i] & OF_READY) || !(oflags[i - 1] & OF_READY)) return;
  if (ops[8 * i + 5] == ops[8 * (i - 1) + 4] && ops[8 * i + 2] == ops[8 * (i - 1) + 2]) atomicOr(&oflags[i], OF_LINK);
}

struct FlagReady { const uint32_t* f; __device__ uint32_t operator()(size_t i) const { return (f[i] & OF_READY) ? 1u : 0u; } };
struct FlagLight { const uint32_t* f; __device__ uint32_t operator()(size_t i) const { return (f[i] & OF_LIGHT) ? 1u : 0u; } };
template <class Fn>
void scan_flags(p3r_ctx* ctx, Fn fn, size_t n, uint32_t* out /* n + 1 */) {
  prims::exclusive_sum<uint32_t>(ctx->stream, fn, n, out);   // the total: k_flags_total
}
template <class Fn>
__global__ void k_flags_total(Fn fn, size_t n, uint32_t* __restrict__ excl) {
  if (threadIdx.x == 0 && blockIdx.x == 0) excl[n] = n ? excl[n - 1] + fn(n - 1) : 0;
}

// compact: members of the Horner scans (ready steps, op order), the light ops (op order)
__global__ void __launch_bounds__(kB) k_compact_ops(const uint32_t* __restrict__ oflags, size_t n_ops, const uint32_t* __restrict__ ready_rank,
                                                    const uint32_t* __restrict__ light_rank, uint32_t* __restrict__ hmem,
                                                    uint32_t* __restrict__ run_start /* per member: 1 = no link to the member before */,
                                                    uint32_t* __restrict__ light_ops) {
  const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
  if (i >= n_ops) return;
  const uint32_t f = oflags[i];
  if (f & OF_READY) { hmem[ready_rank[i]] = (uint32_t)i; run_start[ready_rank[i]] = !(f & OF_LINK); }
  if (f & OF_LIGHT) light_ops[light_rank[i]] = (uint32_t)i;
}
__global__ void __launch_bounds__(kB) k_compact_marked(const uint32_t* __restrict__ marks, const uint32_t* __restrict__ rank, size_t n,
                                                       uint32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
  if (i < n && marks[i]) out[rank[i]] = (uint32_t)i;
}

// Poseidon2 rows: position of a row among the rows of its mode (sponge / Merkle), and the mode lists
__global__ void __launch_bounds__(kB) k_p2_modes(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ p2_ops, size_t n,
                                                 uint32_t* __restrict__ is_merkle) {
  const size_t r = (size_t)blockIdx.x * kB + threadIdx.x;
  if (r < n) is_merkle[r] = (ops[8 * (size_t)p2_ops[r] + 5] >> 1) & 1;
}
// mlist: sponge rows in order, then Merkle rows in order; run_start: the row opens a new chain state
__global__ void __launch_bounds__(kB) k_p2_lists(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ p2_ops, size_t n,
                                                 const uint32_t* __restrict__ merkle_rank /* n + 1 */, uint32_t* __restrict__ mlist,
                                                 uint32_t* __restrict__ mpos, uint32_t* __restrict__ run_start, uint32_t* __restrict__ bad) {
  const size_t r = (size_t)blockIdx.x * kB + threadIdx.x;
  if (r >= n) return;
  const uint32_t aux = ops[8 * (size_t)p2_ops[r] + 5];
  const bool merkle = aux & 2, ns = aux & 1;
  const uint32_t n_normal = (uint32_t)n - merkle_rank[n];
  const uint32_t k = merkle ? merkle_rank[r] : (uint32_t)r - merkle_rank[r];
  const uint32_t pos = merkle ? n_normal + k : k;
  mlist[pos] = (uint32_t)r;
  mpos[r] = pos;
  run_start[pos] = ns || k == 0;
  if (!ns && k == 0) atomicOr(bad, BAD_DEFERRED);  // Poseidon2ChainMissingPreviousState
}
// NonPrimitiveOpId -> Poseidon2 row (bit 31: Merkle row)
__global__ void __launch_bounds__(kB) k_op_ids(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ p2_ops, size_t n,
                                               size_t n_ids, uint32_t* __restrict__ row_of_id, uint32_t* __restrict__ bad) {
  const size_t r = (size_t)blockIdx.x * kB + threadIdx.x;
  if (r >= n) return;
  const size_t o = (size_t)p2_ops[r] * 8;
  const uint32_t id = ops[o + 1];
  if (id >= n_ids) { atomicOr(bad, BAD_OP); return; }
  const uint32_t v = (uint32_t)r | (((ops[o + 5] >> 1) & 1) << 31);
  if (atomicCAS(&row_of_id[id], kNoW, v) != kNoW) atomicOr(bad, BAD_DUP_ID);
}
struct MaxNpoId {
  const uint32_t* ops;
  __device__ uint32_t operator()(size_t i) const {
    const uint32_t k = ops[8 * i];
    return (k == P3R_OP_POSEIDON2_PERM || k == P3R_OP_RECOMPOSE || k == P3R_OP_POSEIDON2_W32_PERM) ? ops[8 * i + 1] + 1 : 0u;
  }
};

// ---- levels: chaotic iteration of "level = 1 + highest level among what the op reads or compares against" -----------
// light ops: one thread each
__global__ void __launch_bounds__(kB) k_level_light(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ ext,
                                                    const uint32_t* __restrict__ light_ops, size_t n_light, const uint32_t* __restrict__ oflags,
                                                    const uint32_t* __restrict__ stime, uint32_t D, uint32_t* __restrict__ wlevel,
                                                    uint32_t* __restrict__ olevel, uint32_t* __restrict__ changed) {
  const size_t k = (size_t)blockIdx.x * kB + threadIdx.x;
  if (k >= n_light) return;
  const uint32_t i = light_ops[k];
  const Op op = load_op(ops, i);
  const uint32_t f = oflags[i], t = i + 1;
  const uint32_t* e = ext + op.ext_off;
  uint32_t lvl = 0;
  auto dep = [&](uint32_t w) { lvl = max(lvl, wlevel[w]); };
  const bool chk_out = f & RUN_CHECK_OUT;
  switch (op.kind) {
    case P3R_OP_CONST: if (chk_out) dep(op.out); break;
    case P3R_OP_ALU_ADD: case P3R_OP_ALU_MUL:
      dep(op.a);
      if (f & RUN_BACKWARD) dep(op.out);
      else { dep(op.b); if (chk_out) dep(op.out); }
      break;
    case P3R_OP_ALU_BOOL_CHECK: dep(op.a); if (chk_out) dep(op.out); break;
    case P3R_OP_ALU_MUL_ADD:
      dep(op.a); dep(op.b);
      if (f & RUN_CHECK_AUX) dep(op.aux);
      if (op.c != kNoW && op.c != op.aux) dep(op.c);
      if (chk_out && !(op.aux != kNoW && op.out == op.aux)) dep(op.out);
      break;
    case P3R_OP_ALU_HORNER_ACC: dep(op.aux); dep(op.a); dep(op.b); dep(op.c); if (chk_out) dep(op.out); break;
    case P3R_OP_HINT_EXT_DECOMPOSITION: case P3R_OP_HINT_BINARY_DECOMPOSITION:
      dep(op.a);
      for (uint32_t q = 0; q < op.ext_len; ++q) {
        const uint32_t w = e[q];
        bool dup = false;
        for (uint32_t j = 0; j < q; ++j) dup |= e[j] == w;
        if (!dup && stime[w] < t) dep(w);
      }
      break;
    case P3R_OP_RECOMPOSE: for (uint32_t q = 0; q < D; ++q) dep(e[q]); if (chk_out) dep(op.out); break;
    default: break;
  }
  lvl += 1;
  if (olevel[i] == lvl) return;
  olevel[i] = lvl;
  *changed = 1;
  // the witnesses this op sets (stime[w] == t in a writing role) take its level
  auto wr = [&](uint32_t w) { wlevel[w] = lvl; };
  switch (op.kind) {
    case P3R_OP_CONST: case P3R_OP_ALU_BOOL_CHECK: case P3R_OP_ALU_HORNER_ACC: case P3R_OP_RECOMPOSE: if (!chk_out) wr(op.out); break;
    case P3R_OP_ALU_ADD: case P3R_OP_ALU_MUL:
      if (f & RUN_BACKWARD) wr(op.b); else if (!chk_out) wr(op.out);
      break;
    case P3R_OP_ALU_MUL_ADD:
      if (op.aux != kNoW && !(f & RUN_CHECK_AUX)) wr(op.aux);
      if (!chk_out) wr(op.out);
      break;
    case P3R_OP_HINT_EXT_DECOMPOSITION: case P3R_OP_HINT_BINARY_DECOMPOSITION:
      for (uint32_t q = 0; q < op.ext_len; ++q) if (stime[e[q]] == t) wr(e[q]);
      break;
    default: break;
  }
}

__device__ __forceinline__ uint32_t wave_prefix_max(uint32_t v, int lane) {
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t u = __shfl_up(v, d);
    if (lane >= d) v = max(v, u);
  }
  return v;
}
// Horner scans: a static run of linked ready steps splits wherever an operand only becomes ready at or after the
// level the scan runs at; the level of step j is the running maximum of 1 + level(operands) - one wave per run.
__global__ void __launch_bounds__(kB) k_level_chains(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ hmem,
                                                     const uint32_t* __restrict__ runs /* member index of each run start */,
                                                     size_t n_runs, size_t n_members, uint32_t* __restrict__ wlevel,
                                                     uint32_t* __restrict__ olevel, uint32_t* __restrict__ chead /* per member */,
                                                     uint32_t* __restrict__ changed) {
  const size_t run = ((size_t)blockIdx.x * kB + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (run >= n_runs) return;
  const uint32_t m0 = runs[run], m1 = run + 1 < n_runs ? runs[run + 1] : (uint32_t)n_members;
  uint32_t carry = 0;  // level of the step before this round
  bool ch = false;
  for (uint32_t base = m0; base < m1; base += 64) {
    const uint32_t m = base + lane;
    uint32_t need = 0, i = 0, out = 0;
    if (m < m1) {
      i = hmem[m];
      const Op op = load_op(ops, i);
      out = op.out;
      need = max(wlevel[op.a], wlevel[op.c]) + 1;
      if (m == m0) need = max(need, max(wlevel[op.aux], wlevel[op.b]) + 1);
    }
    uint32_t lvl = max(wave_prefix_max(need, lane), carry);
    uint32_t prev = __shfl_up(lvl, 1);
    if (lane == 0) prev = carry;
    if (m < m1) {
      const uint32_t head = (m == m0 || lvl > prev) ? 1u : 0u;
      if (olevel[i] != lvl || chead[m] != head) { olevel[i] = lvl; chead[m] = head; wlevel[out] = lvl; ch = true; }
    }
    carry = __shfl(lvl, 63);
  }
  if (ch) *changed = 1;
}
// Poseidon2 chains: a run of chained rows (from a new_start row to the next one of its mode) is cut into segments the
// same way: a row whose witness inputs are not ready before the open segment's level opens a new one above it.
__global__ void __launch_bounds__(kB) k_level_p2(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ ext,
                                                 const uint32_t* __restrict__ p2_ops, const uint32_t* __restrict__ mlist,
                                                 const uint32_t* __restrict__ runs, size_t n_runs, size_t n_rows,
                                                 const uint32_t* __restrict__ stime, Shape sh, uint32_t* __restrict__ wlevel,
                                                 uint32_t* __restrict__ plevel /* per mode position */, uint32_t* __restrict__ phead,
                                                 uint32_t* __restrict__ changed) {
  const size_t run = ((size_t)blockIdx.x * kB + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (run >= n_runs) return;
  const uint32_t m0 = runs[run], m1 = run + 1 < n_runs ? runs[run + 1] : (uint32_t)n_rows;
  uint32_t carry = 0;
  bool ch = false;
  for (uint32_t base = m0; base < m1; base += 64) {
    const uint32_t m = base + lane;
    uint32_t need = 0;
    uint32_t outs_mask = 0;            // bit l: output l is written by this row
    const uint32_t* eo = nullptr;      // its output list
    if (m < m1) {
      const uint32_t i = p2_ops[mlist[m]], t = i + 1;
      const Op op = load_op(ops, i);
      const uint32_t* e = ext + op.ext_off;
      eo = e + sh.il + 3;
      uint32_t lvl = 0;
      for (uint32_t l = 0; l < sh.il + 2; ++l) if (e[l] != kNoW) lvl = max(lvl, wlevel[e[l]]);
      for (uint32_t l = 0; l < e[sh.il + 2]; ++l) {
        const uint32_t w = eo[l];
        if (w == kNoW) continue;
        bool earlier = false;
        for (uint32_t j = 0; j < l; ++j) earlier |= eo[j] == w;
        if (earlier) continue;
        if (stime[w] < t) lvl = max(lvl, wlevel[w]);  // a comparison: the row waits for the value
        else outs_mask |= 1u << l;                    // written by this row
      }
      need = lvl + 1;
    }
    uint32_t lvl = max(wave_prefix_max(need, lane), carry);
    uint32_t prev = __shfl_up(lvl, 1);
    if (lane == 0) prev = carry;
    if (m < m1) {
      const uint32_t head = (m == m0 || lvl > prev) ? 1u : 0u;
      if (plevel[m] != lvl || phead[m] != head) {
        plevel[m] = lvl; phead[m] = head; ch = true;
        for (uint32_t l = 0; l < 16; ++l) if (outs_mask & (1u << l)) wlevel[eo[l]] = lvl;
      }
    }
    carry = __shfl(lvl, 63);
  }
  if (ch) *changed = 1;
}

// ---- the width-32 table (P3R_OP_POSEIDON2_W32_PERM): its own op type with its own chain state (circuit_host.h::build_schedule,
// update_chain_state executor.rs:462-491).  Every row updates the Merkle state and a sponge row the sponge state too, so a chained
// Merkle row continues the row just before it and a chained sponge row the sponge row before it: one predecessor per row, at
// most two successors - row p + 1 and the next sponge row - and the FIRST of them in circuit order whose witnesses are ready
// before p's segment runs extends that segment; any other opens a segment above it.
__global__ void __launch_bounds__(kB) k_pw_modes(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ pw_ops, size_t n,
                                                 uint32_t* __restrict__ is_sponge) {
  const size_t r = (size_t)blockIdx.x * kB + threadIdx.x;
  if (r < n) is_sponge[r] = ((ops[8 * (size_t)pw_ops[r] + 5] >> 1) & 1) ^ 1u;
}
__global__ void __launch_bounds__(kB) k_pw_links(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ pw_ops, size_t n,
                                                 const uint32_t* __restrict__ sponge_rank, const uint32_t* __restrict__ sponge_list,
                                                 uint32_t* __restrict__ prev, uint32_t* __restrict__ bad) {
  const size_t r = (size_t)blockIdx.x * kB + threadIdx.x;
  if (r >= n) return;
  const uint32_t aux = ops[8 * (size_t)pw_ops[r] + 5];
  uint32_t p = kNoW;
  if (!(aux & 1)) {
    if (aux & 2) p = r ? (uint32_t)r - 1 : kNoW;
    else p = sponge_rank[r] ? sponge_list[sponge_rank[r] - 1] : kNoW;
    if (p == kNoW) atomicOr(bad, BAD_DEFERRED);   // Poseidon2ChainMissingPreviousState
  }
  prev[r] = p;
}
// the successor of row p that shares its segment, or kNoW
__device__ __forceinline__ uint32_t pw_next_in_segment(uint32_t p, size_t n, const uint32_t* __restrict__ prev,
                                                       const uint32_t* __restrict__ joined, const uint32_t* __restrict__ is_sponge,
                                                       const uint32_t* __restrict__ sponge_rank, const uint32_t* __restrict__ sponge_list,
                                                       size_t n_sponge) {
  if ((size_t)p + 1 < n && prev[p + 1] == p && joined[p + 1]) return p + 1;
  if (is_sponge[p] && (size_t)sponge_rank[p] + 1 < n_sponge) {
    const uint32_t c = sponge_list[sponge_rank[p] + 1];
    if (c != p + 1 && prev[c] == p && joined[c]) return c;
  }
  return kNoW;
}
// one thread per row; the fixed point of "level = the segment's level" over rows that depend on earlier rows only
__global__ void __launch_bounds__(kB) k_level_p2w(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ ext,
                                                  const uint32_t* __restrict__ pw_ops, size_t n, const uint32_t* __restrict__ prev,
                                                  const uint32_t* __restrict__ is_sponge, const uint32_t* __restrict__ stime,
                                                  uint32_t* __restrict__ wlevel, uint32_t* __restrict__ pwlevel,
                                                  uint32_t* __restrict__ joined, uint32_t* __restrict__ changed) {
  const size_t r = (size_t)blockIdx.x * kB + threadIdx.x;
  if (r >= n) return;
  const uint32_t i = pw_ops[r], t = i + 1;
  const Op op = load_op(ops, i);
  const uint32_t* e = ext + op.ext_off;
  const uint32_t* eo = e + kW32Hdr;
  uint32_t lvl = 0, outs_mask = 0;
  for (uint32_t l = 0; l < kW32NOutSlot; ++l) if (e[l] != kNoW) lvl = max(lvl, wlevel[e[l]]);
  for (uint32_t l = 0; l < e[kW32NOutSlot]; ++l) {
    const uint32_t w = eo[l];
    if (w == kNoW) continue;
    bool earlier = false;
    for (uint32_t j = 0; j < l; ++j) earlier |= eo[j] == w;
    if (earlier) continue;
    if (stime[w] < t) lvl = max(lvl, wlevel[w]);  // a comparison: the row waits for the value
    else outs_mask |= 1u << l;                    // written by this row
  }
  const uint32_t need = lvl + 1, p = prev[r];
  uint32_t level = need, join = 0;
  if (p != kNoW) {
    const uint32_t lp = pwlevel[p];
    bool can = need <= lp;
    // a sponge row's predecessor may have a Merkle successor, row p + 1, which comes first in circuit order
    if (can && is_sponge[r] && p + 1 < r && prev[p + 1] == p && joined[p + 1]) can = false;
    level = can ? lp : max(need, lp + 1);
    join = can;
  }
  if (pwlevel[r] == level && joined[r] == join) return;
  pwlevel[r] = level;
  joined[r] = join;
  *changed = 1;
  for (uint32_t l = 0; l < kW32In; ++l) if (outs_mask & (1u << l)) wlevel[eo[l]] = level;
}
__global__ void __launch_bounds__(kB) k_pw_heads(const uint32_t* __restrict__ joined, size_t n, uint32_t* __restrict__ is_head) {
  const size_t r = (size_t)blockIdx.x * kB + threadIdx.x;
  if (r < n) is_head[r] = joined[r] ^ 1u;
}
__global__ void __launch_bounds__(kB) k_pw_seg_records(const uint32_t* __restrict__ head_rows, size_t n_segs, size_t n,
                                                       const uint32_t* __restrict__ prev, const uint32_t* __restrict__ joined,
                                                       const uint32_t* __restrict__ is_sponge, const uint32_t* __restrict__ sponge_rank,
                                                       const uint32_t* __restrict__ sponge_list, size_t n_sponge,
                                                       const uint32_t* __restrict__ pwlevel, uint32_t* __restrict__ seg_len,
                                                       uint32_t* __restrict__ seg_level) {
  const size_t s = (size_t)blockIdx.x * kB + threadIdx.x;
  if (s >= n_segs) return;
  uint32_t r = head_rows[s], len = 0;
  seg_level[s] = pwlevel[r];
  for (; r != kNoW; r = pw_next_in_segment(r, n, prev, joined, is_sponge, sponge_rank, sponge_list, n_sponge)) ++len;
  seg_len[s] = len;
}
__global__ void __launch_bounds__(kB) k_emit_p2w(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ ext,
                                                 const uint32_t* __restrict__ pw_ops, size_t n, const uint32_t* __restrict__ prev,
                                                 const uint32_t* __restrict__ joined, const uint32_t* __restrict__ is_sponge,
                                                 const uint32_t* __restrict__ sponge_rank, const uint32_t* __restrict__ sponge_list,
                                                 size_t n_sponge, const uint32_t* __restrict__ order, const uint32_t* __restrict__ head_rows,
                                                 const uint32_t* __restrict__ seg_len, const uint32_t* __restrict__ first /* per sorted segment */,
                                                 size_t n_segs, const uint32_t* __restrict__ stime, RunP2W* __restrict__ out,
                                                 RunSchedule::P2Seg* __restrict__ segs) {
  const size_t s = (size_t)blockIdx.x * kB + threadIdx.x;
  if (s >= n_segs) return;
  const uint32_t c = order[s], len = seg_len[c], f0 = first[s];
  segs[s] = RunSchedule::P2Seg{f0, len};
  uint32_t row = head_rows[c];
  for (uint32_t k = 0; k < len; ++k) {
    const uint32_t i = pw_ops[row], t = i + 1;
    const Op op = load_op(ops, i);
    const uint32_t* e = ext + op.ext_off;
    const uint32_t n_out = e[kW32NOutSlot];
    RunP2W q{};
    q.flags = (op.aux & 3) | (n_out << 8);
    q.op_idx = i;
    q.row = row;
    q.prev_row = prev[row];
    q.prev_in_seg = k ? 1u : 0u;
    for (uint32_t l = 0; l < kW32In; ++l) q.in[l] = e[l];
    q.bit_w = e[kW32BitSlot];
    q.bit2_w = e[kW32Bit2Slot];
    for (uint32_t l = 0; l < kW32In; ++l) {
      q.out[l] = l < n_out ? e[kW32Hdr + l] : kNoW;
      if (q.out[l] == kNoW) continue;
      bool earlier = false;
      for (uint32_t j = 0; j < l; ++j) earlier |= q.out[j] == q.out[l];
      if (earlier || stime[q.out[l]] < t) q.flags |= 1u << (16 + l);
    }
    out[f0 + k] = q;
    row = pw_next_in_segment(row, n, prev, joined, is_sponge, sponge_rank, sponge_list, n_sponge);
  }
}
// the two widths share the NonPrimitiveOpId space: one id, one row
__global__ void __launch_bounds__(kB) k_ids_cross(const uint32_t* __restrict__ row_of_id, const uint32_t* __restrict__ roww_of_id,
                                                  size_t n_ids, uint32_t* __restrict__ bad) {
  const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
  if (i < n_ids && row_of_id[i] != kNoW && roww_of_id[i] != kNoW) atomicOr(bad, BAD_DUP_ID);
}
__global__ void __launch_bounds__(kB) k_list_kind(const uint32_t* __restrict__ ops, size_t n_ops, uint32_t kind,
                                                  const uint32_t* __restrict__ rank, uint32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
  if (i < n_ops && ops[8 * i] == kind) out[rank[i]] = (uint32_t)i;
}

// ---- emission ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kB) k_gather_u32(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx, size_t n,
                                                   uint32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
  if (i < n) out[i] = src[idx[i]];
}
// first position of every key value in a SORTED key array (no atomics: 3 M increments of twenty counters serialise)
__global__ void __launch_bounds__(kB) k_first_index(const uint32_t* __restrict__ sorted_keys, size_t n, uint32_t* __restrict__ first) {
  const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
  if (i < n && (i == 0 || sorted_keys[i] != sorted_keys[i - 1])) first[sorted_keys[i]] = (uint32_t)i;
}
__global__ void __launch_bounds__(kB) k_max_u32(const uint32_t* __restrict__ v, size_t n, uint32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
  uint32_t x = i < n ? v[i] : 0u;
  for (int d = 32; d; d >>= 1) x = max(x, __shfl_xor(x, d));
  if ((threadIdx.x & 63) == 0 && x) atomicMax(out, x);
}
__global__ void __launch_bounds__(kB) k_iota(uint32_t* __restrict__ v, size_t n) {
  const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
  if (i < n) v[i] = (uint32_t)i;
}

// the device ext array: constants (Montgomery), hint output lists (bit 31: compare instead of write), recompose inputs
template <class PP>
__global__ void __launch_bounds__(kB) k_emit_ext(const uint32_t* __restrict__ ops, size_t n_ops, const uint32_t* __restrict__ ext,
                                                 const uint64_t* __restrict__ s_re, const uint32_t* __restrict__ stime, uint32_t D,
                                                 uint32_t* __restrict__ dev_ext) {
  const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
  if (i >= n_ops) return;
  const Op op = load_op(ops, i);
  const uint32_t off = (uint32_t)(s_re[i] >> 32), t = (uint32_t)i + 1;
  const uint32_t* e = ext + op.ext_off;
  if (op.kind == P3R_OP_CONST) for (uint32_t k = 0; k < D; ++k) dev_ext[off + k] = Fp<PP>::from_canonical(e[k]).v;
  else if (op.kind == P3R_OP_RECOMPOSE) for (uint32_t k = 0; k < D; ++k) dev_ext[off + k] = e[k];
  else if (is_hint(op.kind))
    for (uint32_t k = 0; k < op.ext_len; ++k) {
      const uint32_t w = e[k];
      bool dup = false;
      for (uint32_t j = 0; j < k; ++j) dup |= e[j] == w;
      dev_ext[off + k] = (dup || stime[w] < t) ? (w | RUN_CHECK_BIT) : w;
    }
}
// (s_rc / n_rec_plain: the trace rows of the `recompose/coeff` ops follow those of the plain ops in the one array)
__device__ __forceinline__ RunOp make_run_op(const Op& op, uint32_t i, uint32_t f, const uint64_t* __restrict__ s_ap,
                                              const uint64_t* __restrict__ s_re, const uint32_t* __restrict__ s_rc, uint32_t n_rec_plain) {
  RunOp r{};
  r.kind_flags = op.kind | (f & (RUN_BACKWARD | RUN_CHECK_OUT | RUN_CHECK_AUX));
  if (is_hint(op.kind)) r.kind_flags |= op.ext_len << 16;
  r.a = op.a; r.b = op.b; r.c = op.c; r.out = op.out; r.aux = op.aux;
  r.op_idx = i;
  if (is_alu(op.kind)) r.rec = (uint32_t)s_ap[i];
  else if (op.kind == P3R_OP_RECOMPOSE) r.rec = op.aux == 1u ? n_rec_plain + s_rc[i] : (uint32_t)s_re[i];
  if (op.kind == P3R_OP_CONST || op.kind == P3R_OP_RECOMPOSE || is_hint(op.kind)) r.ext_off = (uint32_t)(s_re[i] >> 32);
  return r;
}
__global__ void __launch_bounds__(kB) k_emit_light(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ sorted_ops, size_t n,
                                                   const uint32_t* __restrict__ oflags, const uint64_t* __restrict__ s_ap,
                                                   const uint64_t* __restrict__ s_re, const uint32_t* __restrict__ s_rc, uint32_t n_rec_plain,
                                                   RunOp* __restrict__ out) {
  const size_t k = (size_t)blockIdx.x * kB + threadIdx.x;
  if (k >= n) return;
  const uint32_t i = sorted_ops[k];
  out[k] = make_run_op(load_op(ops, i), i, oflags[i], s_ap, s_re, s_rc, n_rec_plain);
}
__global__ void __launch_bounds__(kB) k_emit_chain_ops(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ hmem, size_t n,
                                                       const uint64_t* __restrict__ s_ap, const uint64_t* __restrict__ s_re,
                                                       RunOp* __restrict__ out) {
  const size_t k = (size_t)blockIdx.x * kB + threadIdx.x;
  if (k >= n) return;
  const uint32_t i = hmem[k];
  out[k] = make_run_op(load_op(ops, i), i, 0, s_ap, s_re, nullptr, 0);   // HornerAcc steps only
}
// chains in creation order: {first member, length, acc witness, b witness}, sort key = level * 2 + (short ? 1 : 0)
__global__ void __launch_bounds__(kB) k_chain_records(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ hmem,
                                                      const uint32_t* __restrict__ heads /* member index of each chain head */, size_t n_chains,
                                                      size_t n_members, const uint32_t* __restrict__ olevel,
                                                      RunSchedule::ChainSeg* __restrict__ segs, uint32_t* __restrict__ keys,
                                                      uint32_t* __restrict__ levels) {
  const size_t c = (size_t)blockIdx.x * kB + threadIdx.x;
  if (c >= n_chains) return;
  const uint32_t m0 = heads[c], m1 = c + 1 < n_chains ? heads[c + 1] : (uint32_t)n_members;
  const uint32_t i = hmem[m0];
  const uint32_t lvl = olevel[i];
  segs[c] = RunSchedule::ChainSeg{m0, m1 - m0, ops[8 * (size_t)i + 5], ops[8 * (size_t)i + 2]};
  keys[c] = lvl * 2 + ((m1 - m0) > 128 ? 0u : 1u);
  levels[c] = lvl;
}
__global__ void __launch_bounds__(kB) k_permute_chains(const RunSchedule::ChainSeg* __restrict__ in, const uint32_t* __restrict__ order,
                                                       size_t n, RunSchedule::ChainSeg* __restrict__ out) {
  const size_t c = (size_t)blockIdx.x * kB + threadIdx.x;
  if (c < n) out[c] = in[order[c]];
}
// Poseidon2 segments in creation order (= order of their first rows in the circuit): head marks by ROW, then records
__global__ void __launch_bounds__(kB) k_p2_head_by_row(const uint32_t* __restrict__ mlist, const uint32_t* __restrict__ phead, size_t n,
                                                       uint32_t* __restrict__ head_by_row) {
  const size_t m = (size_t)blockIdx.x * kB + threadIdx.x;
  if (m < n) head_by_row[mlist[m]] = phead[m];
}
__global__ void __launch_bounds__(kB) k_p2_seg_records(const uint32_t* __restrict__ head_rows /* rows that open a segment, row order */,
                                                       size_t n_segs, const uint32_t* __restrict__ mpos, const uint32_t* __restrict__ phead,
                                                       const uint32_t* __restrict__ plevel, size_t n_rows, size_t n_normal,
                                                       uint32_t* __restrict__ seg_pos, uint32_t* __restrict__ seg_len,
                                                       uint32_t* __restrict__ seg_level) {
  const size_t s = (size_t)blockIdx.x * kB + threadIdx.x;
  if (s >= n_segs) return;
  const uint32_t m0 = mpos[head_rows[s]];
  const uint32_t end = m0 < n_normal ? (uint32_t)n_normal : (uint32_t)n_rows;  // a segment never leaves its mode
  uint32_t m = m0 + 1;
  while (m < end && !phead[m]) ++m;
  seg_pos[s] = m0;
  seg_len[s] = m - m0;
  seg_level[s] = plevel[m0];
}
__global__ void __launch_bounds__(kB) k_emit_p2(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ ext,
                                                const uint32_t* __restrict__ p2_ops, const uint32_t* __restrict__ mlist,
                                                const uint32_t* __restrict__ order, const uint32_t* __restrict__ seg_pos,
                                                const uint32_t* __restrict__ seg_len, const uint32_t* __restrict__ first /* per sorted segment */,
                                                size_t n_segs, size_t n_normal, const uint32_t* __restrict__ stime,
                                                RunP2* __restrict__ out, RunSchedule::P2Seg* __restrict__ segs) {
  const size_t s = (size_t)blockIdx.x * kB + threadIdx.x;
  if (s >= n_segs) return;
  const uint32_t c = order[s], m0 = seg_pos[c], len = seg_len[c], f0 = first[s];
  segs[s] = RunSchedule::P2Seg{f0, len};
  for (uint32_t k = 0; k < len; ++k) {
    const uint32_t m = m0 + k, row = mlist[m], i = p2_ops[row], t = i + 1;
    const Op op = load_op(ops, i);
    const uint32_t* e = ext + op.ext_off;
    RunP2 q{};
    q.flags = (op.aux & 3) | (e[6] << 8);
    q.op_idx = i;
    q.row = row;
    for (int l = 0; l < 4; ++l) q.in[l] = e[l];
    q.idx_w = e[4];
    q.bit_w = e[5];
    // the previous permutation of the same mode (last_output_normal / last_output_merkle, executor.rs:340-355)
    q.prev_row = kNoW;
    if (!(op.aux & 1)) q.prev_row = (m == 0 || m == n_normal) ? kNoW : mlist[m - 1];
    for (uint32_t l = 0; l < 4; ++l) {
      q.out[l] = l < e[6] ? e[7 + l] : kNoW;
      if (q.out[l] == kNoW) continue;
      bool earlier = false;
      for (uint32_t j = 0; j < l; ++j) earlier |= q.out[j] == q.out[l];
      if (earlier || stime[q.out[l]] < t) q.flags |= 1u << (4 + l);
    }
    out[f0 + k] = q;
  }
}
__global__ void __launch_bounds__(kB) k_emit_p2_base(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ ext,
                                                const uint32_t* __restrict__ p2_ops, const uint32_t* __restrict__ mlist,
                                                const uint32_t* __restrict__ order, const uint32_t* __restrict__ seg_pos,
                                                const uint32_t* __restrict__ seg_len, const uint32_t* __restrict__ first /* per sorted segment */,
                                                size_t n_segs, size_t n_normal, const uint32_t* __restrict__ stime,
                                                RunP2B* __restrict__ out, RunSchedule::P2Seg* __restrict__ segs) {
  const size_t s = (size_t)blockIdx.x * kB + threadIdx.x;
  if (s >= n_segs) return;
  const uint32_t c = order[s], m0 = seg_pos[c], len = seg_len[c], f0 = first[s];
  segs[s] = RunSchedule::P2Seg{f0, len};
  for (uint32_t k = 0; k < len; ++k) {
    const uint32_t m = m0 + k, row = mlist[m], i = p2_ops[row], t = i + 1;
    const Op op = load_op(ops, i);
    const uint32_t* e = ext + op.ext_off;
    RunP2B q{};
    const uint32_t n_out = e[18];
    q.flags = (op.aux & 3) | (n_out << 8);
    q.op_idx = i;
    q.row = row;
    q.absorb_len = op.b;
    for (int l = 0; l < 16; ++l) q.in[l] = e[l];
    q.idx_w = e[16];
    q.bit_w = e[17];
    // the previous permutation of the same mode (last_output_normal / last_output_merkle, executor.rs:340-355)
    q.prev_row = kNoW;
    if (!(op.aux & 1)) q.prev_row = (m == 0 || m == n_normal) ? kNoW : mlist[m - 1];
    for (uint32_t l = 0; l < 16; ++l) {
      q.out[l] = l < n_out ? e[19 + l] : kNoW;
      if (q.out[l] == kNoW) continue;
      bool earlier = false;
      for (uint32_t j = 0; j < l; ++j) earlier |= q.out[j] == q.out[l];
      if (earlier || stime[q.out[l]] < t) q.check_mask |= 1u << l;
    }
    out[f0 + k] = q;
  }
}
template <class PP>
__global__ void __launch_bounds__(kB) k_const_values(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ ext,
                                                     const uint32_t* __restrict__ const_ops, size_t n, uint32_t D, uint32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
  if (i >= n * D) return;
  out[i] = Fp<PP>::from_canonical(ext[ops[8 * (size_t)const_ops[i / D] + 6] + (uint32_t)(i % D)]).v;
}
__global__ void __launch_bounds__(kB) k_outs_of(const uint32_t* __restrict__ ops, const uint32_t* __restrict__ list, size_t n,
                                                uint32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
  if (i < n) out[i] = ops[8 * (size_t)list[i] + 4];
}
__global__ void __launch_bounds__(kB) k_any_unset(const uint32_t* __restrict__ stime, size_t n, uint32_t* __restrict__ bad) {
  const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
  if (i < n && stime[i] == kUnset) atomicOr(bad, BAD_DEFERRED);
}
__global__ void __launch_bounds__(kB) k_set_zero_at(const uint32_t* __restrict__ idx, size_t n, uint32_t* __restrict__ v) {
  const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
  if (i < n) v[idx[i]] = 0;
}

void sort_by_key(p3r_ctx* ctx, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, size_t n, int bits) {
  if (!n) return;
  prims::sort_pairs(ctx->stream, keys_in, keys_out, vals_in, vals_out, n, bits);
}
inline int bits_for(uint32_t max_value) {
  int b = 1;
  while (b < 32 && (max_value >> b)) ++b;
  return b;
}

template <class T>
std::vector<T> fetch(p3r_ctx* ctx, const void* dev, size_t n) {
  std::vector<T> v(n);
  if (n) P3R_HIP(copy_sync(ctx->stream, v.data(), dev, n * sizeof(T), hipMemcpyDeviceToHost));
  return v;
}

#define LAUNCH(kern, n, ...) \
  do { if ((n) > 0) hipLaunchKernelGGL(kern, dim3(nblk(n)), dim3(kB), 0, ctx->stream, __VA_ARGS__); } while (0)

template <class PP>
bool devprep_impl(p3r_ctx* ctx, const p3r_circuit_desc* d, DevPrep& R) {
  const size_t n_ops = d->n_ops;
  const uint32_t nw = d->witness_count;
  if (nw >= (1u << 31) || n_ops >= (size_t(1) << 28) || n_ops == 0 || nw == 0) return false;  // host path: limits, degenerate circuits
  const uint32_t D = ctx->cfg.ext_degree;
  if (D != 1 && D != 4 && D != 5) return false;   // (the runner computes in these degrees)
  const Shape sh = shape_of(D);
  hipStream_t s = ctx->stream;
  prof_stage(ctx, "prep_upload");
  // ---- the circuit crosses PCIe once
  DevBuf ops(n_ops * 8), ext(std::max<size_t>(d->n_ext, 1));
  P3R_HIP(hipMemcpyAsync(ops.p, d->ops, n_ops * 32, hipMemcpyHostToDevice, s));
  if (d->n_ext) P3R_HIP(hipMemcpyAsync(ext.p, d->ext, d->n_ext * 4, hipMemcpyHostToDevice, s));
  auto up = [&](DevBuf& b, const uint32_t* src, size_t n) {
    b.alloc(std::max<size_t>(n, 1));
    if (n) P3R_HIP(hipMemcpyAsync(b.p, src, n * 4, hipMemcpyHostToDevice, s));
  };
  up(R.d_public_rows, d->public_rows, d->n_public);
  up(R.d_private_rows, d->private_input_rows, d->n_private);
  R.n_public_rows = d->n_public;
  R.n_private_rows = d->n_private;
  for (size_t k = 0; k < 2 * d->n_rewrite; ++k)
    if (d->witness_rewrite[k] >= nw) return false;

  DevBuf bad(2), changed(1);
  P3R_HIP(hipMemsetAsync(bad.p, 0, 8, s));
  DevBuf wflags(nw), tdef(nw), stime(nw), reads(nw), wlevel(nw);
  P3R_HIP(hipMemsetAsync(wflags.p, 0, (size_t)nw * 4, s));
  P3R_HIP(hipMemsetAsync(reads.p, 0, (size_t)nw * 4, s));
  P3R_HIP(hipMemsetAsync(wlevel.p, 0, (size_t)nw * 4, s));
  P3R_HIP(hipMemsetAsync(tdef.p, 0xFF, (size_t)nw * 4, s));
  P3R_HIP(hipMemsetAsync(stime.p, 0xFF, (size_t)nw * 4, s));
  LAUNCH(k_validate<PP>, n_ops, ops.p, n_ops, ext.p, d->n_ext, nw, sh, bad.p);
  LAUNCH(k_mark_rows, d->n_public, R.d_public_rows.p, d->n_public, nw, 0u, wflags.p, stime.p, bad.p);
  LAUNCH(k_mark_rows, d->n_private, R.d_private_rows.p, d->n_private, nw, (uint32_t)WF_PRIVATE, wflags.p, stime.p, bad.p);
  {
    uint32_t b0 = 0;
    P3R_HIP(copy_sync(s, &b0, bad.p, 4, hipMemcpyDeviceToHost));
    if (b0) return false;  // from here on every index is in range
  }
  prof_stage(ctx, "prep_bus_roles");
  // ---- ranks: op -> row of its table, offsets into the device ext array
  DevBuf s_cp, s_ap, s_re;
  scan_pairs(ctx, PairConstPublic{ops.p}, n_ops, s_cp);
  scan_pairs(ctx, PairAluP2{ops.p}, n_ops, s_ap);
  scan_pairs(ctx, PairRecExt{ops.p, D}, n_ops, s_re);
  DevBuf s_rc(n_ops + 1);   // rank among the `recompose/coeff` ops
  scan_flags(ctx, FlagRecCoeff{ops.p}, n_ops, s_rc.p);
  hipLaunchKernelGGL(k_flags_total<FlagRecCoeff>, dim3(1), dim3(64), 0, s, FlagRecCoeff{ops.p}, n_ops, s_rc.p);
  uint64_t* S_cp = reinterpret_cast<uint64_t*>(s_cp.p);
  uint64_t* S_ap = reinterpret_cast<uint64_t*>(s_ap.p);
  uint64_t* S_re = reinterpret_cast<uint64_t*>(s_re.p);
  hipLaunchKernelGGL(k_scan_total<PairConstPublic>, dim3(1), dim3(64), 0, s, PairConstPublic{ops.p}, n_ops, S_cp);
  hipLaunchKernelGGL(k_scan_total<PairAluP2>, dim3(1), dim3(64), 0, s, PairAluP2{ops.p}, n_ops, S_ap);
  hipLaunchKernelGGL(k_scan_total<PairRecExt>, dim3(1), dim3(64), 0, s, PairRecExt{ops.p, D}, n_ops, S_re);
  DevBuf s_pw(n_ops + 1);   // rank among the width-32 Poseidon2 ops
  scan_flags(ctx, FlagP2W{ops.p}, n_ops, s_pw.p);
  hipLaunchKernelGGL(k_flags_total<FlagP2W>, dim3(1), dim3(64), 0, s, FlagP2W{ops.p}, n_ops, s_pw.p);
  uint64_t tot[3];
  uint32_t n_rec_coeff_u32 = 0, n_pw_u32 = 0;
  P3R_HIP(hipMemcpyAsync(&n_pw_u32, s_pw.p + n_ops, 4, hipMemcpyDeviceToHost, s));
  P3R_HIP(hipMemcpyAsync(&n_rec_coeff_u32, s_rc.p + n_ops, 4, hipMemcpyDeviceToHost, s));
  P3R_HIP(hipMemcpyAsync(&tot[0], S_cp + n_ops, 8, hipMemcpyDeviceToHost, s));
  P3R_HIP(hipMemcpyAsync(&tot[1], S_ap + n_ops, 8, hipMemcpyDeviceToHost, s));
  P3R_HIP(copy_sync(s, &tot[2], S_re + n_ops, 8, hipMemcpyDeviceToHost));
  const size_t n_const = (uint32_t)tot[0], n_public = tot[0] >> 32, n_alu_ops = (uint32_t)tot[1], n_p2 = tot[1] >> 32;
  // plain / coefficient-kind Recompose ops; their trace rows share one array, the plain ones first
  const size_t n_rec_plain = (uint32_t)tot[2], n_rec_coeff = n_rec_coeff_u32, n_dev_ext = tot[2] >> 32;
  DevBuf const_ops(std::max<size_t>(n_const, 1)), public_ops(std::max<size_t>(n_public, 1)), alu_ops(std::max<size_t>(n_alu_ops, 1)),
      p2_ops(std::max<size_t>(n_p2, 1)), rec_ops(std::max<size_t>(n_rec_plain, 1)), rec_coeff_ops(std::max<size_t>(n_rec_coeff, 1));
  LAUNCH(k_table_lists, n_ops, ops.p, n_ops, S_cp, S_ap, S_re, const_ops.p, public_ops.p, alu_ops.p, p2_ops.p, rec_ops.p, s_rc.p,
         rec_coeff_ops.p);
  const size_t n_pw = n_pw_u32;
  DevBuf pw_ops(std::max<size_t>(n_pw, 1));
  if (n_pw) LAUNCH(k_list_kind, n_ops, ops.p, n_ops, (uint32_t)P3R_OP_POSEIDON2_W32_PERM, s_pw.p, pw_ops.p);

  // ---- first touches
  LAUNCH(k_mark_cp, n_ops, ops.p, n_ops, wflags.p);
  LAUNCH(k_mark_hint, n_ops, ops.p, n_ops, ext.p, wflags.p);
  LAUNCH(k_times, n_ops, ops.p, n_ops, ext.p, wflags.p, sh, tdef.p, stime.p);
  for (int round = 0;; ++round) {
    P3R_HIP(hipMemsetAsync(changed.p, 0, 4, s));
    for (int k = 0; k < 4; ++k) LAUNCH(k_times_fix, n_ops, ops.p, n_ops, wflags.p, tdef.p, stime.p, changed.p);
    uint32_t c = 0;
    P3R_HIP(copy_sync(s, &c, changed.p, 4, hipMemcpyDeviceToHost));
    if (!c) break;
    if (round > (1 << 20)) fail(P3R_EINVAL, "circuit preparation did not converge");
  }

  // ---- bus roles, read counts
  DevBuf roles(std::max<size_t>(n_alu_ops, 1));
  LAUNCH(k_roles, n_ops, ops.p, n_ops, ext.p, wflags.p, tdef.p, sh, S_ap, reads.p, roles.p);
  const size_t mh = d->min_trace_height;
  R.counts.n_const = n_const; R.counts.n_public = n_public; R.counts.n_alu = std::max<size_t>(n_alu_ops, 1);
  R.counts.n_p2 = n_p2;
  R.counts.n_p2w = n_pw;
  // a table without rows is not proved: a circuit whose Recompose ops are all of the coefficient kind has ONE Recompose
  // table, `recompose/coeff`, in the first slot (circuit_impl.hip.h::circuit_tables)
  R.recompose_coeff = n_rec_plain == 0 && n_rec_coeff > 0;
  R.counts.n_recompose = R.recompose_coeff ? n_rec_coeff : n_rec_plain;
  R.counts.n_recompose_coeff = R.recompose_coeff ? 0 : n_rec_coeff;
  R.public_lanes = n_public <= 1 ? 1 : d->public_lanes;
  R.alu_lanes = R.counts.n_alu <= 1 ? 1 : d->alu_lanes;
  const size_t h_p2 = n_p2 ? padded_h(n_p2, mh) : 0;
  LAUNCH(k_acc_reads, n_p2, ops.p, ext.p, p2_ops.p, n_p2, h_p2, sh, reads.p);
  LAUNCH(k_unclaimed, d->n_private, R.d_private_rows.p, d->n_private, tdef.p, bad.p);

  prof_stage(ctx, "prep_traces");
  // ---- preprocessed traces
  const int lanes = (int)R.alu_lanes, k_max = (int)d->horner_packed_steps, rl = (int)d->recompose_lanes, pl = (int)R.public_lanes;
  R.h[0] = padded_h(std::max<size_t>(n_const, 1), mh);
  R.prep[0] = zero_mat(ctx, R.h[0], 2);
  LAUNCH(k_prep_simple<PP>, n_const, ops.p, const_ops.p, n_const, 1, 0, reads.p, wflags.p, D, R.h[0], R.prep[0]->d);
  R.h[1] = padded_h(std::max<size_t>((n_public + pl - 1) / pl, 1), mh);
  R.prep[1] = zero_mat(ctx, R.h[1], 2 * (size_t)pl);
  LAUNCH(k_prep_simple<PP>, n_public, ops.p, public_ops.p, n_public, pl, 0, reads.p, wflags.p, D, R.h[1], R.prep[1]->d);
  if (n_p2) {
    R.h[3] = h_p2;
    if (D == 4) {
      R.prep[3] = zero_mat(ctx, h_p2, 24);
      LAUNCH(k_prep_p2<PP>, h_p2, ops.p, ext.p, p2_ops.p, n_p2, reads.p, wflags.p, h_p2, R.prep[3]->d);
    } else {
      R.prep[3] = zero_mat(ctx, h_p2, kP2D1PrepWidth);
      LAUNCH(k_prep_p2_d1<PP>, h_p2, ops.p, ext.p, p2_ops.p, n_p2, reads.p, wflags.p, D, h_p2, R.prep[3]->d);
    }
  }
  if (n_pw) {
    R.h[6] = padded_h(n_pw, mh);
    R.prep[6] = zero_mat(ctx, R.h[6], kP2WPrepCols);
    LAUNCH(k_prep_p2w<PP>, R.h[6], ops.p, ext.p, pw_ops.p, n_pw, reads.p, wflags.p, R.h[6], R.prep[6]->d);
  }
  if (n_rec_plain) {
    R.h[4] = padded_h(std::max<size_t>((n_rec_plain + rl - 1) / rl, 1), mh);
    R.prep[4] = zero_mat(ctx, R.h[4], 2 * (size_t)rl);
    LAUNCH(k_prep_simple<PP>, n_rec_plain, ops.p, rec_ops.p, n_rec_plain, rl, 1, reads.p, wflags.p, D, R.h[4], R.prep[4]->d);
  }
  if (n_rec_coeff) {
    const int slot = R.recompose_coeff ? 4 : 5;
    R.h[slot] = padded_h(std::max<size_t>((n_rec_coeff + rl - 1) / rl, 1), mh);
    R.prep[slot] = zero_mat(ctx, R.h[slot], (2 + 2 * (size_t)D) * rl);
    LAUNCH(k_prep_rec_coeff<PP>, n_rec_coeff, ops.p, ext.p, rec_coeff_ops.p, n_rec_coeff, rl, reads.p, wflags.p, D, R.h[slot],
           R.prep[slot]->d);
  }
  {
    // ALU lane schedule
    const size_t n = n_alu_ops;
    size_t rows = 1, chain_rows = 0;
    bool any = false;
    DevBuf nonchain(n + 2), chain_start(n + 2), heads(n + 2), nonchain_rank(n + 2), chain_rank(n + 2), heads_rank(n + 2);
    if (n) {
      LAUNCH(k_alu_marks, n, ops.p, alu_ops.p, n, k_max, nonchain.p, chain_start.p, heads.p);
      P3R_HIP(hipMemsetAsync(nonchain.p + n, 0, 4, s));
      P3R_HIP(hipMemsetAsync(chain_start.p + n, 0, 4, s));
      P3R_HIP(hipMemsetAsync(heads.p + n, 0, 4, s));
      scan_u32(ctx, nonchain.p, nonchain_rank.p, n + 1);
      scan_u32(ctx, chain_start.p, chain_rank.p, n + 1);
      scan_u32(ctx, heads.p, heads_rank.p, n + 1);
      uint32_t t3[3];
      P3R_HIP(hipMemcpyAsync(&t3[0], nonchain_rank.p + n, 4, hipMemcpyDeviceToHost, s));
      P3R_HIP(hipMemcpyAsync(&t3[1], chain_rank.p + n, 4, hipMemcpyDeviceToHost, s));
      P3R_HIP(copy_sync(s, &t3[2], heads_rank.p + n, 4, hipMemcpyDeviceToHost));
      const size_t n_nc = t3[0], n_chains = t3[1], n_groups = t3[2];
      any = n_chains > 0;
      if (any) {
        chain_rows = 1 + n_groups + (n_chains - 1);
        const size_t fill = chain_rows * (size_t)(lanes - 1);
        rows = chain_rows + (n_nc > fill ? (n_nc - fill + lanes - 1) / lanes : 0);
      } else {
        rows = std::max<size_t>((n + lanes - 1) / lanes, 1);
      }
    }
    R.alu_rows = rows;
    R.h[2] = padded_h(rows, mh);
    const int pw = lanes * 13 + 7 * (k_max - 1);
    R.prep[2] = zero_mat(ctx, R.h[2], (size_t)pw);
    R.alu_plan.alloc(rows * lanes * (sizeof(AluPlanEntry) / 4));
    AluPlanEntry* plan = reinterpret_cast<AluPlanEntry*>(R.alu_plan.p);
    LAUNCH(k_plan_init, rows * lanes, rows * lanes, plan);
    if (n == 0) {
      // the dummy row of an empty ALU table (common.rs:283-286): one op, zero cells
      hipLaunchKernelGGL(k_alu_plan_flat, dim3(1), dim3(kB), 0, s, (size_t)1, plan);
    } else if (!any) {
      LAUNCH(k_alu_plan_flat, n, n, plan);
    } else {
      LAUNCH(k_alu_plan, n, ops.p, alu_ops.p, n, lanes, k_max, nonchain_rank.p, chain_start.p, chain_rank.p, heads.p, heads_rank.p, chain_rows,
             plan);
    }
    R.alu_prev_src.alloc(rows);
    LAUNCH(k_alu_prev, rows, plan, rows, lanes, any ? 1 : 0, R.alu_prev_src.p);
    if (n) LAUNCH(k_prep_alu<PP>, rows * lanes, ops.p, alu_ops.p, roles.p, reads.p, plan, rows, lanes, k_max, D, R.h[2], R.prep[2]->d);
  }

  prof_stage(ctx, "prep_schedule_static");
  // ---- execution schedule -------------------------------------------------------------------------------------
  RunSchedule& S = R.sched;
  DevBuf oflags(n_ops), olevel(n_ops);
  P3R_HIP(hipMemsetAsync(olevel.p, 0, n_ops * 4, s));
  LAUNCH(k_sched_static, n_ops, ops.p, n_ops, ext.p, stime.p, sh, oflags.p, bad.p);
  LAUNCH(k_chain_link, n_ops, ops.p, n_ops, oflags.p);
  DevBuf ready_rank(n_ops + 1), light_rank(n_ops + 1);
  scan_flags(ctx, FlagReady{oflags.p}, n_ops, ready_rank.p);
  scan_flags(ctx, FlagLight{oflags.p}, n_ops, light_rank.p);
  hipLaunchKernelGGL(k_flags_total<FlagReady>, dim3(1), dim3(64), 0, s, FlagReady{oflags.p}, n_ops, ready_rank.p);
  hipLaunchKernelGGL(k_flags_total<FlagLight>, dim3(1), dim3(64), 0, s, FlagLight{oflags.p}, n_ops, light_rank.p);
  // NonPrimitiveOpId -> row
  {
    DevBuf mx(1);
    prims::reduce_max(s, MaxNpoId{ops.p}, n_ops, mx.p);
    uint32_t t3[3];
    P3R_HIP(hipMemcpyAsync(&t3[0], mx.p, 4, hipMemcpyDeviceToHost, s));
    P3R_HIP(hipMemcpyAsync(&t3[1], ready_rank.p + n_ops, 4, hipMemcpyDeviceToHost, s));
    P3R_HIP(copy_sync(s, &t3[2], light_rank.p + n_ops, 4, hipMemcpyDeviceToHost));
    R.n_op_ids = t3[0];
    S.n_alu_records = (uint32_t)n_alu_ops;
    // (sizes of the compacted lists)
    S.light.clear();
    R.n_rewrite = 0;
    const size_t n_members = t3[1], n_light = t3[2];
    R.d_row_of_op_id.alloc(std::max<size_t>(R.n_op_ids, 1));
    P3R_HIP(hipMemsetAsync(R.d_row_of_op_id.p, 0xFF, std::max<size_t>(R.n_op_ids, 1) * 4, s));
    LAUNCH(k_op_ids, n_p2, ops.p, p2_ops.p, n_p2, R.n_op_ids, R.d_row_of_op_id.p, bad.p);
    if (n_pw) {
      R.d_roww_of_op_id.alloc(std::max<size_t>(R.n_op_ids, 1));
      P3R_HIP(hipMemsetAsync(R.d_roww_of_op_id.p, 0xFF, std::max<size_t>(R.n_op_ids, 1) * 4, s));
      LAUNCH(k_op_ids, n_pw, ops.p, pw_ops.p, n_pw, R.n_op_ids, R.d_roww_of_op_id.p, bad.p);
      LAUNCH(k_ids_cross, R.n_op_ids, R.d_row_of_op_id.p, R.d_roww_of_op_id.p, R.n_op_ids, bad.p);
    }

    DevBuf hmem(std::max<size_t>(n_members, 1)), run_start(n_members + 2), light_ops(std::max<size_t>(n_light, 1));
    LAUNCH(k_compact_ops, n_ops, oflags.p, n_ops, ready_rank.p, light_rank.p, hmem.p, run_start.p, light_ops.p);
    // Horner runs
    DevBuf run_rank(n_members + 2), runs(std::max<size_t>(n_members, 1)), chead(std::max<size_t>(n_members, 1) + 1);
    size_t n_runs = 0;
    if (n_members) {
      P3R_HIP(hipMemsetAsync(run_start.p + n_members, 0, 4, s));
      scan_u32(ctx, run_start.p, run_rank.p, n_members + 1);
      uint32_t v = 0;
      P3R_HIP(copy_sync(s, &v, run_rank.p + n_members, 4, hipMemcpyDeviceToHost));
      n_runs = v;
      LAUNCH(k_compact_marked, n_members, run_start.p, run_rank.p, n_members, runs.p);
      P3R_HIP(hipMemsetAsync(chead.p, 0, (n_members + 1) * 4, s));
    }
    // Poseidon2 mode lists and runs
    DevBuf is_merkle(n_p2 + 2), merkle_rank(n_p2 + 2), mlist(std::max<size_t>(n_p2, 1)), mpos(std::max<size_t>(n_p2, 1)),
        p2_run_start(n_p2 + 2), p2_run_rank(n_p2 + 2), p2_runs(std::max<size_t>(n_p2, 1)), plevel(std::max<size_t>(n_p2, 1)),
        phead(n_p2 + 2);
    size_t n_p2_runs = 0, n_normal = 0;
    if (n_p2) {
      LAUNCH(k_p2_modes, n_p2, ops.p, p2_ops.p, n_p2, is_merkle.p);
      P3R_HIP(hipMemsetAsync(is_merkle.p + n_p2, 0, 4, s));
      scan_u32(ctx, is_merkle.p, merkle_rank.p, n_p2 + 1);
      LAUNCH(k_p2_lists, n_p2, ops.p, p2_ops.p, n_p2, merkle_rank.p, mlist.p, mpos.p, p2_run_start.p, bad.p);
      P3R_HIP(hipMemsetAsync(p2_run_start.p + n_p2, 0, 4, s));
      scan_u32(ctx, p2_run_start.p, p2_run_rank.p, n_p2 + 1);
      uint32_t v2[2];
      P3R_HIP(hipMemcpyAsync(&v2[0], p2_run_rank.p + n_p2, 4, hipMemcpyDeviceToHost, s));
      P3R_HIP(copy_sync(s, &v2[1], merkle_rank.p + n_p2, 4, hipMemcpyDeviceToHost));
      n_p2_runs = v2[0];
      n_normal = n_p2 - v2[1];
      LAUNCH(k_compact_marked, n_p2, p2_run_start.p, p2_run_rank.p, n_p2, p2_runs.p);
      P3R_HIP(hipMemsetAsync(plevel.p, 0, n_p2 * 4, s));
      P3R_HIP(hipMemsetAsync(phead.p, 0, (n_p2 + 1) * 4, s));
    }
    // rows of the width-32 table: predecessor links
    DevBuf pw_sponge(n_pw + 2), pw_srank(n_pw + 2), pw_slist(std::max<size_t>(n_pw, 1)), pw_prev(std::max<size_t>(n_pw, 1)),
        pw_level(std::max<size_t>(n_pw, 1)), pw_joined(std::max<size_t>(n_pw, 1));
    size_t n_pw_sponge = 0;
    if (n_pw) {
      LAUNCH(k_pw_modes, n_pw, ops.p, pw_ops.p, n_pw, pw_sponge.p);
      P3R_HIP(hipMemsetAsync(pw_sponge.p + n_pw, 0, 4, s));
      scan_u32(ctx, pw_sponge.p, pw_srank.p, n_pw + 1);
      uint32_t v = 0;
      P3R_HIP(copy_sync(s, &v, pw_srank.p + n_pw, 4, hipMemcpyDeviceToHost));
      n_pw_sponge = v;
      LAUNCH(k_compact_marked, n_pw, pw_sponge.p, pw_srank.p, n_pw, pw_slist.p);
      LAUNCH(k_pw_links, n_pw, ops.p, pw_ops.p, n_pw, pw_srank.p, pw_slist.p, pw_prev.p, bad.p);
      P3R_HIP(hipMemsetAsync(pw_level.p, 0, n_pw * 4, s));
      P3R_HIP(hipMemsetAsync(pw_joined.p, 0, n_pw * 4, s));
    }
    {
      uint32_t b0 = 0;
      P3R_HIP(copy_sync(s, &b0, bad.p, 4, hipMemcpyDeviceToHost));
      if (b0) return false;
    }
    prof_stage(ctx, "prep_schedule_levels");
    // levels
    for (int round = 0;; ++round) {
      P3R_HIP(hipMemsetAsync(changed.p, 0, 4, s));
      for (int k = 0; k < 4; ++k) {
        LAUNCH(k_level_light, n_light, ops.p, ext.p, light_ops.p, n_light, oflags.p, stime.p, D, wlevel.p, olevel.p, changed.p);
        LAUNCH(k_level_chains, n_runs * 64, ops.p, hmem.p, runs.p, n_runs, n_members, wlevel.p, olevel.p, chead.p, changed.p);
        LAUNCH(k_level_p2, n_p2_runs * 64, ops.p, ext.p, p2_ops.p, mlist.p, p2_runs.p, n_p2_runs, n_p2, stime.p, sh, wlevel.p, plevel.p,
               phead.p, changed.p);
        LAUNCH(k_level_p2w, n_pw, ops.p, ext.p, pw_ops.p, n_pw, pw_prev.p, pw_sponge.p, stime.p, wlevel.p, pw_level.p, pw_joined.p,
               changed.p);
      }
      uint32_t c = 0;
      P3R_HIP(copy_sync(s, &c, changed.p, 4, hipMemcpyDeviceToHost));
      if (!c) break;
      if (round > (1 << 20)) fail(P3R_EINVAL, "circuit schedule did not converge");
    }
    prof_stage(ctx, "prep_schedule_emit");
    // ALU-dedup leftovers (runner.rs:199-216): duplicates take the value of their canonical witness after the last level
    std::vector<uint32_t> rewrite_triples;
    if (d->n_rewrite) {
      std::unordered_map<uint32_t, uint32_t> canon_of;
      for (size_t k = 0; k < d->n_rewrite; ++k) canon_of.emplace(d->witness_rewrite[2 * k], d->witness_rewrite[2 * k + 1]);
      std::vector<uint32_t> dup(d->n_rewrite), cur(d->n_rewrite), need;
      for (size_t k = 0; k < d->n_rewrite; ++k) {
        dup[k] = d->witness_rewrite[2 * k];
        uint32_t c = d->witness_rewrite[2 * k + 1];
        for (size_t hops = 0; hops <= canon_of.size(); ++hops) {
          auto it = canon_of.find(c);
          if (it == canon_of.end()) break;
          c = it->second;
        }
        cur[k] = c;
        need.push_back(dup[k]);
        need.push_back(c);
      }
      DevBuf d_need(need.size()), d_got(need.size());
      P3R_HIP(hipMemcpyAsync(d_need.p, need.data(), need.size() * 4, hipMemcpyHostToDevice, s));
      LAUNCH(k_gather_u32, need.size(), stime.p, d_need.p, need.size(), d_got.p);
      std::vector<uint32_t> got = fetch<uint32_t>(ctx, d_got.p, need.size());
      std::unordered_map<uint32_t, bool> set;
      for (size_t k = 0; k < need.size(); ++k) set[need[k]] = got[k] != kUnset;
      std::vector<uint32_t> now_set;
      for (size_t k = 0; k < d->n_rewrite; ++k) {
        if (!set[cur[k]]) continue;
        rewrite_triples.insert(rewrite_triples.end(), {dup[k], cur[k], (uint32_t)set[dup[k]]});
        if (!set[dup[k]]) now_set.push_back(dup[k]);
        set[dup[k]] = true;
      }
      if (!now_set.empty()) {
        DevBuf d_now(now_set.size());
        P3R_HIP(hipMemcpyAsync(d_now.p, now_set.data(), now_set.size() * 4, hipMemcpyHostToDevice, s));
        LAUNCH(k_set_zero_at, now_set.size(), d_now.p, now_set.size(), stime.p);
        P3R_HIP(hipStreamSynchronize(s));
      }
    }
    R.n_rewrite = rewrite_triples.size() / 3;
    up(R.d_rewrite, rewrite_triples.data(), rewrite_triples.size());
    LAUNCH(k_any_unset, nw, stime.p, nw, bad.p);  // WitnessNotSetForIndex

    prof_stage(ctx, "prep_emit_light");
    // ---- emission: level histograms and the level-sorted arrays
    DevBuf mxl(1);
    P3R_HIP(hipMemsetAsync(mxl.p, 0, 4, s));
    LAUNCH(k_max_u32, n_ops, olevel.p, n_ops, mxl.p);
    LAUNCH(k_max_u32, n_p2, plevel.p, n_p2, mxl.p);
    LAUNCH(k_max_u32, n_pw, pw_level.p, n_pw, mxl.p);
    uint32_t max_level = 0;
    {
      uint32_t v2[2];
      P3R_HIP(hipMemcpyAsync(&v2[0], mxl.p, 4, hipMemcpyDeviceToHost, s));
      P3R_HIP(copy_sync(s, &v2[1], bad.p, 4, hipMemcpyDeviceToHost));
      if (v2[1]) return false;
      max_level = v2[0];
    }
    S.levels = max_level;
    const size_t nl1 = (size_t)max_level + 2;
    const int lbits = bits_for(2 * max_level + 1);
    // light ops: stable sort by level
    DevBuf light_keys(std::max<size_t>(n_light, 1)), light_keys2(std::max<size_t>(n_light, 1)), light_sorted(std::max<size_t>(n_light, 1));
    // first[key] per sorted array: light ops and segments by level, chains by level * 2 + (short ? 1 : 0)
    DevBuf hist(4 * nl1);
    P3R_HIP(hipMemsetAsync(hist.p, 0xFF, 4 * nl1 * 4, s));
    uint32_t *f_light = hist.p, *f_seg = hist.p + nl1, *f_chain = hist.p + 2 * nl1;
    LAUNCH(k_gather_u32, n_light, olevel.p, light_ops.p, n_light, light_keys.p);
    sort_by_key(ctx, light_keys.p, light_keys2.p, light_ops.p, light_sorted.p, n_light, lbits);
    LAUNCH(k_first_index, n_light, light_keys2.p, n_light, f_light);
    R.d_light.alloc(std::max<size_t>(n_light * (sizeof(RunOp) / 4), 1));
    LAUNCH(k_emit_light, n_light, ops.p, light_sorted.p, n_light, oflags.p, S_ap, S_re, s_rc.p, (uint32_t)n_rec_plain,
           reinterpret_cast<RunOp*>(R.d_light.p));
    R.d_ext.alloc(std::max<size_t>(n_dev_ext, 1));
    LAUNCH(k_emit_ext<PP>, n_ops, ops.p, n_ops, ext.p, S_re, stime.p, D, R.d_ext.p);
    prof_stage(ctx, "prep_emit_chains");
    // Horner chains
    size_t n_chains = 0;
    R.d_chain_ops.alloc(std::max<size_t>(n_members * (sizeof(RunOp) / 4), 1));
    R.d_chains.alloc(1);
    if (n_members) {
      LAUNCH(k_emit_chain_ops, n_members, ops.p, hmem.p, n_members, S_ap, S_re, reinterpret_cast<RunOp*>(R.d_chain_ops.p));
      DevBuf ch_rank(n_members + 2), ch_heads(n_members);
      scan_u32(ctx, chead.p, ch_rank.p, n_members + 1);
      uint32_t v = 0;
      P3R_HIP(copy_sync(s, &v, ch_rank.p + n_members, 4, hipMemcpyDeviceToHost));
      n_chains = v;
      LAUNCH(k_compact_marked, n_members, chead.p, ch_rank.p, n_members, ch_heads.p);
      DevBuf segs(n_chains * 4), keys(n_chains), keys2(n_chains), lv(n_chains), iota(n_chains), order(n_chains);
      LAUNCH(k_chain_records, n_chains, ops.p, hmem.p, ch_heads.p, n_chains, n_members, olevel.p,
             reinterpret_cast<RunSchedule::ChainSeg*>(segs.p), keys.p, lv.p);
      LAUNCH(k_iota, n_chains, iota.p, n_chains);
      sort_by_key(ctx, keys.p, keys2.p, iota.p, order.p, n_chains, lbits);
      LAUNCH(k_first_index, n_chains, keys2.p, n_chains, f_chain);
      R.d_chains.alloc(n_chains * 4);
      LAUNCH(k_permute_chains, n_chains, reinterpret_cast<RunSchedule::ChainSeg*>(segs.p), order.p, n_chains,
             reinterpret_cast<RunSchedule::ChainSeg*>(R.d_chains.p));
    }
    prof_stage(ctx, "prep_emit_p2");
    // Poseidon2 segments
    size_t n_segs = 0;
    R.d_p2.alloc(std::max<size_t>(n_p2 * ((D == 4 ? sizeof(RunP2) : sizeof(RunP2B)) / 4), 1));
    R.d_p2segs.alloc(1);
    if (n_p2) {
      DevBuf head_by_row(n_p2 + 2), head_rank(n_p2 + 2), head_rows(n_p2);
      LAUNCH(k_p2_head_by_row, n_p2, mlist.p, phead.p, n_p2, head_by_row.p);
      P3R_HIP(hipMemsetAsync(head_by_row.p + n_p2, 0, 4, s));
      scan_u32(ctx, head_by_row.p, head_rank.p, n_p2 + 1);
      uint32_t v = 0;
      P3R_HIP(copy_sync(s, &v, head_rank.p + n_p2, 4, hipMemcpyDeviceToHost));
      n_segs = v;
      LAUNCH(k_compact_marked, n_p2, head_by_row.p, head_rank.p, n_p2, head_rows.p);
      DevBuf seg_pos(n_segs), seg_len(n_segs), seg_level(n_segs), keys2(n_segs), iota(n_segs), order(n_segs), len_sorted(n_segs + 1),
          first(n_segs + 1);
      LAUNCH(k_p2_seg_records, n_segs, head_rows.p, n_segs, mpos.p, phead.p, plevel.p, n_p2, n_normal, seg_pos.p, seg_len.p, seg_level.p);
      LAUNCH(k_iota, n_segs, iota.p, n_segs);
      sort_by_key(ctx, seg_level.p, keys2.p, iota.p, order.p, n_segs, lbits);
      LAUNCH(k_first_index, n_segs, keys2.p, n_segs, f_seg);
      LAUNCH(k_gather_u32, n_segs, seg_len.p, order.p, n_segs, len_sorted.p);
      scan_u32(ctx, len_sorted.p, first.p, n_segs);
      R.d_p2segs.alloc(n_segs * 2);
      if (D == 4)
        LAUNCH(k_emit_p2, n_segs, ops.p, ext.p, p2_ops.p, mlist.p, order.p, seg_pos.p, seg_len.p, first.p, n_segs, n_normal, stime.p,
               reinterpret_cast<RunP2*>(R.d_p2.p), reinterpret_cast<RunSchedule::P2Seg*>(R.d_p2segs.p));
      else
        LAUNCH(k_emit_p2_base, n_segs, ops.p, ext.p, p2_ops.p, mlist.p, order.p, seg_pos.p, seg_len.p, first.p, n_segs, n_normal, stime.p,
               reinterpret_cast<RunP2B*>(R.d_p2.p), reinterpret_cast<RunSchedule::P2Seg*>(R.d_p2segs.p));
    }
    // segments of the width-32 table: heads in row order (= the order the sequential walk opens them in), stable sort by level
    size_t n_wsegs = 0;
    DevBuf histw(nl1);
    P3R_HIP(hipMemsetAsync(histw.p, 0xFF, nl1 * 4, s));
    if (n_pw) {
      DevBuf is_head(n_pw + 2), head_rank(n_pw + 2), head_rows(n_pw);
      LAUNCH(k_pw_heads, n_pw, pw_joined.p, n_pw, is_head.p);
      P3R_HIP(hipMemsetAsync(is_head.p + n_pw, 0, 4, s));
      scan_u32(ctx, is_head.p, head_rank.p, n_pw + 1);
      uint32_t v = 0;
      P3R_HIP(copy_sync(s, &v, head_rank.p + n_pw, 4, hipMemcpyDeviceToHost));
      n_wsegs = v;
      LAUNCH(k_compact_marked, n_pw, is_head.p, head_rank.p, n_pw, head_rows.p);
      DevBuf seg_len(n_wsegs), seg_level(n_wsegs), keys2(n_wsegs), iota(n_wsegs), order(n_wsegs), len_sorted(n_wsegs + 1), first(n_wsegs + 1);
      LAUNCH(k_pw_seg_records, n_wsegs, head_rows.p, n_wsegs, n_pw, pw_prev.p, pw_joined.p, pw_sponge.p, pw_srank.p, pw_slist.p, n_pw_sponge,
             pw_level.p, seg_len.p, seg_level.p);
      LAUNCH(k_iota, n_wsegs, iota.p, n_wsegs);
      sort_by_key(ctx, seg_level.p, keys2.p, iota.p, order.p, n_wsegs, lbits);
      LAUNCH(k_first_index, n_wsegs, keys2.p, n_wsegs, histw.p);
      LAUNCH(k_gather_u32, n_wsegs, seg_len.p, order.p, n_wsegs, len_sorted.p);
      scan_u32(ctx, len_sorted.p, first.p, n_wsegs);
      R.d_p2w.alloc(n_pw * (sizeof(RunP2W) / 4));
      R.d_p2wsegs.alloc(n_wsegs * 2);
      LAUNCH(k_emit_p2w, n_wsegs, ops.p, ext.p, pw_ops.p, n_pw, pw_prev.p, pw_joined.p, pw_sponge.p, pw_srank.p, pw_slist.p, n_pw_sponge,
             order.p, head_rows.p, seg_len.p, first.p, n_wsegs, stime.p, reinterpret_cast<RunP2W*>(R.d_p2w.p),
             reinterpret_cast<RunSchedule::P2Seg*>(R.d_p2wsegs.p));
    }
    prof_stage(ctx, "prep_emit_rest");
    // static Const trace, Public gather list
    R.d_const_values.alloc(std::max<size_t>(n_const * D, 1));
    LAUNCH(k_const_values<PP>, n_const * D, ops.p, ext.p, const_ops.p, n_const, D, R.d_const_values.p);
    R.d_public_out.alloc(std::max<size_t>(n_public, 1));
    LAUNCH(k_outs_of, n_public, ops.p, public_ops.p, n_public, R.d_public_out.p);
    // per-level offsets on the host, launch plan
    std::vector<uint32_t> hh = fetch<uint32_t>(ctx, hist.p, 4 * nl1);
    // offsets[k] = first position with key >= k: a key that does not occur starts where the next one does
    auto offsets = [&](const uint32_t* first, size_t n_keys, size_t n_items) {
      std::vector<uint32_t> off(n_keys + 1, (uint32_t)n_items);
      for (size_t k = n_keys; k-- > 0;) off[k] = first[k] != kUnset ? first[k] : off[k + 1];
      return off;
    };
    S.light_off = offsets(hh.data(), nl1 - 1, n_light);            // levels 0 .. max_level, then the total
    S.p2seg_off = offsets(hh.data() + nl1, nl1 - 1, n_segs);
    {
      const std::vector<uint32_t> co = offsets(hh.data() + 2 * nl1, 2 * (nl1 - 1), n_chains);  // keys 2l (long), 2l + 1 (short)
      S.chain_off.assign(nl1, 0);
      S.chain_long.assign(nl1, 0);
      for (size_t l = 0; l + 1 < nl1; ++l) { S.chain_off[l] = co[2 * l]; S.chain_long[l] = co[2 * l + 1] - co[2 * l]; }
      S.chain_off[nl1 - 1] = (uint32_t)n_chains;
    }
    if (S.light_off.back() != n_light || S.p2seg_off.back() != n_segs || S.chain_off.back() != n_chains)
      fail(P3R_EINVAL, "circuit schedule: level histogram does not add up (%u/%zu light, %u/%zu segments, %u/%zu chains)",
           S.light_off.back(), n_light, S.p2seg_off.back(), n_segs, S.chain_off.back(), n_chains);
    if (n_pw) {
      const std::vector<uint32_t> hw = fetch<uint32_t>(ctx, histw.p, nl1);
      S.p2wseg_off = offsets(hw.data(), nl1 - 1, n_wsegs);
      if (S.p2wseg_off.back() != n_wsegs)
        fail(P3R_EINVAL, "circuit schedule: level histogram does not add up (%u/%zu width-32 segments)", S.p2wseg_off.back(), n_wsegs);
      up(R.d_p2wseg_off, S.p2wseg_off.data(), S.p2wseg_off.size());
    }
    finish_segments(S);
    up(R.d_light_off, S.light_off.data(), S.light_off.size());
    up(R.d_p2seg_off, S.p2seg_off.data(), S.p2seg_off.size());
    up(R.d_chunk_bounds, S.chunk_bounds.data(), S.chunk_bounds.size());
  }
  P3R_HIP(hipGetLastError());
  P3R_HIP(hipStreamSynchronize(s));
  return true;
}

}  // namespace

bool devprep_circuit(p3r_ctx* ctx, const p3r_circuit_desc* d, DevPrep& out) {
  return ctx->cfg.field == P3R_FIELD_KOALA_BEAR ? devprep_impl<KoalaBearParams>(ctx, d, out) : devprep_impl<BabyBearParams>(ctx, d, out);
}

}  // namespace p3r
