// Issue rates of single VALU opcodes on gfx950, one opcode per kernel, written as inline assembly so that the compiler
// cannot substitute or fuse anything: what tools/valu_mix.py prices the instruction mix of the integer kernel families
// with (bench.py: valu_families).  Sixteen independent destination registers per lane, 4096 iterations x 16 = 65536
// instructions per lane and launch; 2048 workgroups of 256 lanes.  Output: "<opcode> <T lane-ops/s>" per line.
//   hipcc --offload-arch=gfx950 -O3 -o valu_classes valu_classes.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 4096, UNROLL = 16;

#define REP16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

enum { ADD, SUB, MIN, XOR, AND, LSHR, CNDMASK, MOV, ADD_CO, ADDC_CO, ADD3, LSHL_ADD, AND_OR, BFE, ALIGNBIT, PERM, MUL_LO, MUL_HI,
       MAD64, MUL_U24, MAD_U24, MOV_DPP_QUAD, ADD_DPP_ROR, CMP, FMA_F64, ADD_F64, MUL_F64, FMA_F32, N_OPS };

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
  uint32_t a[UNROLL];
  uint64_t q[UNROLL];
  double d[UNROLL];
  uint32_t b = (seed | 1u) + threadIdx.x, c = seed * 2654435761u + 7u;
  double e = 1.0000001, f = 0.5;
  for (int i = 0; i < UNROLL; ++i) { a[i] = threadIdx.x * 2654435761u + i + seed; q[i] = ((uint64_t)a[i] << 32) | i; d[i] = (double)a[i]; }
  asm volatile("s_mov_b32 vcc_lo, 0x55555555\n\ts_mov_b32 vcc_hi, 0x55555555" ::: "vcc");
  for (int it = 0; it < ITER; ++it) {
#define A1(i, INSN) asm volatile(INSN " %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define A2(i, INSN) asm volatile(INSN " %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    if (OP == ADD) {
#define M(i) A1(i, "v_add_u32")
      REP16(M)
#undef M
    }
    if (OP == SUB) {
#define M(i) A1(i, "v_sub_u32")
      REP16(M)
#undef M
    }
    if (OP == MIN) {
#define M(i) A1(i, "v_min_u32")
      REP16(M)
#undef M
    }
    if (OP == XOR) {
#define M(i) A1(i, "v_xor_b32")
      REP16(M)
#undef M
    }
    if (OP == AND) {
#define M(i) A1(i, "v_and_b32")
      REP16(M)
#undef M
    }
    if (OP == LSHR) {
#define M(i) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[i]));
      REP16(M)
#undef M
    }
    if (OP == CNDMASK) {
#define M(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
      REP16(M)
#undef M
    }
    if (OP == MOV) {
#define M(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
      REP16(M)
#undef M
    }
    if (OP == ADD_CO) {
#define M(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc");
      REP16(M)
#undef M
    }
    if (OP == ADDC_CO) {
#define M(i) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
      REP16(M)
#undef M
    }
    if (OP == ADD3) {
#define M(i) A2(i, "v_add3_u32")
      REP16(M)
#undef M
    }
    if (OP == LSHL_ADD) {
#define M(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
      REP16(M)
#undef M
    }
    if (OP == AND_OR) {
#define M(i) A2(i, "v_and_or_b32")
      REP16(M)
#undef M
    }
    if (OP == BFE) {
#define M(i) asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(a[i]));
      REP16(M)
#undef M
    }
    if (OP == ALIGNBIT) {
#define M(i) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b));
      REP16(M)
#undef M
    }
    if (OP == PERM) {
#define M(i) A2(i, "v_perm_b32")
      REP16(M)
#undef M
    }
    if (OP == MUL_LO) {
#define M(i) A1(i, "v_mul_lo_u32")
      REP16(M)
#undef M
    }
    if (OP == MUL_HI) {
#define M(i) asm volatile("v_mul_hi_u32 %0, %0, %1\n\tv_or_b32 %0, 1, %0" : "+v"(a[i]) : "v"(b));   /* (the or keeps the value alive: counted below) */
      REP16(M)
#undef M
    }
    if (OP == MAD64) {
#define M(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(b), "v"(c) : "vcc");
      REP16(M)
#undef M
    }
    if (OP == MUL_U24) {
#define M(i) A1(i, "v_mul_u32_u24")
      REP16(M)
#undef M
    }
    if (OP == MAD_U24) {
#define M(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP16(M)
#undef M
    }
    if (OP == MOV_DPP_QUAD) {
#define M(i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      REP16(M)
#undef M
    }
    if (OP == ADD_DPP_ROR) {
#define M(i) asm volatile("v_add_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      REP16(M)
#undef M
    }
    if (OP == CMP) {
#define M(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a[i]), "v"(b) : "vcc");
      REP16(M)
#undef M
    }
    if (OP == FMA_F64) {
#define M(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(e), "v"(f));
      REP16(M)
#undef M
    }
    if (OP == ADD_F64) {
#define M(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(f));
      REP16(M)
#undef M
    }
    if (OP == MUL_F64) {
#define M(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e));
      REP16(M)
#undef M
    }
    if (OP == FMA_F32) {
#define M(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP16(M)
#undef M
    }
  }
  uint32_t s = 0;
  for (int i = 0; i < UNROLL; ++i) s ^= a[i] ^ (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32) ^ (uint32_t)d[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(uint32_t* out, const char* name, double insts_per_slot = 1.0) {
  const int blocks = 256 * 8;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 12345u);
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 12345u + r);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double total = 5.0 * blocks * 256 * (double)ITER * UNROLL * insts_per_slot;
  printf("%-24s %8.2f\n", name, total / (ms * 1e-3) / 1e12);
}

int main() {
  uint32_t* out;
  CK(hipMalloc(&out, 256 * 8 * 256 * 4));
  printf("# opcode                  T lane-instructions/s (MI355X, 2048 x 256 lanes, 65536 instructions per lane)\n");
  run<ADD>(out, "v_add_u32"); run<SUB>(out, "v_sub_u32"); run<MIN>(out, "v_min_u32"); run<XOR>(out, "v_xor_b32");
  run<AND>(out, "v_and_b32"); run<LSHR>(out, "v_lshrrev_b32"); run<CNDMASK>(out, "v_cndmask_b32"); run<MOV>(out, "v_mov_b32");
  run<ADD_CO>(out, "v_add_co_u32"); run<ADDC_CO>(out, "v_addc_co_u32"); run<ADD3>(out, "v_add3_u32");
  run<LSHL_ADD>(out, "v_lshl_add_u32"); run<AND_OR>(out, "v_and_or_b32"); run<BFE>(out, "v_bfe_u32");
  run<ALIGNBIT>(out, "v_alignbit_b32"); run<PERM>(out, "v_perm_b32"); run<MUL_LO>(out, "v_mul_lo_u32");
  run<MUL_HI>(out, "v_mul_hi_u32+v_or_b32", 2.0); run<MAD64>(out, "v_mad_u64_u32"); run<MUL_U24>(out, "v_mul_u32_u24");
  run<MAD_U24>(out, "v_mad_u32_u24"); run<MOV_DPP_QUAD>(out, "v_mov_b32_dpp"); run<ADD_DPP_ROR>(out, "v_add_u32_dpp");
  run<CMP>(out, "v_cmp_lt_u32"); run<FMA_F64>(out, "v_fma_f64"); run<ADD_F64>(out, "v_add_f64"); run<MUL_F64>(out, "v_mul_f64");
  run<FMA_F32>(out, "v_fma_f32");
  return 0;
}
