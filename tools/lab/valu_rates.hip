// tools/lab/valu_rates.hip -- what gfx950 charges per VALU instruction kind, per wave, at 1..8 waves per SIMD.
// Each kernel runs ITERS x 32 instructions of one kind on 8 independent register chains between two s_memtime stamps;
// output = shader cycles per instruction seen by ONE wave, and (x waves per SIMD) the SIMD's cycles per instruction.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/lab/bin/valu_rates tools/lab/valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define I8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define B32(op) I8(op) I8(op) I8(op) I8(op)

#define OP_ADD(r) "v_add_f32 %" #r ", %" #r ", %8\n"
#define OP_FMA(r) "v_fma_f32 %" #r ", %" #r ", %8, %9\n"
#define OP_ADDDEP(r) "v_add_f32 %0, %0, %8\n"
#define OP_PKFMA(r) "v_pk_fma_f32 %" #r ", %" #r ", %10, %11\n"
#define OP_PKMUL(r) "v_pk_mul_f32 %" #r ", %" #r ", %10\n"
#define OP_PKADD(r) "v_pk_add_f32 %" #r ", %" #r ", %10\n"
#define OP_DPP_WSHR(r) "v_add_f32_dpp %" #r ", %" #r ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define OP_DPP_WSHL(r) "v_add_f32_dpp %" #r ", %" #r ", %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define OP_DPP_RSHR(r) "v_add_f32_dpp %" #r ", %" #r ", %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define OP_DPP_QP(r) "v_add_f32_dpp %" #r ", %" #r ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define OP_MOV_DPP_WSHR(r) "v_mov_b32_dpp %" #r ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define OP_RCP(r) "v_rcp_f32 %" #r ", %" #r "\n"
#define OP_FLOOR(r) "v_floor_f32 %" #r ", %" #r "\n"
#define OP_CVT(r) "v_cvt_i32_f32 %" #r ", %" #r "\n"
#define OP_MAX3(r) "v_max3_f32 %" #r ", %" #r ", %8, %9\n"
#define OP_MED3(r) "v_med3_f32 %" #r ", %" #r ", %8, %9\n"
#define OP_CNDMASK(r) "v_cndmask_b32 %" #r ", %" #r ", %8, vcc\n"
#define OP_CMP(r) "v_cmp_lt_f32 vcc, %" #r ", %8\n"
#define OP_MADU24(r) "v_mad_u32_u24 %" #r ", %" #r ", %8, %9\n"
#define OP_MADI32(r) "v_mad_i32_i24 %" #r ", %" #r ", %8, %9\n"
#define OP_MULLO(r) "v_mul_lo_u32 %" #r ", %" #r ", %8\n"
#define OP_LSHLADD(r) "v_lshl_add_u32 %" #r ", %" #r ", 2, %8\n"
#define OP_ADDU(r) "v_add_u32 %" #r ", %" #r ", %8\n"
#define OP_SUBABS(r) "v_sub_f32 %" #r ", %" #r ", |%8|\n"
#define OP_PERM32(r) "v_permlane32_swap_b32 %" #r ", %8\n"
#define OP_SALU(r) "s_add_u32 s20, s20, s21\n"
#define OP_MUL(r) "v_mul_f32 %" #r ", %" #r ", %8\n"
#define OP_FMAC(r) "v_fmac_f32 %" #r ", %8, %9\n"
#define OP_MOV(r) "v_mov_b32 %" #r ", %8\n"
#define OP_MAXF(r) "v_max_f32 %" #r ", %" #r ", %8\n"
#define OP_MINF(r) "v_min_f32 %" #r ", %" #r ", %8\n"
#define OP_AND(r) "v_and_b32 %" #r ", %" #r ", %8\n"
#define OP_LSHL(r) "v_lshlrev_b32 %" #r ", 2, %" #r "\n"
#define OP_ADD3(r) "v_add3_u32 %" #r ", %" #r ", %8, %9\n"
#define OP_CVTFI(r) "v_cvt_f32_i32 %" #r ", %" #r "\n"
#define OP_FRACT(r) "v_fract_f32 %" #r ", %" #r "\n"
#define OP_CNDS(r) "v_cndmask_b32 %" #r ", %" #r ", %8, s[20:21]\n"
#define OP_CMPCND(r) "v_cmp_lt_f32 vcc, %" #r ", %8\nv_cndmask_b32 %" #r ", %" #r ", %9, vcc\n"
#define OP_CMPS(r) "v_cmp_lt_f32 s[20:21], %" #r ", %8\n"
#define OP_FMA_NEGABS(r) "v_fma_f32 %" #r ", -%" #r ", |%8|, %9\n"
#define OP_ADD_E64_CLAMP(r) "v_add_f32 %" #r ", %" #r ", %8 clamp\n"
#define OP_MULADDPAIR(r) "v_mul_f32 %" #r ", %" #r ", %8\nv_floor_f32 %" #r ", %" #r "\n"
#define OP_SALUV(r) "s_add_u32 s20, s20, s21\nv_add_f32 %" #r ", %" #r ", %8\n"
#define OP_SALU2(r) "s_add_u32 s20, s20, s21\n"
#define OP_LSHLADD64(r) "v_lshl_add_u64 %" #r ", %" #r ", 2, %10\n"
#define OP_BPERM(r) "ds_bpermute_b32 %" #r ", %8, %" #r "\n"

enum { K_ADD, K_FMA, K_ADDDEP, K_PKFMA, K_PKMUL, K_PKADD, K_WSHR, K_WSHL, K_RSHR, K_QP, K_MOVWSHR, K_RCP, K_FLOOR, K_CVT, K_MAX3,
       K_MED3, K_CND, K_CMP, K_MADU24, K_MADI32, K_MULLO, K_LSHLADD, K_ADDU, K_SUBABS, K_PERM32, K_MUL, K_FMAC, K_MOV, K_MAXF, K_MINF, K_AND, K_LSHL, K_ADD3, K_CVTFI, K_FRACT, K_CNDS, K_CMPCND, K_CMPS, K_FMANEG, K_ADDCLAMP, K_MULFLOOR, K_SALUV, K_SALU, K_LSHLADD64, K_BPERM, K_COUNT };
static const char* kNames[] = {"v_add_f32 (8 chains)", "v_fma_f32", "v_add_f32 (1 dependent chain)", "v_pk_fma_f32", "v_pk_mul_f32",
  "v_pk_add_f32", "v_add_f32_dpp wave_shr:1", "v_add_f32_dpp wave_shl:1", "v_add_f32_dpp row_shr:1", "v_add_f32_dpp quad_perm",
  "v_mov_b32_dpp wave_shr:1", "v_rcp_f32", "v_floor_f32", "v_cvt_i32_f32", "v_max3_f32", "v_med3_f32", "v_cndmask_b32", "v_cmp_lt_f32",
  "v_mad_u32_u24", "v_mad_i32_i24", "v_mul_lo_u32", "v_lshl_add_u32", "v_add_u32", "v_sub_f32 |abs|", "v_permlane32_swap", "v_mul_f32", "v_fmac_f32", "v_mov_b32", "v_max_f32", "v_min_f32", "v_and_b32", "v_lshlrev_b32", "v_add3_u32", "v_cvt_f32_i32", "v_fract_f32", "v_cndmask_b32 (sgpr-pair mask)", "v_cmp + v_cndmask pair (per 2 insts)", "v_cmp_lt_f32 -> sgpr pair", "v_fma_f32 with neg/abs modifiers", "v_add_f32 clamp", "v_mul + v_floor pair (per 2 insts)", "s_add + v_add pair (per 2 insts)", "s_add_u32", "v_lshl_add_u64 (pk regs)",
  "ds_bpermute_b32 (+wait per 32)"};

template <int KIND>
__global__ __launch_bounds__(1024) void rate_kernel(float* out, unsigned long long* cyc, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f,
        a7 = a0 + 7.f;
  float b = 1.0001f, c = 0.9999f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0{a0, a1}, p1{a2, a3}, p2{a4, a5}, p3{a6, a7}, p4{a0, a2}, p5{a1, a3}, p6{a4, a6}, p7{a5, a7}, pb{b, c}, pc{c, b};
  unsigned long long t0, t1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\ns_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
#define RUN(BLK) asm volatile(BLK : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20", "s21")
#define RUNP(BLK) asm volatile(BLK : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(b), "v"(c), "v"(pb), "v"(pc))
    if constexpr (KIND == K_ADD) RUN(B32(OP_ADD));
    if constexpr (KIND == K_FMA) RUN(B32(OP_FMA));
    if constexpr (KIND == K_ADDDEP) RUN(B32(OP_ADDDEP));
    if constexpr (KIND == K_PKFMA) RUNP(B32(OP_PKFMA));
    if constexpr (KIND == K_PKMUL) RUNP(B32(OP_PKMUL));
    if constexpr (KIND == K_PKADD) RUNP(B32(OP_PKADD));
    if constexpr (KIND == K_WSHR) RUN(B32(OP_DPP_WSHR));
    if constexpr (KIND == K_WSHL) RUN(B32(OP_DPP_WSHL));
    if constexpr (KIND == K_RSHR) RUN(B32(OP_DPP_RSHR));
    if constexpr (KIND == K_QP) RUN(B32(OP_DPP_QP));
    if constexpr (KIND == K_MOVWSHR) RUN(B32(OP_MOV_DPP_WSHR));
    if constexpr (KIND == K_RCP) RUN(B32(OP_RCP));
    if constexpr (KIND == K_FLOOR) RUN(B32(OP_FLOOR));
    if constexpr (KIND == K_CVT) RUN(B32(OP_CVT));
    if constexpr (KIND == K_MAX3) RUN(B32(OP_MAX3));
    if constexpr (KIND == K_MED3) RUN(B32(OP_MED3));
    if constexpr (KIND == K_CND) RUN(B32(OP_CNDMASK));
    if constexpr (KIND == K_CMP) RUN(B32(OP_CMP));
    if constexpr (KIND == K_MADU24) RUN(B32(OP_MADU24));
    if constexpr (KIND == K_MADI32) RUN(B32(OP_MADI32));
    if constexpr (KIND == K_MULLO) RUN(B32(OP_MULLO));
    if constexpr (KIND == K_LSHLADD) RUN(B32(OP_LSHLADD));
    if constexpr (KIND == K_ADDU) RUN(B32(OP_ADDU));
    if constexpr (KIND == K_SUBABS) RUN(B32(OP_SUBABS));
    if constexpr (KIND == K_PERM32) RUN(B32(OP_PERM32));
    if constexpr (KIND == K_MUL) RUN(B32(OP_MUL));
    if constexpr (KIND == K_FMAC) RUN(B32(OP_FMAC));
    if constexpr (KIND == K_MOV) RUN(B32(OP_MOV));
    if constexpr (KIND == K_MAXF) RUN(B32(OP_MAXF));
    if constexpr (KIND == K_MINF) RUN(B32(OP_MINF));
    if constexpr (KIND == K_AND) RUN(B32(OP_AND));
    if constexpr (KIND == K_LSHL) RUN(B32(OP_LSHL));
    if constexpr (KIND == K_ADD3) RUN(B32(OP_ADD3));
    if constexpr (KIND == K_CVTFI) RUN(B32(OP_CVTFI));
    if constexpr (KIND == K_FRACT) RUN(B32(OP_FRACT));
    if constexpr (KIND == K_CNDS) RUN(B32(OP_CNDS));
    if constexpr (KIND == K_CMPCND) RUN(B32(OP_CMPCND));
    if constexpr (KIND == K_CMPS) RUN(B32(OP_CMPS));
    if constexpr (KIND == K_FMANEG) RUN(B32(OP_FMA_NEGABS));
    if constexpr (KIND == K_ADDCLAMP) RUN(B32(OP_ADD_E64_CLAMP));
    if constexpr (KIND == K_MULFLOOR) RUN(B32(OP_MULADDPAIR));
    if constexpr (KIND == K_SALUV) RUN(B32(OP_SALUV));
    if constexpr (KIND == K_SALU) RUN(B32(OP_SALU2));
    if constexpr (KIND == K_LSHLADD64) RUNP(B32(OP_LSHLADD64));
    if constexpr (KIND == K_BPERM) RUN(B32(OP_BPERM) "s_waitcnt lgkmcnt(0)\n");
  }
  asm volatile("s_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if ((threadIdx.x & 63) == 0) cyc[gw] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.x + p6.x + p7.x;
}

template <int KIND>
void run(int wps, float* out, unsigned long long* cyc, int iters, double* per_wave, double* wall_us) {
  const int blocks = wps > 4 ? 512 : 256;    // one workgroup of 4*min(wps,4) waves per CU (two for wps = 8): co-residency by construction
  const int threads = 256 * (wps > 4 ? 4 : wps);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * threads / 64);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  *per_wave = (double)h[h.size() / 2] / ((double)iters * 32);
  *wall_us = ms * 1e3;
  hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int KIND>
void report(float* out, unsigned long long* cyc, int iters) {
  printf("%-34s", kNames[KIND]);
  for (int wps : {1, 2, 4, 8}) {
    double pw, us;
    run<KIND>(wps, out, cyc, iters, &pw, &us);
    // s_memtime ticks at 100 MHz-derived constant rate? report both the stamp units and wall-derived ns per instruction per SIMD
    printf(" | w=%d: %6.2f tick/inst/wave = %5.2f /SIMD, %6.3f ns/inst/SIMD", wps, pw, pw / wps, us * 1e3 / ((double)iters * 32 * wps));
  }
  printf("\n");
  fflush(stdout);
}

template <int K>
void all(float* out, unsigned long long* cyc, int iters) {
  report<K>(out, cyc, iters);
  if constexpr (K + 1 < K_COUNT) all<K + 1>(out, cyc, iters);
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 512 * 1024 * 4);
  hipMalloc(&cyc, 512 * 16 * 8);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("device %s, %d CUs, clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  all<0>(out, cyc, 2000);
  return 0;
}
