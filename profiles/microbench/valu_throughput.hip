#include <hip/hip_runtime.h>
#include <cstdio>
// Throughput of common VALU ops at 8 waves/SIMD (wall clock). 8 independent instrs per loop trip via inline asm.
#define BODY8(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(fm), "v"(m), "s"(mask) : "vcc")
#define I_CND_VCC(k) "v_cndmask_b32 %" #k ", %" #k ", %12, vcc\n"
#define I_CND_SGPR(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %12, %14\n"
#define I_CND_E64VCC(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %12, vcc\n"
#define I_CMP_CND(k) "v_cmp_lt_f32 vcc, %" #k ", %12\n v_cndmask_b32 %" #k ", %" #k ", %12, vcc\n"
#define I_CMPS_CND(k) "v_cmp_lt_f32_e64 %14, %" #k ", %12\n v_cndmask_b32_e64 %" #k ", %" #k ", %12, %14\n"
#define I_ADDU(k) "v_add_u32 %" #k ", %" #k ", %12\n"
#define I_AND(k) "v_and_b32 %" #k ", %" #k ", %12\n"
#define I_LSHL(k) "v_lshlrev_b32 %" #k ", 1, %" #k "\n"
#define I_CMPF32(k) "v_cmp_lt_f32 vcc, %" #k ", %12\n"
#define I_CMPF64(k) "v_cmp_lt_f64 vcc, %8, %13\n"
#define I_CMPF64S(k) "v_cmp_lt_f64_e64 %14, %8, %13\n"
#define I_MAXF64(k) "v_max_f64 %8, %8, %13\n v_max_f64 %9, %9, %13\n"
#define I_MOV64(k) "v_mov_b64 %8, %9\n"
#define I_BFI(k) "v_bfi_b32 %" #k ", %12, %" #k ", %" #k "\n"
#define I_MAD(k) "v_mad_u32_u24 %" #k ", %" #k ", %12, %" #k "\n"
#define I_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %12\n"
#define I_RCP64(k) "v_rcp_f64 %8, %8\n"
#define I_RSQ64(k) "v_rsq_f64 %8, %8\n"
#define I_DIVFIX(k) "v_div_fixup_f64 %8, %8, %13, %13\n"
template <int MODE> __global__ void k(unsigned long long* out, int iters, float seed, unsigned long long mask) {
  float f0 = seed + threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
  double a0 = f0, a1 = f1, a2 = f2, a3 = f3; const float fm = 1.0000001f; const double m = 1.0000001;
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) BODY8(I_CND_VCC); if (MODE == 1) BODY8(I_CND_SGPR); if (MODE == 2) BODY8(I_ADDU); if (MODE == 3) BODY8(I_AND);
    if (MODE == 4) BODY8(I_LSHL); if (MODE == 5) BODY8(I_CMPF32); if (MODE == 6) BODY8(I_CMPF64); if (MODE == 7) BODY8(I_CMPF64S);
    if (MODE == 8) BODY8(I_MAXF64); if (MODE == 9) BODY8(I_MOV64); if (MODE == 10) BODY8(I_BFI); if (MODE == 11) BODY8(I_MAD);
    if (MODE == 12) BODY8(I_MULLO); if (MODE == 16) BODY8(I_CND_E64VCC); if (MODE == 17) BODY8(I_CMP_CND); if (MODE == 18) BODY8(I_CMPS_CND); if (MODE == 13) BODY8(I_RCP64); if (MODE == 14) BODY8(I_RSQ64); if (MODE == 15) BODY8(I_DIVFIX);
  }
  if (f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + a0 + a1 + a2 + a3 == 1.2345) out[1] = 1;
}
template <int MODE> void run(const char* name, unsigned long long* d, int per) {
  int iters = 100000, wps = 8; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256 * wps, 256>>>(d, 1000, 1.0f, 0x5555555555555555ull); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<256 * wps, 256>>>(d, iters, 1.0f, 0x5555555555555555ull); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-22s %.2f cycles per wave-instruction per SIMD (8 waves/SIMD, 2.39 GHz)\n", name, ms * 1e-3 * 2.39e9 / ((double)per * iters * wps));
}
int main() {
  unsigned long long* d; hipMalloc(&d, 64);
  run<0>("v_cndmask vcc", d, 8); run<16>("v_cndmask_e64 vcc", d, 8); run<17>("v_cmp vcc + cndmask vcc (pair)", d, 16); run<18>("v_cmp sgpr + cndmask sgpr (pair)", d, 16); run<1>("v_cndmask_e64 sgpr", d, 8); run<2>("v_add_u32", d, 8); run<3>("v_and_b32", d, 8); run<4>("v_lshlrev_b32", d, 8);
  run<5>("v_cmp_lt_f32 vcc", d, 8); run<6>("v_cmp_lt_f64 vcc", d, 8); run<7>("v_cmp_lt_f64 sgpr", d, 8); run<8>("v_max_f64", d, 16); run<9>("v_mov_b64", d, 8);
  run<10>("v_bfi_b32", d, 8); run<11>("v_mad_u32_u24", d, 8); run<12>("v_mul_lo_u32", d, 8); run<13>("v_rcp_f64", d, 8); run<14>("v_rsq_f64", d, 8); run<15>("v_div_fixup_f64", d, 8);
}
