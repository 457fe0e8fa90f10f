// micro-benchmark: vector-ALU throughput per opcode on gfx950 (wave64), 8 independent chains per wave, 8 waves per SIMD:
// nominal clocks (2.4 GHz) per instruction per SIMD.  The opcode is pinned with inline asm.
#include <hip/hip_runtime.h>
#include <cstdio>
#define OPK(NAME, ASM)                                                                              \
  __global__ __launch_bounds__(64) void NAME(float* out, int iters, float a, float b) {             \
    const float sa = __builtin_amdgcn_readfirstlane(__float_as_int(a)) * 1.0f, sb = b;             \
    const unsigned long long mask = __ballot(threadIdx.x & 1);                                     \
    float x[8];                                                                                     \
    for (int i = 0; i < 8; ++i) x[i] = (float)threadIdx.x * 0.001f + i + 1.5f;                      \
    for (int it = 0; it < iters; ++it) {                                                            \
      _Pragma("unroll") for (int u = 0; u < 8; ++u)                                                 \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(x[i]) : "v"(a), "v"(b), "s"(sa), "s"(sb), "s"(mask)); \
    }                                                                                               \
    float r = 0.f;                                                                                  \
    for (int i = 0; i < 8; ++i) r += x[i];                                                          \
    if (r == 12345.678f) out[0] = r;                                                                \
  }
OPK(k_fma, "v_fma_f32 %0, %0, %1, %2")
OPK(k_fmac, "v_fmac_f32 %0, %1, %2")
OPK(k_mul, "v_mul_f32 %0, %0, %1")
OPK(k_add, "v_add_f32 %0, %0, %1")
OPK(k_addu, "v_add_u32 %0, %0, %1")
OPK(k_xor, "v_xor_b32 %0, %0, %1")
OPK(k_mov, "v_mov_b32 %0, %1")
OPK(k_cnd, "v_cndmask_b32 %0, %0, %1, vcc")
OPK(k_floor, "v_floor_f32 %0, %0")
OPK(k_cvti, "v_cvt_i32_f32 %0, %0")
OPK(k_cvtf, "v_cvt_f32_u32 %0, %0")
OPK(k_med3, "v_med3_i32 %0, %0, %1, %2")
OPK(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
OPK(k_lshladd, "v_lshl_add_u32 %0, %0, 2, %1")
OPK(k_align, "v_alignbit_b32 %0, %0, %0, 7")
OPK(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
OPK(k_min3, "v_min3_f32 %0, %0, %1, %2")
OPK(k_log, "v_log_f32 %0, %0")
OPK(k_exp, "v_exp_f32 %0, %0")
OPK(k_rcp, "v_rcp_f32 %0, %0")
OPK(k_ldexp, "v_ldexp_f32 %0, %0, %1")
OPK(k_mullo, "v_mul_lo_u32 %0, %0, %1")
OPK(k_dpp, "v_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
OPK(k_cmp, "v_cmp_lt_f32 vcc, %0, %1")

OPK(k_fma_s1, "v_fma_f32 %0, %0, %3, %2")
OPK(k_fma_s2, "v_fma_f32 %0, %0, %3, %4")
OPK(k_fmac_s, "v_fmac_f32 %0, %3, %2")
OPK(k_mul_s, "v_mul_f32 %0, %3, %0")
OPK(k_mul_lit, "v_mul_f32 %0, 0x3f8ccccd, %0")
OPK(k_mul_inl, "v_mul_f32 %0, 2.0, %0")
OPK(k_fmamk, "v_fmamk_f32 %0, %0, 0x3f8ccccd, %1")
OPK(k_add_e64, "v_add_f32_e64 %0, %0, %1")
OPK(k_sub, "v_sub_f32 %0, %0, %1")
OPK(k_max, "v_max_f32 %0, %0, %1")
OPK(k_and, "v_and_b32 %0, %0, %1")
OPK(k_lshl, "v_lshlrev_b32 %0, 3, %0")
OPK(k_lshr, "v_lshrrev_b32 %0, 3, %0")
OPK(k_or3, "v_or3_b32 %0, %0, %1, %2")
OPK(k_add3, "v_add3_u32 %0, %0, %1, %2")
OPK(k_xad, "v_xad_u32 %0, %0, %1, %2")
OPK(k_bfe, "v_bfe_u32 %0, %0, 3, 5")
OPK(k_perm, "v_perm_b32 %0, %0, %1, %2")
OPK(k_cvtub, "v_cvt_f32_ubyte0 %0, %0")
OPK(k_cnd64, "v_cndmask_b32_e64 %0, %0, %1, %5")
OPK(k_cmp64, "v_cmp_lt_f32_e64 s[40:41], %0, %1")
OPK(k_cmpx, "v_cmp_class_f32 vcc, %0, %1")
OPK(k_fract, "v_fract_f32 %0, %0")
OPK(k_ceil, "v_ceil_f32 %0, %0")
OPK(k_rndne, "v_rndne_f32 %0, %0")
OPK(k_sqrt, "v_sqrt_f32 %0, %0")
OPK(k_rsq, "v_rsq_f32 %0, %0")
OPK(k_sin, "v_sin_f32 %0, %0")
OPK(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, -1, %0")
OPK(k_readlane, "v_readlane_b32 s42, %0, 3")
OPK(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
OPK(k_subrev, "v_subrev_u32 %0, %3, %0")
OPK(k_addc, "v_addc_co_u32 %0, vcc, 0, %0, vcc")
OPK(k_mulhi, "v_mul_hi_u32 %0, %0, %1")
OPK(k_mul24, "v_mul_u32_u24 %0, %0, %1")
OPK(k_min, "v_min_i32 %0, %0, %1")
OPK(k_minu, "v_min_u32 %0, %0, %1")
OPK(k_ashr, "v_ashrrev_i32 %0, 3, %0")
OPK(k_cvtu, "v_cvt_u32_f32 %0, %0")
OPK(k_cvtfi, "v_cvt_f32_i32 %0, %0")
OPK(k_dpp_mov, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
OPK(k_add_dpp, "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
OPK(k_sdwa, "v_mov_b32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1")

#define OPK64(NAME, ASM)                                                                            \
  __global__ __launch_bounds__(64) void NAME(float* out, int iters, float a, float b) {             \
    unsigned long long x[8];                                                                        \
    const unsigned ua = __float_as_uint(a) & 1023u, ub = __float_as_uint(b) & 1023u;                \
    const unsigned long long sb = (unsigned long long)(size_t)out;                                  \
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;                                             \
    for (int it = 0; it < iters; ++it) {                                                            \
      _Pragma("unroll") for (int u = 0; u < 8; ++u)                                                 \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(x[i]) : "v"(ua), "v"(ub), "s"(sb)); \
    }                                                                                               \
    unsigned long long r = 0;                                                                       \
    for (int i = 0; i < 8; ++i) r += x[i];                                                          \
    if (r == 12345678ull) out[0] = 1.0f;                                                            \
  }
OPK64(k_mad64, "v_mad_u64_u32 %0, s[40:41], %1, %2, %0")
OPK64(k_lshladd64, "v_lshl_add_u64 %0, %0, 4, %3")
OPK64(k_lshladd64v, "v_lshl_add_u64 %0, %0, 4, %0")
OPK64(k_mov64, "v_mov_b64 %0, %0")
OPK64(k_pkfma, "v_pk_fma_f32 %0, %0, %0, %0")
OPK64(k_pkmul, "v_pk_mul_f32 %0, %0, %0")
OPK64(k_pkadd, "v_pk_add_f32 %0, %0, %0")
OPK64(k_lshl64, "v_lshlrev_b64 %0, 4, %0")
OPK(k_lshl_or, "v_lshl_or_b32 %0, %0, 3, %1")
OPK(k_or, "v_or_b32 %0, %0, %1")
OPK(k_bfi, "v_bfi_b32 %0, %0, %1, %2")
OPK(k_fmaak, "v_fmaak_f32 %0, %0, %1, 0x3f8ccccd")
OPK(k_fmac_inl, "v_fmac_f32 %0, 2.0, %1")
OPK(k_fma_inl, "v_fma_f32 %0, %0, 2.0, %1")
OPK(k_fma_neg, "v_fma_f32 %0, -%0, %1, %2")
OPK(k_mul_abs, "v_mul_f32_e64 %0, |%0|, %1")
OPK(k_addco, "v_add_co_u32 %0, vcc, %0, %1")
OPK(k_add_lshl, "v_add_lshl_u32 %0, %0, %1, 2")
OPK(k_sub_u, "v_sub_u32 %0, %0, %1")
OPK(k_not, "v_not_b32 %0, %0")
OPK(k_max_u, "v_max_u32 %0, %0, %1")
OPK(k_min_f, "v_min_f32 %0, %0, %1")
OPK(k_med3f, "v_med3_f32 %0, %0, %1, %2")
OPK(k_cnd_vcc2, "v_cndmask_b32 %0, %1, %0, vcc")
OPK(k_cnd_lit, "v_cndmask_b32_e64 %0, 0, %0, %5")
OPK(k_mul_u16, "v_mul_lo_u16 %0, %0, %1")
OPK(k_sad, "v_sad_u32 %0, %0, %1, %2")
OPK(k_cvt_pk, "v_cvt_pkrtz_f16_f32 %0, %0, %1")
OPK(k_exp_leg, "v_exp_legacy_f32 %0, %0")
OPK(k_frexp, "v_frexp_mant_f32 %0, %0")
OPK(k_trunc, "v_trunc_f32 %0, %0")
OPK(k_mul_leg, "v_mul_legacy_f32 %0, %0, %1")

OPK(k_pair_vcc, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
OPK(k_pair_s, "v_cmp_lt_f32_e64 s[40:41], %0, %1\n v_cndmask_b32_e64 %0, %0, %2, s[40:41]")
OPK(k_cnd_vccw, "v_cndmask_b32 %0, %0, %1, vcc\n s_mov_b64 vcc, %5")
OPK(k_add_s, "v_add_f32 %0, %3, %0")
OPK(k_sub_s, "v_sub_f32 %0, %3, %0")
OPK(k_addu_s, "v_add_u32 %0, %3, %0")
OPK(k_fma_sc, "v_fma_f32 %0, %0, %1, %3")
OPK(k_and_s, "v_and_b32 %0, %3, %0")
OPK(k_cmp_s, "v_cmp_lt_f32 vcc, %3, %0")
OPK(k_max_s, "v_max_f32 %0, %3, %0")
OPK(k_nop, "s_nop 0")
OPK(k_salu, "s_add_u32 s42, s42, 1")

OPK(k_B, "v_cmp_lt_f32 vcc, %0, %1\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_cndmask_b32 %0, %0, %2, vcc")
OPK(k_C, "v_cndmask_b32_e64 %0, %0, %1, vcc")
OPK(k_D, "s_mov_b64 s[40:41], %5\n v_cndmask_b32_e64 %0, %0, %1, s[40:41]")
OPK(k_E, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %0, %0, %1, vcc")
OPK(k_F, "v_cmp_lt_f32 vcc, %0, %1\n s_and_b64 vcc, vcc, %5\n v_cndmask_b32 %0, %0, %2, vcc")
OPK(k_G, "v_cmp_lt_f32_e64 s[40:41], %0, %1\n s_and_b64 s[42:43], s[40:41], %5\n v_cndmask_b32_e64 %0, %0, %2, s[42:43]")
OPK(k_H, "v_cmp_lt_f32 vcc, %0, %1\n s_and_b64 s[42:43], vcc, %5\n v_cndmask_b32_e64 %0, %0, %2, s[42:43]")
OPK(k_I, "v_cmp_lt_f32 vcc, %0, %1\n s_and_b64 vcc, vcc, %5\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_cndmask_b32 %0, %0, %2, vcc")
OPK(k_J, "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2")
OPK(k_K, "v_cmp_lt_f32 vcc, %0, %1\n s_cbranch_vccz 0")
typedef void (*kern)(float*, int, float, float);
static void run(const char* name, kern f, float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2048, blocks = 256 * 4 * 8;
  f<<<blocks, 64>>>(d, 64, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  f<<<blocks, 64>>>(d, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  const double inst = (double)iters * 64 * 8;   // per SIMD: 8 waves x 64 instructions per trip
  printf("%-14s %.3f ms  %.2f clk per instruction per SIMD\n", name, ms, ms * 1e-3 * 2.4e9 / inst);
}
int main() {
  float* d; (void)hipMalloc(&d, 4);
#define R(k) run(#k, k, d)
  R(k_fma); R(k_fmac); R(k_mul); R(k_add); R(k_addu); R(k_xor); R(k_mov); R(k_cnd); R(k_floor); R(k_cvti); R(k_cvtf);
  R(k_med3); R(k_mad24); R(k_lshladd); R(k_align); R(k_and_or); R(k_min3); R(k_log); R(k_exp); R(k_rcp); R(k_ldexp);
  R(k_mullo); R(k_dpp); R(k_cmp);
  R(k_fma_s1); R(k_fma_s2); R(k_fmac_s); R(k_mul_s); R(k_mul_lit); R(k_mul_inl); R(k_fmamk); R(k_add_e64); R(k_sub); R(k_max);
  R(k_and); R(k_lshl); R(k_lshr); R(k_or3); R(k_add3); R(k_xad); R(k_bfe); R(k_perm); R(k_cvtub); R(k_cnd64); R(k_cmp64); R(k_cmpx);
  R(k_fract); R(k_ceil); R(k_rndne); R(k_sqrt); R(k_rsq); R(k_sin); R(k_mbcnt); R(k_readlane); R(k_bcnt);
  R(k_subrev); R(k_addc); R(k_mulhi); R(k_mul24); R(k_min); R(k_minu); R(k_ashr); R(k_cvtu); R(k_cvtfi); R(k_dpp_mov); R(k_add_dpp); R(k_sdwa);
  R(k_mad64); R(k_lshladd64); R(k_lshladd64v); R(k_mov64); R(k_pkfma); R(k_pkmul); R(k_pkadd); R(k_lshl64);
  R(k_lshl_or); R(k_or); R(k_bfi); R(k_fmaak); R(k_fmac_inl); R(k_fma_inl); R(k_fma_neg); R(k_mul_abs); R(k_addco); R(k_add_lshl);
  R(k_sub_u); R(k_not); R(k_max_u); R(k_min_f); R(k_med3f); R(k_cnd_vcc2); R(k_cnd_lit); R(k_mul_u16); R(k_sad); R(k_cvt_pk);
  R(k_exp_leg); R(k_frexp); R(k_trunc); R(k_mul_leg);
  R(k_pair_vcc); R(k_pair_s); R(k_cnd_vccw); R(k_add_s); R(k_sub_s); R(k_addu_s); R(k_fma_sc); R(k_and_s); R(k_cmp_s); R(k_max_s); R(k_nop); R(k_salu);
  R(k_B); R(k_C); R(k_D); R(k_E); R(k_F); R(k_G); R(k_H); R(k_I); R(k_J);
  R(k_fma);
  return 0;
}
