// Issue cost of instruction classes on one gfx950 SIMD: cycles per wave-instruction when the SIMD never runs out of ready waves
// (8 waves per SIMD, 8 independent chains per wave) and when ONE wave issues a dependent chain.  The covariance k-NN and the search
// are bound by vector issue at ~4 cycles per instruction, not at v_fma_f32's 2 (docs/experiments.md, round 5): this lists which
// instructions are the 4- and 8-cycle ones.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/ubench_issue tools/ubench_issue.hip && /tmp/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define CHK(x)                                                                  \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
      return 1;                                                                 \
    }                                                                           \
  } while (0)

constexpr int ITERS = 512, UNROLL = 4;  // x 8 instructions per block

// B8: eight independent instructions on registers %0..%7 (32-bit) -- throughput form; D8: the same opcode as one dependent chain on %0
#define KERNEL32(name, B8, D8)                                                                        \
  __global__ void k_##name(float* out, int dep) {                                                     \
    float r0 = threadIdx.x, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f, r4 = r0 + 4.f, r5 = r0 + 5.f, r6 = r0 + 6.f, r7 = r0 + 7.f; \
    if (!dep) {                                                                                       \
      for (int i = 0; i < ITERS; i++) {                                                               \
        _Pragma("unroll") for (int u = 0; u < UNROLL; u++)                                            \
          asm volatile(B8 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)::"vcc", "scc", "s20", "s21", "s22"); \
      }                                                                                               \
    } else {                                                                                          \
      for (int i = 0; i < ITERS; i++) {                                                               \
        _Pragma("unroll") for (int u = 0; u < UNROLL; u++)                                            \
          asm volatile(D8 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)::"vcc", "scc", "s20", "s21", "s22"); \
      }                                                                                               \
    }                                                                                                 \
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678f) out[0] = r0;                            \
  }
#define KERNEL64(name, B8, D8)                                                                        \
  __global__ void k_##name(float* out, int dep) {                                                     \
    double r0 = threadIdx.x, r1 = r0 + 1., r2 = r0 + 2., r3 = r0 + 3., r4 = r0 + 4., r5 = r0 + 5., r6 = r0 + 6., r7 = r0 + 7.; \
    if (!dep) {                                                                                       \
      for (int i = 0; i < ITERS; i++) {                                                               \
        _Pragma("unroll") for (int u = 0; u < UNROLL; u++)                                            \
          asm volatile(B8 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)::"vcc", "scc", "s20", "s21", "s22"); \
      }                                                                                               \
    } else {                                                                                          \
      for (int i = 0; i < ITERS; i++) {                                                               \
        _Pragma("unroll") for (int u = 0; u < UNROLL; u++)                                            \
          asm volatile(D8 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)::"vcc", "scc", "s20", "s21", "s22"); \
      }                                                                                               \
    }                                                                                                 \
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678) out[0] = (float)r0;                       \
  }

// op dst, src, src forms
#define R8_3(op) op " %0, %0, %1\n" op " %1, %1, %2\n" op " %2, %2, %3\n" op " %3, %3, %4\n" op " %4, %4, %5\n" op " %5, %5, %6\n" op " %6, %6, %7\n" op " %7, %7, %0\n"
#define I8_3(op) op " %0, %0, %0\n" op " %1, %1, %1\n" op " %2, %2, %2\n" op " %3, %3, %3\n" op " %4, %4, %4\n" op " %5, %5, %5\n" op " %6, %6, %6\n" op " %7, %7, %7\n"
#define D8_3(op) op " %0, %0, %0\n" op " %0, %0, %0\n" op " %0, %0, %0\n" op " %0, %0, %0\n" op " %0, %0, %0\n" op " %0, %0, %0\n" op " %0, %0, %0\n" op " %0, %0, %0\n"
// op dst, a, b, c
#define I8_4(op) op " %0, %0, %0, %0\n" op " %1, %1, %1, %1\n" op " %2, %2, %2, %2\n" op " %3, %3, %3, %3\n" op " %4, %4, %4, %4\n" op " %5, %5, %5, %5\n" op " %6, %6, %6, %6\n" op " %7, %7, %7, %7\n"
#define D8_4(op) op " %0, %0, %0, %0\n" op " %0, %0, %0, %0\n" op " %0, %0, %0, %0\n" op " %0, %0, %0, %0\n" op " %0, %0, %0, %0\n" op " %0, %0, %0, %0\n" op " %0, %0, %0, %0\n" op " %0, %0, %0, %0\n"
// op dst, src
#define I8_2(op) op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n"
#define D8_2(op) op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0\n"
// with a suffix (dpp controls, constants)
#define I8_2S(op, sfx) op " %0, %0 " sfx "\n" op " %1, %1 " sfx "\n" op " %2, %2 " sfx "\n" op " %3, %3 " sfx "\n" op " %4, %4 " sfx "\n" op " %5, %5 " sfx "\n" op " %6, %6 " sfx "\n" op " %7, %7 " sfx "\n"
#define D8_2S(op, sfx) op " %0, %0 " sfx "\n" op " %0, %0 " sfx "\n" op " %0, %0 " sfx "\n" op " %0, %0 " sfx "\n" op " %0, %0 " sfx "\n" op " %0, %0 " sfx "\n" op " %0, %0 " sfx "\n" op " %0, %0 " sfx "\n"
#define I8_3S(op, sfx) op " %0, %0, %0 " sfx "\n" op " %1, %1, %1 " sfx "\n" op " %2, %2, %2 " sfx "\n" op " %3, %3, %3 " sfx "\n" op " %4, %4, %4 " sfx "\n" op " %5, %5, %5 " sfx "\n" op " %6, %6, %6 " sfx "\n" op " %7, %7, %7 " sfx "\n"
#define D8_3S(op, sfx) op " %0, %0, %0 " sfx "\n" op " %0, %0, %0 " sfx "\n" op " %0, %0, %0 " sfx "\n" op " %0, %0, %0 " sfx "\n" op " %0, %0, %0 " sfx "\n" op " %0, %0, %0 " sfx "\n" op " %0, %0, %0 " sfx "\n" op " %0, %0, %0 " sfx "\n"
// compares into vcc, then nothing reads it (the chain form is the same: a compare has no vector result)
#define C8(op) op " vcc, %0, %1\n" op " vcc, %1, %2\n" op " vcc, %2, %3\n" op " vcc, %3, %4\n" op " vcc, %4, %5\n" op " vcc, %5, %6\n" op " vcc, %6, %7\n" op " vcc, %7, %0\n"
#define C8S(op) op " s[20:21], %0, %1\n" op " s[20:21], %1, %2\n" op " s[20:21], %2, %3\n" op " s[20:21], %3, %4\n" op " s[20:21], %4, %5\n" op " s[20:21], %5, %6\n" op " s[20:21], %6, %7\n" op " s[20:21], %7, %0\n"

KERNEL32(fma_f32, I8_4("v_fma_f32"), D8_4("v_fma_f32"))
KERNEL32(add_f32, I8_3("v_add_f32"), D8_3("v_add_f32"))
KERNEL32(mul_f32, I8_3("v_mul_f32"), D8_3("v_mul_f32"))
KERNEL32(min_f32, I8_3("v_min_f32"), D8_3("v_min_f32"))
KERNEL32(max3_f32, I8_4("v_max3_f32"), D8_4("v_max3_f32"))
KERNEL32(add_u32, I8_3("v_add_u32"), D8_3("v_add_u32"))
KERNEL32(and_b32, I8_3("v_and_b32"), D8_3("v_and_b32"))
KERNEL32(min_u32, I8_3("v_min_u32"), D8_3("v_min_u32"))
KERNEL32(lshlrev_b32, I8_3("v_lshlrev_b32"), D8_3("v_lshlrev_b32"))
KERNEL32(lshl_add_u32, I8_4("v_lshl_add_u32"), D8_4("v_lshl_add_u32"))
KERNEL32(mul_lo_u32, I8_3("v_mul_lo_u32"), D8_3("v_mul_lo_u32"))
KERNEL32(sad_u8, I8_4("v_sad_u8"), D8_4("v_sad_u8"))
KERNEL32(bfe_u32, I8_4("v_bfe_u32"), D8_4("v_bfe_u32"))
KERNEL32(perm_b32, I8_4("v_perm_b32"), D8_4("v_perm_b32"))
KERNEL32(mov_b32, I8_2("v_mov_b32"), D8_2("v_mov_b32"))
KERNEL32(cndmask_vcc, I8_3S("v_cndmask_b32", ", vcc"), D8_3S("v_cndmask_b32", ", vcc"))
KERNEL32(cmp_le_f32, C8("v_cmp_le_f32"), C8("v_cmp_le_f32"))
KERNEL32(cmp_le_u32, C8("v_cmp_le_u32"), C8("v_cmp_le_u32"))
KERNEL32(cmp_le_f32_sgpr, C8S("v_cmp_le_f32"), C8S("v_cmp_le_f32"))
KERNEL32(mov_dpp_quad, I8_2S("v_mov_b32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"), D8_2S("v_mov_b32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))
KERNEL32(add_u32_dpp_quad, I8_3S("v_add_u32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"), D8_3S("v_add_u32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))
KERNEL32(min_f32_dpp_shr, I8_3S("v_min_f32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf"), D8_3S("v_min_f32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf"))
KERNEL32(mov_dpp_bcast, I8_2S("v_mov_b32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf"), D8_2S("v_mov_b32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf"))
KERNEL32(readlane, "v_readlane_b32 s22, %0, 3\n v_readlane_b32 s22, %1, 3\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s22, %3, 3\n v_readlane_b32 s22, %4, 3\n v_readlane_b32 s22, %5, 3\n v_readlane_b32 s22, %6, 3\n v_readlane_b32 s22, %7, 3\n",
         "v_readlane_b32 s22, %0, 3\n v_readlane_b32 s22, %1, 3\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s22, %3, 3\n v_readlane_b32 s22, %4, 3\n v_readlane_b32 s22, %5, 3\n v_readlane_b32 s22, %6, 3\n v_readlane_b32 s22, %7, 3\n")
KERNEL32(readlane_then_use, "v_readlane_b32 s22, %0, 3\n v_add_u32 %1, s22, %1\n v_readlane_b32 s22, %2, 3\n v_add_u32 %3, s22, %3\n v_readlane_b32 s22, %4, 3\n v_add_u32 %5, s22, %5\n v_readlane_b32 s22, %6, 3\n v_add_u32 %7, s22, %7\n",
         "v_readlane_b32 s22, %0, 3\n v_add_u32 %0, s22, %0\n v_readlane_b32 s22, %0, 3\n v_add_u32 %0, s22, %0\n v_readlane_b32 s22, %0, 3\n v_add_u32 %0, s22, %0\n v_readlane_b32 s22, %0, 3\n v_add_u32 %0, s22, %0\n")
KERNEL32(mbcnt_lo, I8_3S("v_mbcnt_lo_u32_b32", ""), D8_3S("v_mbcnt_lo_u32_b32", ""))
KERNEL32(rcp_f32, I8_2("v_rcp_f32"), D8_2("v_rcp_f32"))
KERNEL32(sqrt_f32, I8_2("v_sqrt_f32"), D8_2("v_sqrt_f32"))
KERNEL32(salu_add, "s_add_u32 s22, s22, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s22, s22, 1\n",
         "s_add_u32 s22, s22, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s22, s22, 1\n")
KERNEL32(salu_bcnt_ff1, "s_bcnt1_i32_b64 s22, s[20:21]\n s_ff1_i32_b64 s22, s[20:21]\n s_bcnt1_i32_b64 s22, s[20:21]\n s_ff1_i32_b64 s22, s[20:21]\n s_bcnt1_i32_b64 s22, s[20:21]\n s_ff1_i32_b64 s22, s[20:21]\n s_bcnt1_i32_b64 s22, s[20:21]\n s_ff1_i32_b64 s22, s[20:21]\n",
         "s_bcnt1_i32_b64 s22, s[20:21]\n s_ff1_i32_b64 s22, s[20:21]\n s_bcnt1_i32_b64 s22, s[20:21]\n s_ff1_i32_b64 s22, s[20:21]\n s_bcnt1_i32_b64 s22, s[20:21]\n s_ff1_i32_b64 s22, s[20:21]\n s_bcnt1_i32_b64 s22, s[20:21]\n s_ff1_i32_b64 s22, s[20:21]\n")
KERNEL32(valu_salu_mix, "v_add_u32 %0, %0, %0\n s_add_u32 s22, s22, 1\n v_add_u32 %1, %1, %1\n s_add_u32 s22, s22, 1\n v_add_u32 %2, %2, %2\n s_add_u32 s22, s22, 1\n v_add_u32 %3, %3, %3\n s_add_u32 s22, s22, 1\n",
         "v_add_u32 %0, %0, %0\n s_add_u32 s22, s22, 1\n v_add_u32 %0, %0, %0\n s_add_u32 s22, s22, 1\n v_add_u32 %0, %0, %0\n s_add_u32 s22, s22, 1\n v_add_u32 %0, %0, %0\n s_add_u32 s22, s22, 1\n")

KERNEL32(sub_f32, I8_3("v_sub_f32"), D8_3("v_sub_f32"))
KERNEL32(max_f32, I8_3("v_max_f32"), D8_3("v_max_f32"))
KERNEL32(fmac_f32, I8_3("v_fmac_f32"), D8_3("v_fmac_f32"))
KERNEL32(or_b32, I8_3("v_or_b32"), D8_3("v_or_b32"))
KERNEL32(xor_b32, I8_3("v_xor_b32"), D8_3("v_xor_b32"))
KERNEL32(sub_u32, I8_3("v_sub_u32"), D8_3("v_sub_u32"))
KERNEL32(lshrrev_b32, I8_3("v_lshrrev_b32"), D8_3("v_lshrrev_b32"))
KERNEL32(add3_u32, I8_4("v_add3_u32"), D8_4("v_add3_u32"))
KERNEL32(or3_b32, I8_4("v_or3_b32"), D8_4("v_or3_b32"))
KERNEL32(and_or_b32, I8_4("v_and_or_b32"), D8_4("v_and_or_b32"))
KERNEL32(bfi_b32, I8_4("v_bfi_b32"), D8_4("v_bfi_b32"))
KERNEL32(add_lshl_u32, I8_4("v_add_lshl_u32"), D8_4("v_add_lshl_u32"))
KERNEL32(mad_u32_u24, I8_4("v_mad_u32_u24"), D8_4("v_mad_u32_u24"))
KERNEL32(mul_u32_u24, I8_3("v_mul_u32_u24"), D8_3("v_mul_u32_u24"))
KERNEL32(max_i32, I8_3("v_max_i32"), D8_3("v_max_i32"))
KERNEL32(med3_f32, I8_4("v_med3_f32"), D8_4("v_med3_f32"))
KERNEL32(min3_u32, I8_4("v_min3_u32"), D8_4("v_min3_u32"))
KERNEL32(cvt_f32_u32, I8_2("v_cvt_f32_u32"), D8_2("v_cvt_f32_u32"))
KERNEL32(bcnt_u32, I8_3("v_bcnt_u32_b32"), D8_3("v_bcnt_u32_b32"))
KERNEL32(ffbl_b32, I8_2("v_ffbl_b32"), D8_2("v_ffbl_b32"))
KERNEL32(cndmask_e64_sgpr, I8_3S("v_cndmask_b32_e64", ", s[20:21]"), D8_3S("v_cndmask_b32_e64", ", s[20:21]"))
KERNEL32(cndmask_after_cmp, "v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_le_f32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n v_cmp_le_f32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc\n v_cmp_le_f32 vcc, %5, %4\n v_cndmask_b32 %7, %7, %6, vcc\n",
         "v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n")
KERNEL32(cndmask_const_vcc, "s_mov_b64 vcc, 0x5555\n" I8_3S("v_cndmask_b32", ", vcc"), "s_mov_b64 vcc, 0x5555\n" D8_3S("v_cndmask_b32", ", vcc"))
KERNEL32(cndmask_2src, "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n",
         "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n")
KERNEL32(mix_fma_min, "v_fma_f32 %0, %0, %0, %0\n v_min_f32 %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_min_f32 %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_min_f32 %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_min_f32 %7, %7, %7\n",
         "v_fma_f32 %0, %0, %0, %0\n v_min_f32 %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_min_f32 %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_min_f32 %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_min_f32 %0, %0, %0\n")

KERNEL64(pk_add_f32, I8_3("v_pk_add_f32"), D8_3("v_pk_add_f32"))
KERNEL64(pk_mul_f32, I8_3("v_pk_mul_f32"), D8_3("v_pk_mul_f32"))
KERNEL64(pk_fma_f32, I8_4("v_pk_fma_f32"), D8_4("v_pk_fma_f32"))
KERNEL64(add_f64, I8_3("v_add_f64"), D8_3("v_add_f64"))
KERNEL64(mul_f64, I8_3("v_mul_f64"), D8_3("v_mul_f64"))
KERNEL64(fma_f64, I8_4("v_fma_f64"), D8_4("v_fma_f64"))
KERNEL64(min_f64, I8_3("v_min_f64"), D8_3("v_min_f64"))
KERNEL64(cmp_le_u64, C8("v_cmp_le_u64"), C8("v_cmp_le_u64"))
KERNEL64(cmp_le_f64, C8("v_cmp_le_f64"), C8("v_cmp_le_f64"))
KERNEL64(lshlrev_b64, "v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %2, 1, %2\n v_lshlrev_b64 %3, 1, %3\n v_lshlrev_b64 %4, 1, %4\n v_lshlrev_b64 %5, 1, %5\n v_lshlrev_b64 %6, 1, %6\n v_lshlrev_b64 %7, 1, %7\n",
         "v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %0, 1, %0\n")
KERNEL64(mov_b64, I8_2("v_mov_b64"), D8_2("v_mov_b64"))
KERNEL64(rcp_f64, I8_2("v_rcp_f64"), D8_2("v_rcp_f64"))
KERNEL64(cvt_f64_f32, "v_cvt_f64_f32 %0, v1\n v_cvt_f64_f32 %1, v1\n v_cvt_f64_f32 %2, v1\n v_cvt_f64_f32 %3, v1\n v_cvt_f64_f32 %4, v1\n v_cvt_f64_f32 %5, v1\n v_cvt_f64_f32 %6, v1\n v_cvt_f64_f32 %7, v1\n",
         "v_cvt_f64_f32 %0, v1\n v_cvt_f64_f32 %1, v1\n v_cvt_f64_f32 %2, v1\n v_cvt_f64_f32 %3, v1\n v_cvt_f64_f32 %4, v1\n v_cvt_f64_f32 %5, v1\n v_cvt_f64_f32 %6, v1\n v_cvt_f64_f32 %7, v1\n")

// LDS: eight independent reads / writes of each width (addresses spread over the banks: lane * width)
typedef float f4 __attribute__((ext_vector_type(4)));
template <int W>
__global__ void k_lds(float* out, int write) {
  __shared__ __attribute__((aligned(16))) float buf[64 * 4 * 2];
  const unsigned a = threadIdx.x * W * 4;
  float v1 = threadIdx.x;
  double v2 = threadIdx.x;
  f4 v4 = {v1, 1.f, 2.f, 3.f};
  buf[threadIdx.x] = v1;
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int u = 0; u < UNROLL * 8; u++) {
      if (write) {
        if (W == 1) asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v1) : "memory");
        if (W == 2) asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v2) : "memory");
        if (W == 4) asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(v4) : "memory");
      } else {
        if (W == 1) asm volatile("ds_read_b32 %0, %1" : "=v"(v1) : "v"(a) : "memory");
        if (W == 2) asm volatile("ds_read_b64 %0, %1" : "=v"(v2) : "v"(a) : "memory");
        if (W == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(v4) : "v"(a) : "memory");
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (v1 + (float)v2 + v4.x == 12345.678f) out[0] = v1;
}

struct Entry {
  const char* name;
  void (*fn)(float*, int);
};

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const char* only = argc > 1 ? argv[1] : nullptr;
  float* d_out;
  CHK(hipMalloc(&d_out, 64));
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double ghz = prop.clockRate * 1e-6;
  std::printf("# %s: %d CUs, clockRate %.2f GHz; cycles below assume that clock (the relative order does not)\n", prop.name, cus, ghz);
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
#define E(n) {#n, k_##n}
  const std::vector<Entry> list = {E(fma_f32), E(add_f32), E(mul_f32), E(min_f32), E(max3_f32), E(pk_add_f32), E(pk_mul_f32), E(pk_fma_f32), E(add_u32), E(and_b32),
                                   E(min_u32), E(lshlrev_b32), E(lshl_add_u32), E(mul_lo_u32), E(sad_u8), E(bfe_u32), E(perm_b32), E(mov_b32), E(mov_b64), E(cndmask_vcc),
                                   E(cmp_le_f32), E(cmp_le_u32), E(cmp_le_f32_sgpr), E(cmp_le_u64), E(cmp_le_f64), E(mov_dpp_quad), E(add_u32_dpp_quad), E(min_f32_dpp_shr),
                                   E(mov_dpp_bcast), E(readlane), E(readlane_then_use), E(mbcnt_lo), E(rcp_f32), E(sqrt_f32), E(add_f64), E(mul_f64), E(fma_f64), E(min_f64),
                                   E(lshlrev_b64), E(rcp_f64), E(cvt_f64_f32), E(salu_add), E(salu_bcnt_ff1), E(valu_salu_mix), E(sub_f32), E(max_f32), E(fmac_f32), E(or_b32), E(xor_b32), E(sub_u32), E(lshrrev_b32), E(add3_u32), E(or3_b32), E(and_or_b32), E(bfi_b32),
                                   E(add_lshl_u32), E(mad_u32_u24), E(mul_u32_u24), E(max_i32), E(med3_f32), E(min3_u32), E(cvt_f32_u32), E(bcnt_u32), E(ffbl_b32), E(cndmask_e64_sgpr),
                                   E(cndmask_after_cmp), E(cndmask_const_vcc), E(cndmask_2src), E(mix_fma_min)};
  const double n_inst = (double)ITERS * UNROLL * 8;
  std::printf("%-24s %12s %12s %14s\n", "instruction", "8 waves/SIMD", "1 wave/SIMD", "1 wave, chain");
  auto run = [&](void (*fn)(float*, int), int waves_per_simd, int dep, double& cyc) -> int {
    const dim3 grid((unsigned)(cus * 4 * waves_per_simd)), block(64);
    hipLaunchKernelGGL(fn, grid, block, 0, 0, d_out, dep);  // warm-up
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(fn, grid, block, 0, 0, d_out, dep);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    cyc = (double)ms / 5 * 1e-3 * ghz * 1e9 / (n_inst * waves_per_simd);  // cycles of a SIMD per wave-instruction
    return 0;
  };
  for (const Entry& e : list) {
    if (only && !std::strstr(e.name, only)) continue;
    std::fprintf(stderr, "[%s]\n", e.name);
    double c8, c1, cd;
    if (run(e.fn, 8, 0, c8) || run(e.fn, 1, 0, c1) || run(e.fn, 1, 1, cd)) return 1;
    std::printf("%-24s %12.2f %12.2f %14.2f\n", e.name, c8, c1, cd);
  }
  struct L {
    const char* name;
    void (*fn)(float*, int);
    int w;
  };
  const L lds[] = {{"ds_read_b32", k_lds<1>, 0}, {"ds_read_b64", k_lds<2>, 0}, {"ds_read_b128", k_lds<4>, 0}, {"ds_write_b32", k_lds<1>, 1}, {"ds_write_b64", k_lds<2>, 1}, {"ds_write_b128", k_lds<4>, 1}};
  for (const L& l : lds) {
    if (only && !std::strstr(l.name, only)) continue;
    std::fprintf(stderr, "[%s]\n", l.name);
    double c8, c1;
    if (run(l.fn, 8, l.w, c8) || run(l.fn, 1, l.w, c1)) return 1;
    std::printf("%-24s %12.2f %12.2f   (cycles of a SIMD; the CU's four SIMDs share one LDS)\n", l.name, c8, c1);
  }
  return 0;
}
