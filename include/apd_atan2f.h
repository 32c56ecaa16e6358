/* apd_atan2f.h -- ONE atan2f for the kernels and for the checker.
 *
 * The reference evaluates the three angles of the APD sensor model with the C library's float overloads
 * (fast_apdgicp/include/fast_gicp/gicp/impl/fast_apdgicp_impl.hpp:168,172-173: `atan2(float, float)` under
 * `using namespace std`), i.e. with glibc's generic flt-32 `atan2f` / `atanf` on the platforms its README names
 * (Ubuntu 18.04 / 20.04: glibc 2.27 / 2.31; x86-64 has no assembly or multiarch variant of either function).  That
 * implementation is the fdlibm algorithm (Sun Microsystems' e_atan2f.c / s_atanf.c as shipped under
 * sysdeps/ieee754/flt-32 up to glibc 2.40; not correctly rounded: up to ~1 ulp).  The device's libm (ocml) is a
 * different ~1 ulp implementation, which made every H / b / cost comparison a 5e-6 one.  This header restates the
 * published fdlibm algorithm -- same argument reduction, same constants, same operation order, no contraction -- once,
 * for host and device; `tests/test_atan2f.py` checks it bit for bit against the C library of the box it runs on.
 *
 * Must be compiled without floating-point contraction (`-ffp-contract=off`; the functions also carry the pragma) and with
 * IEEE fp32 division (hipcc's default: -fhip-fp32-correctly-rounded-divide-sqrt).
 */
#ifndef APD_ATAN2F_H_
#define APD_ATAN2F_H_

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define APD_ATAN_HD __host__ __device__ __forceinline__
#else
#define APD_ATAN_HD static inline
#endif

#ifdef __cplusplus
namespace apd {
#endif

APD_ATAN_HD int32_t apd_f2i(float x) {
  int32_t i;
  memcpy(&i, &x, 4);
  return i;
}
APD_ATAN_HD float apd_i2f(int32_t i) {
  float x;
  memcpy(&x, &i, 4);
  return x;
}
APD_ATAN_HD float apd_fabsf(float x) { return apd_i2f(apd_f2i(x) & 0x7fffffff); }

/* fdlibm s_atanf.c: |x| is reduced to one of five intervals by the identities atan(x) = atan(c) + atan((x - c) / (1 + c x)),
 * c = 0.5, 1, 1.5, inf; an odd polynomial of degree 23 (eleven coefficients, split into even and odd powers) on the reduced
 * argument; atan(c) as hi + lo.
 *
 * Written without branches, for a wavefront whose 64 lanes fall into different intervals: the original's five-way `if` becomes
 * ONE row of a table -- the reduced argument of every interval is (nm |x| + na) / (dm |x| + da) with a multiplier that is 0, 1,
 * 1.5 or 2 and an addend the original's own constant (1 |x| and |x| + 0 are |x| itself, 0 |x| + c is c: the operations that
 * round are the original's, with the original's operands), and the innermost interval, whose result the original writes as
 * x - x p, is the general hi - ((x p - lo) - x) with hi = lo = 0 (0 - (t - r) is r - t bit for bit).  The sign is put on at the
 * end (the original does the same above 7/16, and below it x - x p is odd bit for bit).  tests/test_atan2f.py compares the
 * result with the C library's over every fp32 bit pattern.
 *
 * Row i of the table: {nm, na, dm, da, hi, lo, 0, 0}; rows: |x| < 7/16, < 11/16, < 19/16, < 39/16, above.  The kernels keep a
 * copy in LDS (one 16-byte and one 8-byte read per call instead of two dozen selects between literals). */
#define APD_ATAN_TAB_ROWS 5
#define APD_ATAN_TAB_STRIDE 8
#define APD_ATAN_TAB_INIT                                                                            \
  {1.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f,                                                   \
   2.0f, -1.0f, 1.0f, 2.0f, 4.6364760399e-01f, 5.0121582440e-09f, 0.0f, 0.0f,                        \
   1.0f, -1.0f, 1.0f, 1.0f, 7.8539812565e-01f, 3.7748947079e-08f, 0.0f, 0.0f,                        \
   1.0f, -1.5f, 1.5f, 1.0f, 9.8279368877e-01f, 3.4473217170e-08f, 0.0f, 0.0f,                        \
   0.0f, -1.0f, 1.0f, 0.0f, 1.5707962513e+00f, 7.5497894159e-08f, 0.0f, 0.0f}

/* atanf of a NON-NEGATIVE finite or infinite argument (what atan2f feeds it: |y / x|); `tab` as above */
APD_ATAN_HD float apd_atanf_pos(float ax, const float* tab) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const uint32_t ix = (uint32_t)apd_f2i(ax);
  /* row: the number of interval ends at or below |x| (ix < 2^31: `end - 1 - ix` is negative, i.e. has bit 31 set, iff ix >= end) */
  const uint32_t row = ((0x3ee00000u - 1u - ix) >> 31) + ((0x3f300000u - 1u - ix) >> 31) + ((0x3f980000u - 1u - ix) >> 31) + ((0x401c0000u - 1u - ix) >> 31);
  const float* t = tab + row * APD_ATAN_TAB_STRIDE;
  const float num = t[0] * ax + t[1], den = t[2] * ax + t[3];
  const float r = num / den;
  const float z = r * r, w = z * z;
  const float s1 = z * (3.3333334327e-01f + w * (1.4285714924e-01f + w * (9.0908870101e-02f + w * (6.6610731184e-02f + w * (4.9768779427e-02f + w * 1.6285819933e-02f)))));
  const float s2 = w * (-2.0000000298e-01f + w * (-1.1111110449e-01f + w * (-7.6918758452e-02f + w * (-5.8335702866e-02f + w * -3.6531571299e-02f))));
  const float p = r * (s1 + s2);
  float res = t[4] - ((p - t[5]) - r);
  res = ix < 0x31000000u ? ax : res;                                            /* |x| < 2^-29 */
  res = ix >= 0x4c000000u ? 1.5707962513e+00f + 7.5497894159e-08f : res;        /* |x| >= 2^25: atan(inf) hi + lo */
  return res;
}

#if defined(__HIP_DEVICE_COMPILE__)
#define APD_ATAN_TAB_DECL(name)  /* device code passes its own (LDS) copy */
#else
#define APD_ATAN_TAB_DECL(name) static const float name[APD_ATAN_TAB_ROWS * APD_ATAN_TAB_STRIDE] = APD_ATAN_TAB_INIT
#endif

/* fdlibm e_atan2f.c: the quadrant from the signs, atanf(|y / x|) in between.  `tiny` of the original only raises the inexact
 * flag (pi + 1e-30f == pi in fp32) and is left out.  The original's shortcuts are the general path's own results and need no
 * line here: `x == 1: atanf(y)` (atanf is odd bit for bit), `|y / x| > 2^60: pi/2 + pi_lo/2` (= atanf of anything >= 2^25) and
 * `x < 0, |y / x| < 2^-60: z = 0` (z - pi_lo rounds to -pi_lo for every z below 2^-49).  (-,-) is written (z - pi_lo) - pi in the
 * original: the negative of pi - (z - pi_lo), bit for bit.  The special operands -- a zero, an infinity, a NaN -- are rare and
 * sit behind ONE test. */
APD_ATAN_HD float apd_atan2f_tab(float y, float x, const float* tab) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const float pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
  const uint32_t hx = (uint32_t)apd_f2i(x), hy = (uint32_t)apd_f2i(y), ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
  const float z = apd_atanf_pos(apd_fabsf(y / x), tab);
  const float v = (hx >> 31) ? pi - (z - pi_lo) : z;
  float res = apd_i2f((int32_t)((uint32_t)apd_f2i(v) ^ (hy & 0x80000000u)));
  if (((ix - 1u) | (iy - 1u)) >= 0x7f7fffffu) { /* a zero, an infinity or a NaN among the operands */
    const int m = (int)((hy >> 31) & 1u) | (int)((hx >> 30) & 2u); /* 2 sign(x) + sign(y) */
    const float sy_pio2 = (hy >> 31) ? -pi_o_2 : pi_o_2;
    res = iy == 0x7f800000u ? sy_pio2 : res;                                          /* y infinite */
    if (ix == 0x7f800000u)                                                             /* x infinite */
      res = iy == 0x7f800000u ? (m == 0 ? pi_o_4 : m == 1 ? -pi_o_4 : m == 2 ? 3.0f * pi_o_4 : -3.0f * pi_o_4)
                              : (m == 0 ? 0.0f : m == 1 ? -0.0f : m == 2 ? pi : -pi);
    res = ix == 0 ? sy_pio2 : res;                                                     /* x == 0 */
    res = iy == 0 ? (m < 2 ? y : (m == 2 ? pi : -pi)) : res;                           /* y == 0 */
    res = (ix > 0x7f800000u || iy > 0x7f800000u) ? x + y : res;                        /* NaN */
  }
  return res;
}

#if !defined(__HIP_DEVICE_COMPILE__)
/* host: the table is a static constant */
static inline float apd_atan2f(float y, float x) {
  APD_ATAN_TAB_DECL(tab);
  return apd_atan2f_tab(y, x, tab);
}
static inline float apd_atanf(float x) {
  APD_ATAN_TAB_DECL(tab);
  const float r = apd_atanf_pos(apd_fabsf(x), tab);
  return x != x ? x + x : (apd_f2i(x) < 0 ? -r : r);
}
#endif

#ifdef __cplusplus
}  // namespace apd
#endif
#endif /* APD_ATAN2F_H_ */
