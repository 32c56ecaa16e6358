/* Compares include/apd_atan2f.h with the C library's atan2f / atanf of the box this runs on, bit for bit.
 * usage: atan2f_check [n_random (default 12000000)]   -> prints "checked N mismatches M" and up to 20 differing inputs; exit 1 on any. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/apd_atan2f.h"

static uint64_t s[2] = {0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull};
static uint64_t rnd(void) { /* xorshift128+ */
  uint64_t a = s[0], b = s[1];
  s[0] = b;
  a ^= a << 23;
  s[1] = a ^ b ^ (a >> 17) ^ (b >> 26);
  return s[1] + b;
}
static float bits(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static uint32_t ubits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
static long checked = 0, bad = 0;
static int same(float a, float b) { return ubits(a) == ubits(b) || (a != a && b != b); }
static void check2(float y, float x) {
  volatile float vy = y, vx = x; /* keep the compiler from folding the library call */
  const float lib = atan2f(vy, vx), mine = apd_atan2f(y, x);
  checked++;
  if (!same(lib, mine)) {
    if (bad < 20) printf("atan2f(%a, %a): libm %a (%08x)  apd %a (%08x)\n", y, x, lib, ubits(lib), mine, ubits(mine));
    bad++;
  }
}
static void check1(float x) {
  volatile float vx = x;
  const float lib = atanf(vx), mine = apd_atanf(x);
  checked++;
  if (!same(lib, mine)) {
    if (bad < 20) printf("atanf(%a): libm %a (%08x)  apd %a (%08x)\n", x, lib, ubits(lib), mine, ubits(mine));
    bad++;
  }
}
static float uni(float lo, float hi) { return lo + (hi - lo) * (float)((rnd() >> 40) * (1.0 / 16777216.0)); }

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 12000000;
  /* 1. atanf over EVERY fp32 bit pattern when asked for (n < 0), else a stride through all of them */
  const uint32_t stride = n < 0 ? 1u : 97u;
  for (uint64_t u = 0; u < (1ull << 32); u += stride) check1(bits((uint32_t)u));
  const long m = n < 0 ? 40000000 : n;
  /* 2. random bit patterns of both arguments (all exponents, signs, subnormals, infinities, NaNs) */
  for (long i = 0; i < m / 4; i++) {
    const uint64_t r = rnd();
    check2(bits((uint32_t)r), bits((uint32_t)(r >> 32)));
  }
  /* 3. what the sensor model feeds it: coordinates of points 0.1 .. 300 m from the sensor (A:168,172-173) */
  for (long i = 0; i < m / 4; i++) {
    const float x = uni(-300.f, 300.f), y = uni(-300.f, 300.f), z = uni(-30.f, 30.f);
    check2(x, sqrtf(y * y + z * z));
    check2(sqrtf(x * x + y * y), z);
    check2(y, x);
  }
  /* 4. ratios close to the interval ends of the reduction, and special values */
  const float ends[] = {0.4375f, 0.6875f, 1.1875f, 2.4375f, 1.0f, 0x1p-29f, 0x1p26f, 0x1p25f, 0x1p60f, 0x1p-60f, 0x1p61f, 0x1p-61f};
  for (unsigned e = 0; e < sizeof ends / sizeof *ends; e++)
    for (int d = -2000; d <= 2000; d++) {
      const float r = bits(ubits(ends[e]) + d);
      for (int q = 0; q < 8; q++) {
        const float x = uni(0.01f, 100.f) * ((q & 1) ? -1.f : 1.f);
        check2(r * x * ((q & 2) ? -1.f : 1.f), x);
      }
      check2(r, 1.0f), check2(-r, 1.0f), check2(r, -1.0f), check2(1.0f, r), check2(1.0f, -r);
    }
  const float sp[] = {0.f, -0.f, 1.f, -1.f, INFINITY, -INFINITY, NAN, 0x1p-149f, -0x1p-149f, 0x1.fffffep127f, -0x1.fffffep127f, 0x1p-126f};
  for (unsigned a = 0; a < sizeof sp / sizeof *sp; a++)
    for (unsigned b = 0; b < sizeof sp / sizeof *sp; b++) check2(sp[a], sp[b]);
  printf("checked %ld mismatches %ld\n", checked, bad);
  return bad != 0;
}
