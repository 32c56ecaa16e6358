// Micro-benchmark: issue rate of scalar vs packed fp32 VALU ops on gfx950 (decides whether
// v_pk_{add,mul}_f32 buys anything for the exact, non-fused nearest-neighbour arithmetic).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_valu.hip -o /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a[8];
  float2v p[8];
  for (int i = 0; i < 8; i++) a[i] = seed + i + threadIdx.x * 1e-3f, p[i] = float2v{a[i], a[i] + 0.5f};
  const float m = 1.0000001f, c = 1e-7f;
  const float2v m2 = {m, m}, c2 = {c, c};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (MODE == 0) a[i] = a[i] * m;                 // v_mul_f32
        if (MODE == 1) a[i] = a[i] + c;                 // v_add_f32
        if (MODE == 2) a[i] = __builtin_fmaf(a[i], m, c); // v_fma_f32
        if (MODE == 3) p[i] = p[i] * m2;                // v_pk_mul_f32
        if (MODE == 4) p[i] = p[i] + c2;                // v_pk_add_f32
        if (MODE == 5) p[i] = __builtin_elementwise_fma(p[i], m2, c2);  // v_pk_fma_f32
        if (MODE == 6) a[i] = fminf(fminf(a[i], m), c + a[(i + 1) & 7]);  // add + min3
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int per_iter_ops, int flops_per_op) {
  const int blocks = 256 * 8, iters = 4096;
  float* d;
  hipMalloc(&d, blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, 16, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(d, iters, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double winstr = (double)blocks * 4 * iters * per_iter_ops;  // wave-instructions
  const double rate = winstr / (ms * 1e-3);                         // wave-instr / s chip-wide
  printf("%-14s %8.3f ms  %.3e wave-instr/s  = %.2f cycles/instr/SIMD @2.4GHz  %.1f TFLOP/s\n", name, ms, rate, 1024 * 2.4e9 / rate,
         rate * 64 * flops_per_op / 1e12);
  hipFree(d);
}

int main() {
  run<0>("v_mul_f32", 64, 1);
  run<1>("v_add_f32", 64, 1);
  run<2>("v_fma_f32", 64, 2);
  run<3>("v_pk_mul_f32", 64, 2);
  run<4>("v_pk_add_f32", 64, 2);
  run<5>("v_pk_fma_f32", 64, 4);
  run<6>("add+min3", 128, 1);
  return 0;
}
