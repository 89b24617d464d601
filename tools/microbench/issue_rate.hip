// issue_rate.hip -- how fast does ONE wave per SIMD issue fp32 work on gfx950?
//   A: scalar v_fma_f32 chains (8 independent accumulators)
//   B: packed v_pk_fma_f32 chains (8 independent float2 accumulators)
//   C: as A, but only lanes 0..31 of each wave active (does the SIMD-32 skip the empty half?)
//   D: as A with 2 waves per SIMD (2048 waves)
//   F: v_fma_f64 chains (8 independent double accumulators), one wave per SIMD and two: what do the fp64 decision
//      distances and accumulators of brs_core.hpp cost per instruction next to the fp32 force path?
//   G: v_cndmask_b32 chains (the select chains of the collision code), H: v_mov_b32 dpp quad_perm (lane-pair exchange)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void __launch_bounds__(64) k(float* out, int iters, float s) {
  int lane = threadIdx.x;
  if (MODE == 2 && lane >= 32) return;
  if (MODE == 3) {
    double a[8], m = (double)s, c = 0.5;
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = (double)i + lane;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < 8; i++) a[i] = __builtin_fma(a[i], m, c);
    }
    double r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r += a[i];
    out[blockIdx.x * 64 + lane] = (float)r;
  } else if (MODE == 4) {
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = (float)i + lane; b[i] = s * (float)(i + 1); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < 8; i++) { float t = a[i] > b[(i + 1) & 7] ? b[i] : a[i]; asm volatile("" : "+v"(t)); a[i] = t; }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r += a[i];
    out[blockIdx.x * 64 + lane] = r;
  } else if (MODE == 5) {
    int a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = i + lane;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < 8; i++) a[i] = __builtin_amdgcn_update_dpp(a[(i + 1) & 7], a[i], 0xB1, 0xF, 0xA, false);
    }
    int r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r += a[i];
    out[blockIdx.x * 64 + lane] = (float)r;
  } else if (MODE == 1) {
    f2 a[8], m = {s, s * 1.0001f}, c = {0.5f, 0.25f};
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = (f2){(float)i + lane, (float)i - lane};
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < 8; i++) a[i] = __builtin_elementwise_fma(a[i], m, c);
    }
    f2 r = {0, 0};
#pragma unroll
    for (int i = 0; i < 8; i++) r += a[i];
    out[blockIdx.x * 64 + lane] = r.x + r.y;
  } else {
    float a[8], m = s, c = 0.5f;
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = (float)i + lane;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < 8; i++) a[i] = __builtin_fmaf(a[i], m, c);
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r += a[i];
    out[blockIdx.x * 64 + lane] = r;
  }
}
template <int MODE> float run(float* out, int blocks, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, 0.999f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, 0.999f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 64 * 4);
  const int iters = 200000;
  float a = run<0>(out, 1024, iters), b = run<1>(out, 1024, iters), c = run<2>(out, 1024, iters), d = run<0>(out, 2048, iters), e = run<1>(out, 2048, iters);
  double ninst = 8.0 * iters;
  printf("A scalar fma   1 wave/SIMD : %.3f ms  -> %.2f cycles/instr @2.4GHz\n", a, a * 1e-3 * 2.4e9 / ninst);
  printf("B packed fma   1 wave/SIMD : %.3f ms  -> %.2f cycles/instr (2 fma each)\n", b, b * 1e-3 * 2.4e9 / ninst);
  printf("C scalar fma half-wave     : %.3f ms  -> %.2f cycles/instr\n", c, c * 1e-3 * 2.4e9 / ninst);
  printf("D scalar fma   2 waves/SIMD: %.3f ms  -> %.2f cycles/instr per wave\n", d, d * 1e-3 * 2.4e9 / ninst);
  printf("E packed fma   2 waves/SIMD: %.3f ms  -> %.2f cycles/instr per wave\n", e, e * 1e-3 * 2.4e9 / ninst);
  float f1 = run<3>(out, 1024, iters), f2 = run<3>(out, 2048, iters), g1 = run<4>(out, 1024, iters), g2 = run<4>(out, 2048, iters), h1 = run<5>(out, 1024, iters), h2 = run<5>(out, 2048, iters);
  printf("F v_fma_f64    1 wave/SIMD : %.3f ms  -> %.2f cycles/instr\n", f1, f1 * 1e-3 * 2.4e9 / ninst);
  printf("F v_fma_f64    2 waves/SIMD: %.3f ms  -> %.2f cycles/instr per wave\n", f2, f2 * 1e-3 * 2.4e9 / ninst);
  printf("G cmp+cndmask  1 wave/SIMD : %.3f ms  -> %.2f cycles per (v_cmp, v_cndmask) pair\n", g1, g1 * 1e-3 * 2.4e9 / ninst);
  printf("G cmp+cndmask  2 waves/SIMD: %.3f ms  -> %.2f cycles per pair per wave\n", g2, g2 * 1e-3 * 2.4e9 / ninst);
  printf("H v_mov dpp    1 wave/SIMD : %.3f ms  -> %.2f cycles/instr\n", h1, h1 * 1e-3 * 2.4e9 / ninst);
  printf("H v_mov dpp    2 waves/SIMD: %.3f ms  -> %.2f cycles/instr per wave\n", h2, h2 * 1e-3 * 2.4e9 / ninst);
  return 0;
}
