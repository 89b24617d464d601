// dispatch_ramp.hip -- next-round experiment: how long does the GPU need to get 1,024 one-wave
// workgroups with the step kernel's footprint (34 KB LDS, one wave per SIMD) running, and how far apart do they end?
// Every wave records s_memrealtime at entry and exit around a fixed dependent FMA chain; the host prints the spread of
// the start stamps (dispatch ramp), of the end stamps, and the kernel duration, for 1 and for 2 concurrent streams.
//   hipcc --offload-arch=gfx950 -O3 dispatch_ramp.hip -o dispatch_ramp && ./dispatch_ramp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(64) busy(unsigned long long* stamps, float* out, int iters, float s) {
  extern __shared__ float lds[];
  unsigned long long t0, t1;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  float a = threadIdx.x, b = s;
  lds[threadIdx.x] = a;
  for (int i = 0; i < iters; i++) a = __builtin_fmaf(a, b, 0.5f);   // dependent chain: ~4 cycles per iteration
  a += lds[threadIdx.x ^ 1];
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
  out[blockIdx.x * 64 + threadIdx.x] = a;
}

int main() {
  const int W = 1024, iters = 2400000;   // ~4 ms at 2.4 GHz
  const size_t lds = 34816;
  CK(hipFuncSetAttribute((const void*)busy, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int nstream = 1; nstream <= 2; nstream++) {
    std::vector<hipStream_t> st(nstream);
    std::vector<unsigned long long*> ds(nstream);
    std::vector<float*> dout(nstream);
    for (int k = 0; k < nstream; k++) { CK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking)); CK(hipMalloc(&ds[k], 2 * W * 8)); CK(hipMalloc(&dout[k], W * 64 * 4)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, st[0]));
      for (int it = 0; it < 4; it++)
        for (int k = 0; k < nstream; k++) hipLaunchKernelGGL(busy, dim3(W), dim3(64), lds, st[k], ds[k], dout[k], iters, 1.0000001f);
      for (int k = 1; k < nstream; k++) CK(hipStreamSynchronize(st[k]));
      CK(hipEventRecord(e1, st[0])); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<unsigned long long> h(2 * W);
      CK(hipMemcpy(h.data(), ds[0], 2 * W * 8, hipMemcpyDeviceToHost));
      unsigned long long s0 = ~0ull, s1 = 0, f0 = ~0ull, f1 = 0;
      for (int i = 0; i < W; i++) { s0 = std::min(s0, h[2 * i]); s1 = std::max(s1, h[2 * i]); f0 = std::min(f0, h[2 * i + 1]); f1 = std::max(f1, h[2 * i + 1]); }
      // s_memrealtime ticks at 100 MHz
      printf("streams %d rep %d: 4 launches per stream in %.3f ms (%.3f ms per launch-round); last launch of stream 0: start spread %.1f us, "
             "end spread %.1f us, first start -> last end %.1f us\n", nstream, rep, ms, ms / 4, (s1 - s0) / 100.0, (f1 - f0) / 100.0, (f1 - s0) / 100.0);
    }
    for (int k = 0; k < nstream; k++) { CK(hipFree(ds[k])); CK(hipFree(dout[k])); CK(hipStreamDestroy(st[k])); }
  }
  return 0;
}
