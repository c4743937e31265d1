// Diagnostic: ratio of s_memtime ticks to the 100 MHz wall clock (calibrates tools/diag/*_phases).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint64_t* o, float* x) {
  uint64_t m0 = __builtin_readcyclecounter(), s0, w0 = wall_clock64();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s0)::"memory");
  float v = x[threadIdx.x];
  for (int i = 0; i < 200000; ++i) v = v * 1.0001f + 0.5f;
  x[threadIdx.x] = v;
  uint64_t m1 = __builtin_readcyclecounter(), s1, w1 = wall_clock64();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s1)::"memory");
  if (threadIdx.x == 0) { o[0] = m1 - m0; o[1] = s1 - s0; o[2] = w1 - w0; }
}
int main() {
  uint64_t* o; float* x; hipMalloc(&o, 64); hipMalloc(&x, 1024); hipMemset(x, 0, 1024);
  k<<<1, 64>>>(o, x); hipDeviceSynchronize();
  uint64_t h[3]; hipMemcpy(h, o, 24, hipMemcpyDeviceToHost);
  printf("readcyclecounter %llu  s_memtime %llu  wall(100MHz) %llu -> s_memtime = %.1f MHz\n", (unsigned long long)h[0],
         (unsigned long long)h[1], (unsigned long long)h[2], (double)h[1] / h[2] * 100.0);
  return 0;
}
