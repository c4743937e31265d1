// Diagnostic (not part of the product): HIP-event time of an (almost) empty kernel as a function of workgroup size and
// dynamic LDS, to price the fixed cost of a one-workgroup-per-CU launch.   hipcc --offload-arch=gfx950 -O3 launch_cost.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* x) {
  extern __shared__ char sm[];
  if (threadIdx.x == 0 && x[0] == 123.f) { sm[0] = 1; x[1] = sm[0]; }
}
int main() {
  float* x; hipMalloc(&x, 64); hipMemset(x, 0, 64);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int cfg[][3] = {{768, 256, 40960}, {256, 704, 90112}, {256, 704, 0}, {256, 1024, 131072}, {256, 256, 90112}, {512, 256, 0}, {2048, 256, 0}};
  for (auto& c : cfg) {
    float best = 1e9;
    for (int it = 0; it < 20; ++it) {
      hipEventRecord(e0);
      k<<<c[0], c[1], c[2]>>>(x);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (it > 2 && ms < best) best = ms;
    }
    printf("grid %4d block %4d lds %6d: %.2f us\n", c[0], c[1], c[2], best * 1e3);
  }
  return 0;
}
