// Diagnostic: does a kernel take a 26 KB by-value argument at run time on this stack?  (the Adam launch's descriptor table)
//   hipcc --offload-arch=gfx950 -O2 tools/diag/kernarg_probe.hip -o /tmp/kernarg_probe && /tmp/kernarg_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Big { float* p[3200]; int n; };
__global__ void k(Big b) { if ((int)threadIdx.x < b.n) b.p[threadIdx.x * 20][0] = (float)threadIdx.x + 1.f; }
int main() {
  Big b;
  float* d;
  if (hipMalloc(&d, 64 * sizeof(float)) != hipSuccess) return 2;
  hipMemset(d, 0, 64 * sizeof(float));
  for (int i = 0; i < 3200; ++i) b.p[i] = d + (i / 20 % 64);
  b.n = 64;
  k<<<1, 64>>>(b);
  hipError_t e = hipDeviceSynchronize();
  float h[64];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("sizeof(arg) = %zu, launch: %s, h[0] = %g h[63] = %g\n", sizeof(Big), hipGetErrorString(e), h[0], h[63]);
  return e == hipSuccess && h[63] == 64.f ? 0 : 1;
}
