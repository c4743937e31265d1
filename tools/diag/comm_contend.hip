// comm_contend.hip - a stand-in for a collective's kernels on a one-GPU box: `blocks` workgroups of 256 threads that stay
// resident for ~`usec` microseconds each (s_sleep loop on the shader clock), launched on the caller's stream.  RCCL's ring
// kernels hold one workgroup per channel for the whole collective; the persistent weight-stationary GEMM (gemm_ws.hip) wants
// every CU's full register file - tools/diag/comm_contend.py measures what the step loses while such kernels are resident.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/diag/comm_contend.hip -o tools/diag/bin/libcomm_contend.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void resident_kernel(unsigned long long cycles, float* sink) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  float v = 0.f;
  while (__builtin_readcyclecounter() - t0 < cycles) {  // every wave reaches the exit: the counter is monotonic
    __builtin_amdgcn_s_sleep(32);
    v += 1.f;
  }
  if (sink && v < 0.f) sink[threadIdx.x] = v;
}

extern "C" int comm_contend_launch(void* stream, int blocks, double usec, double clock_mhz) {
  const unsigned long long cycles = (unsigned long long)(usec * clock_mhz);
  resident_kernel<<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(cycles, nullptr);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
