// mfma_valu_overlap.hip - do MFMA and VALU work of the two waves that share a SIMD overlap on gfx950?
//
// One workgroup of 8 waves per CU (two per SIMD, as gemm_ws.hip runs).  Waves 0..3 run NM dependent-free MFMA 16x16x32 bf16
// (4 accumulator chains), waves 4..7 run NV VALU instructions (v_fma_f32 chains, or v_exp_f32 with TRANS).  Three launches:
// MFMA waves alone (the others exit), VALU waves alone, both.  If the pipes overlap, both ~ max(alone); if the SIMD
// time-shares them, both ~ sum.  Also: ONE wave interleaving the two streams (independent), 4 waves per CU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/diag/mfma_valu_overlap.hip -o tools/diag/bin/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;

#ifndef PLAIN_VALU
#define PLAIN_VALU 0  // 1: the non-transcendental stream is v_fma_f32 kept scalar (inline asm) instead of whatever hipcc packs
#endif
template <bool TRANS>
__device__ __forceinline__ void valu_block(float (&v)[8]) {  // 8 independent instructions
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (TRANS) v[i] = __builtin_amdgcn_exp2f(v[i]);
    else if (PLAIN_VALU) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(1.0001f), "v"(0.5f));
    else v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
  }
}

typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;
// the same with v_mfma_f32_32x32x16_bf16 (8 passes each): 8 per iteration = the same 256 pipe cycles
template <bool TRANS>
__global__ __launch_bounds__(512) void split32_kernel(int mode, int iters, float* sink, unsigned long long* cyc) {
  const int wave = threadIdx.x >> 6;
  const bool mf = wave < 4;
  if (mf && !(mode & 1)) return;
  if (!mf && !(mode & 2)) return;
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x & 7)); b[i] = (__bf16)0.5f; }
  f32x16_t acc[2];
  for (int c = 0; c < 2; ++c)
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = -0.001f * (threadIdx.x + i);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (mf) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) valu_block<TRANS>(v);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int c = 0; c < 2; ++c) s += acc[c][0] + acc[c][5];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 12345.678f) sink[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

// mode bit 0: MFMA waves work; bit 1: VALU waves work
template <bool TRANS>
__global__ __launch_bounds__(512) void split_kernel(int mode, int iters, float* sink, unsigned long long* cyc) {
  const int wave = threadIdx.x >> 6;
  const bool mf = wave < 4;
  if (mf && !(mode & 1)) return;
  if (!mf && !(mode & 2)) return;
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x & 7)); b[i] = (__bf16)0.5f; }
  f32x4_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = -0.001f * (threadIdx.x + i);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (mf) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) valu_block<TRANS>(v);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 12345.678f) sink[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

// one wave issues both streams: per iteration 16 MFMAs and NVB blocks of 8 VALU instructions, interleaved by the compiler's
// order (sched_group_barrier: 1 MFMA then NVB*8/16 VALU)
template <bool TRANS, int NVB>
__global__ __launch_bounds__(256) void inter_kernel(int mode, int iters, float* sink, unsigned long long* cyc) {
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x & 7)); b[i] = (__bf16)0.5f; }
  f32x4_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = -0.001f * (threadIdx.x + i);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (mode & 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
    }
    if (mode & 2) {
#pragma unroll
      for (int u = 0; u < NVB; ++u) valu_block<TRANS>(v);
    }
    if (mode == 3) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, NVB * 8 / 16, 0);   // then VALU
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 12345.678f) sink[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

int main() {
  float* sink;
  unsigned long long *cyc, h[8];
  CK(hipMalloc(&sink, 4096));
  CK(hipMalloc(&cyc, 64));
  const int iters = 2000;
  auto run = [&](auto kernel, int threads, int mode, const char* what) {
    CK(hipMemset(cyc, 0, 64));
    kernel<<<256, threads>>>(mode, iters, sink, cyc);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost));
    printf("  %-34s wave0 %8.1f  wave4 %8.1f  cycles per iteration\n", what, (double)h[0] / iters, (double)h[4] / iters);
  };
  printf("two waves per SIMD; per iteration: MFMA waves 16 MFMA 16x16x32 (256 pipe cycles), VALU waves 64 v_fma_f32 (256 issue cycles)\n");
  run(split_kernel<false>, 512, 1, "MFMA waves alone");
  run(split_kernel<false>, 512, 2, "VALU (fma) waves alone");
  run(split_kernel<false>, 512, 3, "both");
  printf("... VALU waves 64 v_exp_f32 (quarter rate)\n");
  run(split_kernel<true>, 512, 2, "VALU (exp) waves alone");
  run(split_kernel<true>, 512, 3, "both");
  printf("... MFMA waves 8 MFMA 32x32x16 (the same 256 pipe cycles), VALU waves 64 v_fma_f32\n");
  run(split32_kernel<false>, 512, 1, "MFMA 32x32x16 waves alone");
  run(split32_kernel<false>, 512, 3, "both");
  run(split32_kernel<true>, 512, 3, "both (v_exp_f32)");
  printf("one wave per SIMD issuing both streams; per iteration 16 MFMA + 64 v_fma_f32\n");
  run(inter_kernel<false, 8>, 256, 1, "MFMA only");
  run(inter_kernel<false, 8>, 256, 2, "VALU only");
  run(inter_kernel<false, 8>, 256, 3, "interleaved");
  printf("... 16 MFMA + 32 v_exp_f32\n");
  run(inter_kernel<true, 4>, 256, 2, "VALU (exp) only");
  run(inter_kernel<true, 4>, 256, 3, "interleaved");
  return 0;
}
