// Diagnostic (not part of the product): the product's bf16 attention-forward kernel compiled with s_memtime phase
// stamps (the AVF_PHASE_* hooks in csrc/attn_bf16.hip).  Prints, averaged over active wavefronts, the cycles spent in
//   0 prologue (Q fragments, first K/V tile, barrier)   1 issue of the next tile's global loads
//   2 S = K Q^T MFMAs issued   3 online softmax   4 O += V^T P^T issued
//   5 commit (wait for the loads, write LDS)   6 barrier   7 epilogue
// Build + run on the GPU box, from the repo root:
//   P=multi-modal-multi-label-facial-action-unit-detection-with-transformer_amd
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -I$P/csrc tools/diag/attn_phases.hip -L$P/lib -lavformer_hip \
//         -Wl,-rpath,$PWD/$P/lib -o /tmp/attn_phases && /tmp/attn_phases 32 324 8 [1 = head-resident kernel, 0 = streaming]
// (slots of the head-resident kernel: 0 wait for the first tiles, 1 issue of all DMA pieces, 5 wait for the rest)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <vector>

__device__ uint64_t* g_phase_out;
__device__ __forceinline__ uint64_t avf_stamp() {
  uint64_t t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define AVF_PHASE_INIT() \
  uint64_t ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; \
  uint64_t ph_t = avf_stamp(); \
  const uint64_t ph_t0 = ph_t; \
  const uint64_t ph_w0 = wall_clock64()
#define AVF_PHASE_MARK(slot) \
  do { \
    __builtin_amdgcn_sched_barrier(0); \
    const uint64_t ph_n = avf_stamp(); \
    ph_acc[slot] += ph_n - ph_t; \
    ph_t = ph_n; \
    __builtin_amdgcn_sched_barrier(0); \
  } while (0)
#define AVF_PHASE_FLUSH() \
  do { \
    if ((threadIdx.x & 63) == 0) { \
      uint64_t* o = g_phase_out + ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 12; \
      for (int i = 0; i < 8; ++i) o[i] = ph_acc[i]; \
      o[8] = ph_t0; \
      o[9] = ph_t; \
      o[10] = wall_clock64() - ph_w0; /* 100 MHz */ \
    } \
  } while (0)

#include "attn_bf16.hip"

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32, N = argc > 2 ? atoi(argv[2]) : 324, H = argc > 3 ? atoi(argv[3]) : 8;
  const int dh = 64, I = H * dh;
  const size_t nq = (size_t)B * N * 3 * I;
  std::vector<uint16_t> h(nq);
  uint32_t st = 12345u;
  for (auto& v : h) { st = st * 1664525u + 1013904223u; v = (uint16_t)(0x3c00u + ((st >> 16) & 0x1ffu) + ((st >> 31) << 15)); }
  uint16_t *qkv, *o; float* lse; uint64_t* ph;
  hipMalloc(&qkv, nq * 2); hipMalloc(&o, (size_t)B * N * I * 2); hipMalloc(&lse, (size_t)B * H * N * 4);
  hipMemcpy(qkv, h.data(), nq * 2, hipMemcpyHostToDevice);
  const bool res = argc > 4 ? atoi(argv[4]) != 0 : (N <= 512);
  const int mode = argc > 5 ? atoi(argv[5]) : 0;  // 0 forward, 2 dK/dV (head-resident only; slots 2 = S, dP, softmax  4 = dV, dK MFMAs)
  uint16_t* dqkv; hipMalloc(&dqkv, nq * 2);
  const int grid = res ? B * H : ((N + 127) / 128) * B * H;
  hipMalloc(&ph, (size_t)grid * 16 * 12 * 8);
  hipMemset(ph, 0, (size_t)grid * 16 * 12 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(g_phase_out), &ph, sizeof(ph));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 5; ++it) {
    hipEventRecord(e0);
    if (res) {
      const int V = (N + 31) / 32, passes = V <= 12 ? 1 : (V + 7) / 8, W = (V + passes - 1) / passes;
      const size_t smem = (size_t)((N + 31) & ~31) * 256 + (mode == 2 ? (size_t)((N + 63) & ~63) * 8 : 0);
      const avf::bf16 *q = (const avf::bf16*)qkv, *g = (const avf::bf16*)o;
#define AVF_DIAG_LAUNCH(K, ...) \
  hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
  K<<<grid, W * 64, smem>>>(__VA_ARGS__)
      typedef avf::bf16 T;
      if (mode == 2 && passes > 1) { AVF_DIAG_LAUNCH((avf::attn_dkv_res_kernel<8, true, false>), q, g, lse, lse, (T*)dqkv, N, H); }
      else if (mode == 2 && W <= 8) { AVF_DIAG_LAUNCH((avf::attn_dkv_res_kernel<8, false, false>), q, g, lse, lse, (T*)dqkv, N, H); }
      else if (mode == 2) { AVF_DIAG_LAUNCH((avf::attn_dkv_res_kernel<12, false, false>), q, g, lse, lse, (T*)dqkv, N, H); }
      else if (passes > 1) { AVF_DIAG_LAUNCH((avf::attn_fwd_res_kernel<8, true, false>), q, (T*)o, lse, N, H); }
      else if (W <= 8) { AVF_DIAG_LAUNCH((avf::attn_fwd_res_kernel<8, false, false>), q, (T*)o, lse, N, H); }
      else { AVF_DIAG_LAUNCH((avf::attn_fwd_res_kernel<12, false, false>), q, (T*)o, lse, N, H); }
    } else {
      avf::attn_fwd_bf16_kernel<64><<<grid, 256>>>((const avf::bf16*)qkv, (avf::bf16*)o, lse, B, N, H, 0);
    }
    hipEventRecord(e1);
  }
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<uint64_t> out((size_t)grid * 16 * 12);
  hipMemcpy(out.data(), ph, out.size() * 8, hipMemcpyDeviceToHost);
  // s_memtime counters are not synchronised across the chip: spans are taken per workgroup (one CU)
  double acc[8] = {0}; double life = 0, span = 0, stagger = 0, wall = 0; size_t nw = 0, nb = 0;
  for (size_t g = 0; g < (size_t)grid; ++g) {
    uint64_t tmin = ~0ull, tmax = 0, smax = 0;
    for (int w = 0; w < 16; ++w) {
      const uint64_t* r = &out[(g * 16 + w) * 12];
      if (r[2] == 0) continue;  // wave without query rows (or absent)
      for (int i = 0; i < 8; ++i) acc[i] += (double)r[i];
      life += (double)(r[9] - r[8]); ++nw; wall += (double)r[10];
      if (r[8] < tmin) tmin = r[8];
      if (r[8] > smax) smax = r[8];
      if (r[9] > tmax) tmax = r[9];
    }
    if (tmax) { span += (double)(tmax - tmin); stagger += (double)(smax - tmin); ++nb; }
  }
  printf("kernel %.1f us (stamped build), %zu active waves, mean wave life %.0f ticks; per workgroup: first start -> last end "
         "%.0f ticks, first -> last wave start %.0f ticks; tick rate %.0f MHz\n", ms * 1e3, nw, life / nw, span / nb, stagger / nb, life / wall * 100.0);
  const char* names[8] = {"prologue", "issue loads", "S MFMAs", "softmax", "PV MFMAs", "commit", "barrier", "epilogue"};
  for (int i = 0; i < 8; ++i) printf("  %-12s %8.0f ticks/wave  %5.1f %%\n", names[i], acc[i] / nw, 100.0 * acc[i] / life);
  return 0;
}
