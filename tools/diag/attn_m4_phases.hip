// Diagnostic (not part of the product): the merged attention-backward kernel (csrc/attn_bwd_merged.hip) compiled with
// s_memtime phase stamps.  Prints, averaged over the wavefronts, the cycles spent in
//   0 prologue   1 top of a slice (DMA issue, O chunk, dQ rows of the previous slice)   2 the KB key blocks
//   3 dQ partial -> reduction tile   4 wait for the next slice + its delta   5 barrier   6 epilogue (dK, dV stores)
// Build (CPU container or GPU box) + run on the GPU box, from the repo root:
//   P=multi-modal-multi-label-facial-action-unit-detection-with-transformer_amd
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -Iinclude -I$P/csrc tools/diag/attn_m4_phases.hip \
//         -L$P/lib -lavformer_hip -Wl,-rpath,'$ORIGIN/../../../'$P/lib -o tools/diag/bin/attn_m4_phases
//   tools/diag/bin/attn_m4_phases 32 512 8
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
      uint64_t* o = g_phase_out + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 12; \
      for (int i = 0; i < 8; ++i) o[i] = ph_acc[i]; \
      o[8] = ph_t0; \
      o[9] = ph_t; \
      o[10] = wall_clock64() - ph_w0; /* 100 MHz */ \
    } \
  } while (0)

#include "attn_bwd_merged.hip"

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32, N = argc > 2 ? atoi(argv[2]) : 512, H = argc > 3 ? atoi(argv[3]) : 8;
  const int dh = 64, I = H * dh;
  const size_t nq = (size_t)B * N * 3 * I, no = (size_t)B * N * I;
  std::vector<uint16_t> h(nq);
  uint32_t st = 12345u;
  for (auto& v : h) { st = st * 1664525u + 1013904223u; v = (uint16_t)(0x3c00u + ((st >> 16) & 0x1ffu) + ((st >> 31) << 15)); }
  uint16_t *qkv, *o, *d_o, *dqkv; float* lse; uint64_t* ph;
  hipMalloc(&qkv, nq * 2); hipMalloc(&dqkv, nq * 2); hipMalloc(&o, no * 2); hipMalloc(&d_o, no * 2);
  hipMalloc(&lse, (size_t)B * H * N * 4);
  hipMemcpy(qkv, h.data(), nq * 2, hipMemcpyHostToDevice);
  hipMemcpy(o, h.data(), no * 2, hipMemcpyHostToDevice);
  hipMemcpy(d_o, h.data() + no, no * 2, hipMemcpyHostToDevice);
  std::vector<float> hl((size_t)B * H * N, 12.0f);
  hipMemcpy(lse, hl.data(), hl.size() * 4, hipMemcpyHostToDevice);
  const int grid = B * H;
  hipMalloc(&ph, (size_t)grid * 4 * 12 * 8);
  hipMemset(ph, 0, (size_t)grid * 4 * 12 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(g_phase_out), &ph, sizeof(ph));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 5; ++it) {
    hipEventRecord(e0);
    int rc = avf::attn_bwd_merged(nullptr, (const avf::bf16*)qkv, (const avf::bf16*)o, (const avf::bf16*)d_o, lse, (avf::bf16*)dqkv, B, N, H, 0);
    if (rc) { printf("launch failed: %s\n", avf_last_error()); return 1; }
    hipEventRecord(e1);
  }
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<uint64_t> out((size_t)grid * 4 * 12);
  hipMemcpy(out.data(), ph, out.size() * 8, hipMemcpyDeviceToHost);
  double acc[8] = {0}; double life = 0, wall = 0; size_t nw = 0;
  for (size_t g = 0; g < (size_t)grid; ++g)
    for (int w = 0; w < 4; ++w) {
      const uint64_t* r = &out[(g * 4 + w) * 12];
      for (int i = 0; i < 8; ++i) acc[i] += (double)r[i];
      life += (double)(r[9] - r[8]); ++nw; wall += (double)r[10];
    }
  printf("B=%d N=%d H=%d: kernel %.1f us (stamped build), mean wave life %.0f ticks, tick rate %.0f MHz\n", B, N, H, ms * 1e3, life / nw, life / wall * 100.0);
  const char* names[8] = {"prologue", "slice top", "loads + A(0)", "dQ job + B(0)", "stages A(1..)", "delta+barrier", "last dQ + epilogue", "phases C"};
  const int NS = (N + 31) / 32;
  for (int i = 0; i < 8; ++i) printf("  %-12s %9.0f ticks/wave  %5.1f %%   (%.0f per slice)\n", names[i], acc[i] / nw, 100.0 * acc[i] / life, acc[i] / nw / NS);
  return 0;
}
