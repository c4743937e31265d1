// store_pattern.hip - what does the lane -> address map of a 16-byte-per-lane global store / load cost on gfx950?
// A wave writes (reads) tiles of 16 rows x 64 bytes of a row-major bf16 matrix [M][N]:
//   pattern 0 "fragment": lane (li = lane & 15, lg = lane >> 4) -> row li, 16-byte chunk lg    (the MFMA accumulator layout after
//                          store_pair16: the four lanes of a TA quad touch four different rows)
//   pattern 1 "quad":     lane -> row lane >> 2, chunk lane & 3                                  (a quad = 64 contiguous bytes)
//   pattern 2 "line":     8 rows x 128 bytes: lane -> row lane >> 3, chunk lane & 7             (two quads = one 128-byte line)
//   pattern 3 "linear":   1 KiB contiguous per wave-instruction
// Build: hipcc --offload-arch=gfx950 -O3 tools/diag/store_pattern.hip -o tools/diag/bin/store_pattern ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int PAT, bool LOAD>
__global__ __launch_bounds__(512) void k(uint4* __restrict__ mat, int M, int N /* bf16 columns */, uint4* __restrict__ sink) {
  // the matrix as 64-byte column groups: a wave owns column group (wave id within WG + 8 * (blockIdx % (N/256))) like the
  // persistent GEMM: 8 waves x 32 columns = 256 columns per workgroup, row tiles strided over the workgroups of a panel
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int P = N / 256, panel = blockIdx.x % P, grp = blockIdx.x / P, G = gridDim.x / P;
  const int row_u4 = N / 8;  // uint4 per row
  uint4 acc = make_uint4(0, 0, 0, 0);
  const uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
  if (PAT == 3) {
    // linear: the same number of bytes per workgroup, fully contiguous
    const size_t total = (size_t)M * row_u4;
    const size_t per = total / gridDim.x;
    uint4* base = mat + per * blockIdx.x;
    for (size_t i = threadIdx.x; i < per; i += 512) {
      if (LOAD) { uint4 t = base[i]; acc.x ^= t.x; acc.y ^= t.y; acc.z ^= t.z; acc.w ^= t.w; }
      else base[i] = v;
    }
  } else {
    for (int r0 = grp * 16; r0 < M; r0 += G * 16) {
      int row, chunk;
      if (PAT == 0) { row = lane & 15; chunk = lane >> 4; }
      else if (PAT == 1) { row = lane >> 2; chunk = lane & 3; }
      else { row = lane >> 3; chunk = lane & 7; }
      if (PAT == 2) {
        // 8 rows x 128 B: the wave pair (2w, 2w+1) would share a line in the GEMM; here one wave covers 128 B of 8 rows, two
        // instructions per 16 rows
        for (int h = 0; h < 2; ++h) {
          uint4* p = mat + (size_t)(r0 + h * 8 + row) * row_u4 + panel * 32 + (wave >> 1) * 8 + chunk;
          if ((wave & 1) == h) {  // each wave of the pair takes one 8-row half: same bytes per wave overall
            if (LOAD) { uint4 t = *p; acc.x ^= t.x; acc.y ^= t.y; acc.z ^= t.z; acc.w ^= t.w; }
            else *p = v;
          }
        }
      } else {
        uint4* p = mat + (size_t)(r0 + row) * row_u4 + panel * 32 + wave * 4 + chunk;
        if (LOAD) { uint4 t = *p; acc.x ^= t.x; acc.y ^= t.y; acc.z ^= t.z; acc.w ^= t.w; }
        else *p = v;
      }
    }
  }
  if (LOAD && acc.x == 0x12345678u) sink[threadIdx.x] = acc;
}

template <int PAT, bool LOAD>
float run(uint4* mat, int M, int N, uint4* sink, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) k<PAT, LOAD><<<256, 512>>>(mat, M, N, sink);
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) k<PAT, LOAD><<<256, 512>>>(mat, M, N, sink);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters * 1e3f;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 16384;
  const char* names[4] = {"fragment (16 rows x 64 B, quad = 4 rows)", "quad     (16 rows x 64 B, quad = 64 B)  ", "line     ( 8 rows x 128 B)              ", "linear   (1 KiB contiguous)             "};
  for (int N : {512, 1024, 1536}) {
    const size_t bytes = (size_t)M * N * 2;
    // several matrices in rotation (320 MB+): every launch writes / reads lines that left the caches
    const int NB = (int)((size_t)640 * 1024 * 1024 / bytes) + 1;
    std::vector<uint4*> bufs(NB);
    for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 1, bytes)); }
    uint4* sink;
    CK(hipMalloc(&sink, 8192));
    printf("M=%d N=%d (%.1f MB)\n", M, N, bytes / 1e6);
    for (int pat = 0; pat < 4; ++pat) {
      float ts = 0, tl = 0, ts_hot, tl_hot;
      const int R = 20;
      // cold: rotate buffers
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      auto launch = [&](bool load, uint4* m) {
        switch (pat * 2 + (load ? 1 : 0)) {
          case 0: k<0, false><<<256, 512>>>(m, M, N, sink); break;
          case 1: k<0, true><<<256, 512>>>(m, M, N, sink); break;
          case 2: k<1, false><<<256, 512>>>(m, M, N, sink); break;
          case 3: k<1, true><<<256, 512>>>(m, M, N, sink); break;
          case 4: k<2, false><<<256, 512>>>(m, M, N, sink); break;
          case 5: k<2, true><<<256, 512>>>(m, M, N, sink); break;
          case 6: k<3, false><<<256, 512>>>(m, M, N, sink); break;
          default: k<3, true><<<256, 512>>>(m, M, N, sink); break;
        }
      };
      for (int ld = 0; ld < 2; ++ld) {
        for (int i = 0; i < NB; ++i) launch(ld, bufs[i]);
        CK(hipEventRecord(e0));
        for (int i = 0; i < R; ++i) launch(ld, bufs[i % NB]);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        (ld ? tl : ts) = ms / R * 1e3f;
        CK(hipEventRecord(e0));
        for (int i = 0; i < R; ++i) launch(ld, bufs[0]);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        (ld ? tl_hot : ts_hot) = ms / R * 1e3f;
      }
      printf("  %s store %6.1f us (%5.2f TB/s; same buffer %6.1f us)   load %6.1f us (%5.2f TB/s; same buffer %6.1f us)\n", names[pat], ts,
             bytes / ts / 1e6, ts_hot, tl, bytes / tl / 1e6, tl_hot);
    }
    for (auto b : bufs) CK(hipFree(b));
    CK(hipFree(sink));
  }
  return 0;
}
