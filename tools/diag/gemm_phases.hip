// Diagnostic (not part of the product): the LDS-DMA NT GEMM main structure (128x128 tile, 8 waves of 64x32, 2 stages)
// with s_memtime stamps per workgroup: start, first tile landed, K-loop done, epilogue stores issued.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/diag/gemm_phases.hip -o /tmp/gemm_phases && /tmp/gemm_phases M N K
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

__device__ __forceinline__ int nt_off(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }
__device__ __forceinline__ void glds16(const void* g, char* l) { __builtin_amdgcn_global_load_lds((gptr_t*)g, (lptr_t*)l, 16, 0, 0); }
__device__ __forceinline__ uint64_t stamp() {
  uint64_t t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

constexpr int WM = 2, WN = 4, MI = 4, NI = 2, TK = 64;
constexpr int BMT = 128, BNT = 128, NW = 8, A_BYTES = BMT * 128, B_BYTES = BNT * 128, STAGE = A_BYTES + B_BYTES;
constexpr int A_INS = 2, B_INS = 2;

__global__ __launch_bounds__(512) void k(const uint16_t* A, const uint16_t* B, uint16_t* C, int M, int N, int K,
                                         int tiles_n, uint64_t* stamps) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN, li = lane & 15, lg = lane >> 4;
  const int wg = blockIdx.x;
  const int m0 = (wg / tiles_n) * BMT, n0 = (wg % tiles_n) * BNT;
  uint64_t t0 = stamp();
  const int lrow = lane >> 3, lchunk = (lane & 7) ^ (lane >> 3);
  const uint16_t* ga[A_INS];
  const uint16_t* gb[B_INS];
  for (int j = 0; j < A_INS; ++j) { int r = m0 + (wave * A_INS + j) * 8 + lrow; r = r < M ? r : M - 1; ga[j] = A + (int64_t)r * K + lchunk * 8; }
  for (int j = 0; j < B_INS; ++j) { int r = n0 + (wave * B_INS + j) * 8 + lrow; r = r < N ? r : N - 1; gb[j] = B + (int64_t)r * K + lchunk * 8; }
  auto stage = [&](int st, int k0) {
    char* sa = dsm + st * STAGE; char* sb = sa + A_BYTES;
    for (int j = 0; j < A_INS; ++j) glds16(ga[j] + k0, sa + (wave * A_INS + j) * 1024);
    for (int j = 0; j < B_INS; ++j) glds16(gb[j] + k0, sb + (wave * B_INS + j) * 1024);
  };
  f32x4_t acc[MI][NI];
  for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0, 0, 0, 0};
  const int nt = K / TK;
  stage(0, 0);
  uint64_t t1 = 0;
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t == 0) t1 = stamp();
    if (t + 1 < nt) stage(cur ^ 1, (t + 1) * TK);
    const char* sa = dsm + cur * STAGE; const char* sb = sa + A_BYTES;
    bf16x8_t fa[2][MI], fb[2][NI];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[ks][i] = *reinterpret_cast<const bf16x8_t*>(sa + nt_off(wm * 64 + i * 16 + li, ks * 4 + lg));
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[ks][j] = *reinterpret_cast<const bf16x8_t*>(sb + nt_off(wn * 32 + j * 16 + li, ks * 4 + lg));
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][j], fa[ks][i], acc[i][j], 0, 0, 0);
    cur ^= 1;
  }
  uint64_t t2 = stamp();
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wm * 64 + i * 16 + li;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wn * 32 + j * 16 + 4 * lg;
      if (m < M && n < N) {
        typedef __attribute__((ext_vector_type(2))) float f2; typedef __attribute__((ext_vector_type(2))) __bf16 b2;
        f2 lo = {acc[i][j][0], acc[i][j][1]}, hi = {acc[i][j][2], acc[i][j][3]};
        uint2 r; r.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, b2)); r.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, b2));
        *reinterpret_cast<uint2*>(C + (int64_t)m * N + n) = r;
      }
    }
  }
  uint64_t t3 = stamp();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  uint64_t t4 = stamp();
  if (tid == 0) {
    uint32_t xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    uint64_t* o = stamps + (int64_t)wg * 8;
    o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = t4; o[5] = ((uint64_t)(xcc & 15) << 32) | hwid;
  }
}

int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 10368, N = argc > 2 ? atoi(argv[2]) : 1536, K = argc > 3 ? atoi(argv[3]) : 512;
  size_t na = (size_t)M * K, nb = (size_t)N * K, nc = (size_t)M * N;
  std::vector<uint16_t> ha(na), hb(nb);
  for (auto& v : ha) v = 0x3c00 + (rand() & 0x3ff);
  for (auto& v : hb) v = 0x3c00 + (rand() & 0x3ff);
  uint16_t *A, *B, *C; uint64_t* S;
  const int tiles_m = (M + 127) / 128, tiles_n = (N + 127) / 128, nwg = tiles_m * tiles_n;
  hipMalloc(&A, na * 2); hipMalloc(&B, nb * 2); hipMalloc(&C, nc * 2); hipMalloc(&S, (size_t)nwg * 64);
  hipMemcpy(A, ha.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(B, hb.data(), nb * 2, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  for (int it = 0; it < 5; ++it) k<<<nwg, 512, 2 * STAGE>>>(A, B, C, M, N, K, tiles_n, S);
  hipDeviceSynchronize();
  std::vector<uint64_t> hs((size_t)nwg * 8);
  hipMemcpy(hs.data(), S, hs.size() * 8, hipMemcpyDeviceToHost);
  uint64_t tmin = ~0ull, tmax = 0;
  for (int w = 0; w < nwg; ++w) { tmin = std::min(tmin, hs[w * 8]); tmax = std::max(tmax, hs[w * 8 + 4]); }
  auto pct = [&](int a, int b, const char* name) {
    std::vector<double> v;
    for (int w = 0; w < nwg; ++w) v.push_back((double)(hs[w * 8 + b] - hs[w * 8 + a]));
    std::sort(v.begin(), v.end());
    double s = 0; for (double x : v) s += x;
    printf("%-28s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f cycles\n", name, s / v.size(), v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10]);
  };
  printf("M=%d N=%d K=%d  workgroups=%d  kernel span (first start -> last store retired) = %llu cycles\n", M, N, K, nwg, (unsigned long long)(tmax - tmin));
  pct(0, 1, "start -> first tile landed");
  pct(1, 2, "K loop");
  pct(2, 3, "epilogue issue");
  pct(3, 4, "store drain");
  pct(0, 4, "workgroup lifetime");
  // start-time histogram: how many rounds
  std::vector<double> st; for (int w = 0; w < nwg; ++w) st.push_back((double)(hs[w * 8] - tmin)); std::sort(st.begin(), st.end());
  printf("start offsets: p25 %.0f p50 %.0f p75 %.0f p95 %.0f max %.0f\n", st[nwg / 4], st[nwg / 2], st[nwg * 3 / 4], st[nwg * 95 / 100], st.back());
  return 0;
}
