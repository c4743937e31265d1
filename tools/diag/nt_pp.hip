// Development harness (not part of the product): candidate structures for the bf16 NT GEMM C[M,N] = A[M,K] B[N,K]^T with a
// plain bf16 epilogue, timed back to back and checked against a host reference on sampled elements.
//   K0  the shipped structure: 128 x 128 tile, 8 waves of 64 x 32, 2 LDS stages, one barrier per K-step, 2 workgroups / CU
//   K0A K0 with inline-asm fragment reads (variant 5: one wait, 6: split wait) - what the product kernel now does
//   K0B K0A with a 3- / 4-slot ring (variants 7 / 8; one workgroup per CU): 15-20 % slower
//   K1  256 x 128 tile, 8 waves of 64 x 64 in two groups that alternate LOAD and COMPUTE phases (ping-pong), 3-slot
//       LDS-DMA ring, fragment reads by inline asm (no compiler-inserted vmcnt(0)), 1 workgroup / CU
// Build here (cross-compile), run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/diag/nt_pp.hip -o tools/diag/bin/nt_pp
//   tools/diag/bin/nt_pp M N K [variant] [stamps]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
typedef __attribute__((address_space(3))) char lds_char;

__device__ __forceinline__ int nt_off(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }
__device__ __forceinline__ void glds16(const void* g, char* l) { __builtin_amdgcn_global_load_lds((gptr_t*)g, (lptr_t*)l, 16, 0, 0); }
__device__ __forceinline__ uint64_t stamp() {
  uint64_t t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int OFF>
__device__ __forceinline__ bf16x8_t lds_b128(uint32_t addr) {
  bf16x8_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ void store_bf16x4(uint16_t* p, f32x4_t a) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  f2 lo = {a[0], a[1]}, hi = {a[2], a[3]};
  uint2 r;
  r.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, b2));
  r.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, b2));
  *reinterpret_cast<uint2*>(p) = r;
}
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = id & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
}

// ------------------------------------------------------------------------------------------------ K0 (shipped structure)
namespace k0 {
constexpr int WM = 2, WN = 4, MI = 4, NI = 2, TK = 64;
constexpr int BMT = 128, BNT = 128, A_BYTES = BMT * 128, B_BYTES = BNT * 128, STAGE = A_BYTES + B_BYTES;
constexpr int A_INS = 2, B_INS = 2;
__global__ __launch_bounds__(512) void k(const uint16_t* A, const uint16_t* B, uint16_t* C, int M, int N, int K, int tiles_n,
                                         int nwg) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN, li = lane & 15, lg = lane >> 4;
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int m0 = (wg / tiles_n) * BMT, n0 = (wg % tiles_n) * BNT;
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
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
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
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wm * 64 + i * 16 + li;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wn * 32 + j * 16 + 4 * lg;
      if (m < M && n < N) store_bf16x4(C + (int64_t)m * N + n, acc[i][j]);
    }
  }
}
}  // namespace k0

// ------------------------------------------------------------------------------------------------ K0A (K0, asm fragment reads)
// the shipped structure with the fragment reads issued as inline asm: the compiler no longer drains vmcnt before the first
// ds_read of a K-step, so the DMA of tile t+1 really overlaps the reads and MFMAs of tile t inside one workgroup
namespace k0a {
using namespace k0;
template <int SPLIT>
__global__ __launch_bounds__(512) void k(const uint16_t* A, const uint16_t* B, uint16_t* C, int M, int N, int K, int tiles_n,
                                         int nwg) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN, li = lane & 15, lg = lane >> 4;
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int m0 = (wg / tiles_n) * BMT, n0 = (wg % tiles_n) * BNT;
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
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)dsm;
  uint32_t abase[2], bbase[2];
  for (int ks = 0; ks < 2; ++ks) {
    abase[ks] = lds0 + nt_off(wm * 64 + li, ks * 4 + lg);
    bbase[ks] = lds0 + A_BYTES + nt_off(wn * 32 + li, ks * 4 + lg);
  }
  const int nt = K / TK;
  stage(0, 0);
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 1 < nt) stage(cur ^ 1, (t + 1) * TK);
    const uint32_t so = cur * STAGE;
    bf16x8_t fa[2][MI], fb[2][NI];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      fa[ks][0] = lds_b128<0>(abase[ks] + so); fa[ks][1] = lds_b128<2048>(abase[ks] + so);
      fa[ks][2] = lds_b128<4096>(abase[ks] + so); fa[ks][3] = lds_b128<6144>(abase[ks] + so);
      fb[ks][0] = lds_b128<0>(bbase[ks] + so); fb[ks][1] = lds_b128<2048>(bbase[ks] + so);
    }
    if (SPLIT) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0][j], fa[0][i], acc[i][j], 0, 0, 0);
    if (SPLIT) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1][j], fa[1][i], acc[i][j], 0, 0, 0);
    cur ^= 1;
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wm * 64 + i * 16 + li;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wn * 32 + j * 16 + 4 * lg;
      if (m < M && n < N) store_bf16x4(C + (int64_t)m * N + n, acc[i][j]);
    }
  }
}
}  // namespace k0a

// ------------------------------------------------------------------------------------------------ K0B (K0A with an NS-slot ring)
// the same tile and waves with NS LDS slots (counted vmcnt: tiles t+1 .. t+NS-2 stay in flight): does a deeper DMA ring pay now
// that the compiler no longer drains it?  NS = 3: 96 KB, one workgroup per CU; NS = 2 is K0A
namespace k0b {
using namespace k0;
template <int NS>
__global__ __launch_bounds__(512) void k(const uint16_t* A, const uint16_t* B, uint16_t* C, int M, int N, int K, int tiles_n,
                                         int nwg) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN, li = lane & 15, lg = lane >> 4;
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int m0 = (wg / tiles_n) * BMT, n0 = (wg % tiles_n) * BNT;
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
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)dsm;
  uint32_t abase[2], bbase[2];
  for (int ks = 0; ks < 2; ++ks) {
    abase[ks] = lds0 + nt_off(wm * 64 + li, ks * 4 + lg);
    bbase[ks] = lds0 + A_BYTES + nt_off(wn * 32 + li, ks * 4 + lg);
  }
  constexpr int INS = A_INS + B_INS;
  const int nt = K / TK;
  for (int i = 0; i < NS - 1; ++i)
    if (i < nt) stage(i, i * TK);
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    const int ahead = (nt - 1 - t) < (NS - 2) ? (nt - 1 - t) : (NS - 2);
    if (ahead >= 2) wait_vmcnt<2 * INS>();
    else if (ahead == 1) wait_vmcnt<INS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (t + NS - 1 < nt) {
      int slot = cur + NS - 1;
      slot = slot >= NS ? slot - NS : slot;
      stage(slot, (t + NS - 1) * TK);
    }
    const uint32_t so = cur * STAGE;
    bf16x8_t fa[2][MI], fb[2][NI];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      fa[ks][0] = lds_b128<0>(abase[ks] + so); fa[ks][1] = lds_b128<2048>(abase[ks] + so);
      fa[ks][2] = lds_b128<4096>(abase[ks] + so); fa[ks][3] = lds_b128<6144>(abase[ks] + so);
      fb[ks][0] = lds_b128<0>(bbase[ks] + so); fb[ks][1] = lds_b128<2048>(bbase[ks] + so);
    }
    asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0][j], fa[0][i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1][j], fa[1][i], acc[i][j], 0, 0, 0);
    cur = (cur + 1 == NS) ? 0 : cur + 1;
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wm * 64 + i * 16 + li;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wn * 32 + j * 16 + 4 * lg;
      if (m < M && n < N) store_bf16x4(C + (int64_t)m * N + n, acc[i][j]);
    }
  }
}
}  // namespace k0b

// ------------------------------------------------------------------------------------------------ K1 (ping-pong, 256 x 128)
// PH = compute phases per K-tile per wave (1: 32 MFMAs per phase, 2: 16 per phase)
namespace k1 {
constexpr int TK = 64, BMT = 256, BNT = 128;
constexpr int A_BYTES = BMT * 128, B_BYTES = BNT * 128, STAGE = A_BYTES + B_BYTES;  // 32 + 16 KiB
constexpr int A_INS = 4, B_INS = 2, INS = A_INS + B_INS;
constexpr int NSLOT = 3;

template <int PH, bool STAMPS>
__global__ __launch_bounds__(512) void k(const uint16_t* A, const uint16_t* B, uint16_t* C, int M, int N, int K, int tiles_n,
                                         int nwg, uint64_t* stamps) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, li = lane & 15, lg = lane >> 4;
  const int half = wave >> 2;
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int m0 = (wg / tiles_n) * BMT, n0 = (wg % tiles_n) * BNT;
  uint64_t tS = 0, tL = 0, tI = 0, tV = 0, tB = 0, tC = 0;  // accumulated phase cycles (wave 0 / wave 4 of a workgroup)
  uint64_t t_start = STAMPS ? stamp() : 0;

  const int lrow = lane >> 3, lchunk = (lane & 7) ^ (lane >> 3);
  const uint16_t* ga[A_INS];
  const uint16_t* gb[B_INS];
#pragma unroll
  for (int j = 0; j < A_INS; ++j) { int r = m0 + (wave * A_INS + j) * 8 + lrow; r = r < M ? r : M - 1; ga[j] = A + (int64_t)r * K + lchunk * 8; }
#pragma unroll
  for (int j = 0; j < B_INS; ++j) { int r = n0 + (wave * B_INS + j) * 8 + lrow; r = r < N ? r : N - 1; gb[j] = B + (int64_t)r * K + lchunk * 8; }
  auto stage = [&](int st, int k0) {
    char* sa = dsm + st * STAGE; char* sb = sa + A_BYTES;
#pragma unroll
    for (int j = 0; j < A_INS; ++j) glds16(ga[j] + k0, sa + (wave * A_INS + j) * 1024);
#pragma unroll
    for (int j = 0; j < B_INS; ++j) glds16(gb[j] + k0, sb + (wave * B_INS + j) * 1024);
  };
  // fragment addresses: row r = base + 16 i + li (r & 7 = li & 7), chunk (ks*4 + lg) ^ (li & 7); ks = 1 toggles bit 2
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)dsm;
  const int c0 = lg ^ (li & 7), c1 = c0 ^ 4;
  const uint32_t a_ks0 = (uint32_t)((wm * 64 + li) * 128 + (c0 << 4)), a_ks1 = (uint32_t)((wm * 64 + li) * 128 + (c1 << 4));
  const uint32_t b_ks0 = (uint32_t)(A_BYTES + (wn * 64 + li) * 128 + (c0 << 4)), b_ks1 = (uint32_t)(A_BYTES + (wn * 64 + li) * 128 + (c1 << 4));

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0, 0, 0, 0};

  const int nt = K / TK;
  if (nt > 0) stage(0, 0);
  if (nt > 1) stage(1, TK);
  if (nt > 1) wait_vmcnt<INS>(); else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (half == 1) __builtin_amdgcn_s_barrier();
  uint64_t t_loop = STAMPS ? stamp() : 0;
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    const uint32_t sb_ = lds0 + (uint32_t)cur * STAGE;
    bf16x8_t fa[2][4], fb[2][4];
    uint64_t s0 = STAMPS ? stamp() : 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (PH == 2 && ks == 1) break;
      const uint32_t aa = sb_ + (ks ? a_ks1 : a_ks0), bb = sb_ + (ks ? b_ks1 : b_ks0);
      fa[ks][0] = lds_b128<0>(aa); fa[ks][1] = lds_b128<2048>(aa); fa[ks][2] = lds_b128<4096>(aa); fa[ks][3] = lds_b128<6144>(aa);
      fb[ks][0] = lds_b128<0>(bb); fb[ks][1] = lds_b128<2048>(bb); fb[ks][2] = lds_b128<4096>(bb); fb[ks][3] = lds_b128<6144>(bb);
    }
    uint64_t s1 = STAMPS ? stamp() : 0;
    if (t + 2 < nt) {
      int slot = cur + 2; slot = slot >= NSLOT ? slot - NSLOT : slot;
      if (PH == 2) {  // half of the pieces now, half in the second load phase
        char* sa = dsm + slot * STAGE; char* sbp = sa + A_BYTES;
        glds16(ga[0] + (t + 2) * TK, sa + (wave * A_INS + 0) * 1024);
        glds16(ga[1] + (t + 2) * TK, sa + (wave * A_INS + 1) * 1024);
        glds16(gb[0] + (t + 2) * TK, sbp + (wave * B_INS + 0) * 1024);
      } else {
        stage(slot, (t + 2) * TK);
      }
    }
    uint64_t s2 = STAMPS ? stamp() : 0;
    if (PH == 1) {
      if (t + 2 < nt) wait_vmcnt<INS>(); else wait_vmcnt<0>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint64_t s3 = STAMPS ? stamp() : 0;
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    uint64_t s4 = STAMPS ? stamp() : 0;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0][j], fa[0][i], acc[i][j], 0, 0, 0);
    if (PH == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1][j], fa[1][i], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    uint64_t s5 = STAMPS ? stamp() : 0;
    if (STAMPS) { tL += s1 - s0; tI += s2 - s1; tV += s3 - s2; tB += s4 - s3; tC += s5 - s4; }
    if (PH == 2) {
      // second phase of the K-tile: ks = 1 fragments, the other half of the DMA pieces, then the waits for tile t+1
      const uint32_t aa = sb_ + a_ks1, bb = sb_ + b_ks1;
      fa[1][0] = lds_b128<0>(aa); fa[1][1] = lds_b128<2048>(aa); fa[1][2] = lds_b128<4096>(aa); fa[1][3] = lds_b128<6144>(aa);
      fb[1][0] = lds_b128<0>(bb); fb[1][1] = lds_b128<2048>(bb); fb[1][2] = lds_b128<4096>(bb); fb[1][3] = lds_b128<6144>(bb);
      if (t + 2 < nt) {
        int slot = cur + 2; slot = slot >= NSLOT ? slot - NSLOT : slot;
        char* sa = dsm + slot * STAGE; char* sbp = sa + A_BYTES;
        glds16(ga[2] + (t + 2) * TK, sa + (wave * A_INS + 2) * 1024);
        glds16(ga[3] + (t + 2) * TK, sa + (wave * A_INS + 3) * 1024);
        glds16(gb[1] + (t + 2) * TK, sbp + (wave * B_INS + 1) * 1024);
        wait_vmcnt<INS>();
      } else {
        wait_vmcnt<0>();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1][j], fa[1][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    cur = (cur + 1 == NSLOT) ? 0 : cur + 1;
  }
  if (half == 0) __builtin_amdgcn_s_barrier();
  uint64_t t_epi = STAMPS ? stamp() : 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + li;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * lg;
      if (m < M && n < N) store_bf16x4(C + (int64_t)m * N + n, acc[i][j]);
    }
  }
  if (STAMPS) {
    uint64_t t_end = stamp();
    if (lane == 0 && (wave == 0 || wave == 4)) {
      uint64_t* o = stamps + ((int64_t)wg * 2 + half) * 16;
      o[0] = t_start; o[1] = t_loop; o[2] = t_epi; o[3] = t_end; o[4] = tL; o[5] = tI; o[6] = tV; o[7] = tB; o[8] = tC; o[9] = tS;
    }
  }
}
}  // namespace k1


// ------------------------------------------------------------------------------------------------ K2 (full-row tile + LayerNorm)
// x_new = A W^T + bias + x_res ; h = LayerNorm(x_new) in ONE kernel: a workgroup owns 64 full rows (N = 512 = 8 waves x 64
// columns), so the row statistics are complete inside it.  Ping-pong halves, 2-slot ring; A (8 KB / K-tile) is shared and
// issued by the first half, every wave streams its OWN 64 weight rows (B is private per wave: no cross-wave hazard on it).
namespace k2 {
constexpr int TK = 64, BMT = 64, BNT = 512;
constexpr int A_BYTES = BMT * 128, B_BYTES = BNT * 128, STAGE = A_BYTES + B_BYTES;  // 8 + 64 KiB
constexpr int NSLOT = 2;

template <bool RES16>
__global__ __launch_bounds__(512) void k(const uint16_t* A, const uint16_t* B, const float* bias, const void* res,
                                         const float* gamma, const float* beta, void* X, uint16_t* H, float* mean,
                                         float* rstd, int M, int K, float eps) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int half = wave >> 2;
  const int m0 = blockIdx.x * BMT, n0 = wave * 64;

  const int lrow = lane >> 3, lchunk = (lane & 7) ^ (lane >> 3);
  const uint16_t* ga[2];
  const uint16_t* gb[8];
#pragma unroll
  for (int j = 0; j < 2; ++j) { int r = m0 + ((wave & 3) * 2 + j) * 8 + lrow; r = r < M ? r : M - 1; ga[j] = A + (int64_t)r * K + lchunk * 8; }
#pragma unroll
  for (int j = 0; j < 8; ++j) gb[j] = B + (int64_t)(n0 + j * 8 + lrow) * K + lchunk * 8;
  auto stage = [&](int st, int k0) {
    char* sa = dsm + st * STAGE; char* sb = sa + A_BYTES;
    if (half == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) glds16(ga[j] + k0, sa + ((wave & 3) * 2 + j) * 1024);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) glds16(gb[j] + k0, sb + (wave * 8 + j) * 1024);
  };
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)dsm;
  const int c0 = lg ^ (li & 7), c1 = c0 ^ 4;
  const uint32_t a_ks0 = (uint32_t)(li * 128 + (c0 << 4)), a_ks1 = (uint32_t)(li * 128 + (c1 << 4));
  const uint32_t b_ks0 = (uint32_t)(A_BYTES + (n0 + li) * 128 + (c0 << 4)), b_ks1 = (uint32_t)(A_BYTES + (n0 + li) * 128 + (c1 << 4));

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0, 0, 0, 0};

  const int nt = K / TK;
  stage(0, 0);
  // residual tile of this wave (rows 16 i + li, columns n0 + 16 j + 4 lg ..): loaded now, consumed in the epilogue
  f32x4_t rres[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + 16 * i + li; m = m < M ? m : M - 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + 16 * j + 4 * lg;
      if (RES16) {
        const uint2 r = *reinterpret_cast<const uint2*>((const uint16_t*)res + (int64_t)m * BNT + n);
        rres[i][j] = f32x4_t{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                             __uint_as_float(r.y & 0xffff0000u)};
      } else {
        rres[i][j] = *reinterpret_cast<const f32x4_t*>((const float*)res + (int64_t)m * BNT + n);
      }
    }
  }
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (half == 1) __builtin_amdgcn_s_barrier();
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    const uint32_t sb_ = lds0 + (uint32_t)cur * STAGE;
    bf16x8_t fa[2][4], fb[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint32_t aa = sb_ + (ks ? a_ks1 : a_ks0), bb = sb_ + (ks ? b_ks1 : b_ks0);
      fa[ks][0] = lds_b128<0>(aa); fa[ks][1] = lds_b128<2048>(aa); fa[ks][2] = lds_b128<4096>(aa); fa[ks][3] = lds_b128<6144>(aa);
      fb[ks][0] = lds_b128<0>(bb); fb[ks][1] = lds_b128<2048>(bb); fb[ks][2] = lds_b128<4096>(bb); fb[ks][3] = lds_b128<6144>(bb);
    }
    if (t + 1 < nt) stage(cur ^ 1, (t + 1) * TK);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][j], fa[ks][i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    wait_vmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    cur ^= 1;
  }
  if (half == 0) __builtin_amdgcn_s_barrier();
  // ---- epilogue: x = acc + bias + residual; row statistics over the 8 waves through LDS; h = LN(x) ----------------------
  float* red = reinterpret_cast<float*>(dsm);  // [8 waves][64 rows] (the ring is idle now: every wave passed the last barrier)
  float4 bj[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bj[j] = *reinterpret_cast<const float4*>(bias + n0 + 16 * j + 4 * lg);
  float rs_[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[i][j][0] += bj[j].x + rres[i][j][0]; acc[i][j][1] += bj[j].y + rres[i][j][1];
      acc[i][j][2] += bj[j].z + rres[i][j][2]; acc[i][j][3] += bj[j].w + rres[i][j][3];
      s += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
    }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    rs_[i] = s;
  }
  if (lg == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave * 64 + 16 * i + li] = rs_[i];
  }
  __syncthreads();
  float mu[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += red[w * 64 + 16 * i + li];
    mu[i] = s * (1.0f / BNT);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = acc[i][j][r] - mu[i]; q += d * d; }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    rs_[i] = q;
  }
  if (lg == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave * 64 + 16 * i + li] = rs_[i];
  }
  __syncthreads();
  float4 gj[4], btj[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    gj[j] = *reinterpret_cast<const float4*>(gamma + n0 + 16 * j + 4 * lg);
    btj[j] = *reinterpret_cast<const float4*>(beta + n0 + 16 * j + 4 * lg);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float q = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) q += red[w * 64 + 16 * i + li];
    const float rstdv = 1.0f / sqrtf(q * (1.0f / BNT) + eps);
    const int m = m0 + 16 * i + li;
    if (m >= M) continue;
    if (wave == 0 && lg == 0) { mean[m] = mu[i]; rstd[m] = rstdv; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + 16 * j + 4 * lg;
      if (RES16) store_bf16x4((uint16_t*)X + (int64_t)m * BNT + n, acc[i][j]);
      else *reinterpret_cast<f32x4_t*>((float*)X + (int64_t)m * BNT + n) = acc[i][j];
      f32x4_t h;
      h[0] = (acc[i][j][0] - mu[i]) * rstdv * gj[j].x + btj[j].x; h[1] = (acc[i][j][1] - mu[i]) * rstdv * gj[j].y + btj[j].y;
      h[2] = (acc[i][j][2] - mu[i]) * rstdv * gj[j].z + btj[j].z; h[3] = (acc[i][j][3] - mu[i]) * rstdv * gj[j].w + btj[j].w;
      store_bf16x4(H + (int64_t)m * BNT + n, h);
    }
  }
}
}  // namespace k2

static float ref_elem(const std::vector<uint16_t>& a, const std::vector<uint16_t>& b, int K, int m, int n) {
  double s = 0;
  for (int k = 0; k < K; ++k) {
    uint32_t x = (uint32_t)a[(size_t)m * K + k] << 16, y = (uint32_t)b[(size_t)n * K + k] << 16;
    float fx, fy;
    memcpy(&fx, &x, 4); memcpy(&fy, &y, 4);
    s += (double)fx * fy;
  }
  return (float)s;
}

int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 1536, K = argc > 3 ? atoi(argv[3]) : 512;
  int variant = argc > 4 ? atoi(argv[4]) : 1, want_stamps = argc > 5 ? atoi(argv[5]) : 0;
  size_t na = (size_t)M * K, nb = (size_t)N * K, nc = (size_t)M * N;
  std::vector<uint16_t> ha(na), hb(nb), hc(nc);
  srand(1);
  auto rnd = [] { float f = (float)rand() / RAND_MAX * 2.f - 1.f; uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); };
  for (auto& v : ha) v = rnd();
  for (auto& v : hb) v = rnd();
  uint16_t *A, *B, *C; uint64_t* S = nullptr;
  hipMalloc(&A, na * 2); hipMalloc(&B, nb * 2); hipMalloc(&C, nc * 2);
  hipMemcpy(A, ha.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(B, hb.data(), nb * 2, hipMemcpyHostToDevice);
  hipMemset(C, 0, nc * 2);
  int nwg, tiles_n;
  // K2 operands (N must be 512)
  float *biasd = nullptr, *gammad = nullptr, *betad = nullptr, *meand = nullptr, *rstdd = nullptr;
  void *resd = nullptr, *Xd = nullptr;
  uint16_t* Hd = nullptr;
  std::vector<float> hbias(N), hres;
  if ((variant == 3 || variant == 4)) {
    if (N != 512) { printf("variant %d needs N = 512\n", variant); return 1; }
    for (auto& v : hbias) v = (float)rand() / RAND_MAX - 0.5f;
    std::vector<float> ones(N, 1.f), zeros(N, 0.f);
    hipMalloc(&biasd, N * 4); hipMalloc(&gammad, N * 4); hipMalloc(&betad, N * 4); hipMalloc(&meand, M * 4); hipMalloc(&rstdd, M * 4);
    hipMemcpy(biasd, hbias.data(), N * 4, hipMemcpyHostToDevice);
    hipMemcpy(gammad, ones.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(betad, zeros.data(), N * 4, hipMemcpyHostToDevice);
    hres.resize(nc);
    for (auto& v : hres) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    if (variant == 4) {
      std::vector<uint16_t> r16(nc);
      for (size_t i = 0; i < nc; ++i) { uint32_t u; memcpy(&u, &hres[i], 4); r16[i] = (uint16_t)(u >> 16); u &= 0xffff0000u; memcpy(&hres[i], &u, 4); }
      hipMalloc(&resd, nc * 2); hipMemcpy(resd, r16.data(), nc * 2, hipMemcpyHostToDevice);
      hipMalloc(&Xd, nc * 2);
    } else {
      hipMalloc(&resd, nc * 4); hipMemcpy(resd, hres.data(), nc * 4, hipMemcpyHostToDevice);
      hipMalloc(&Xd, nc * 4);
    }
    hipMalloc(&Hd, nc * 2);
  }
  auto launch = [&](bool stamps) {
    if ((variant == 3 || variant == 4)) {
      nwg = (M + 63) / 64; tiles_n = 1;
      const int smem = k2::NSLOT * k2::STAGE;
      if (variant == 4) {
        hipFuncSetAttribute((const void*)k2::k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        k2::k<true><<<nwg, 512, smem>>>(A, B, biasd, resd, gammad, betad, Xd, Hd, meand, rstdd, M, K, 1e-5f);
      } else {
        hipFuncSetAttribute((const void*)k2::k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        k2::k<false><<<nwg, 512, smem>>>(A, B, biasd, resd, gammad, betad, Xd, Hd, meand, rstdd, M, K, 1e-5f);
      }
      return;
    }
    if (variant == 7 || variant == 8) {
      tiles_n = (N + 127) / 128; nwg = ((M + 127) / 128) * tiles_n;
      if (variant == 7) {
        hipFuncSetAttribute((const void*)k0b::k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * k0::STAGE);
        k0b::k<3><<<nwg, 512, 3 * k0::STAGE>>>(A, B, C, M, N, K, tiles_n, nwg);
      } else {
        hipFuncSetAttribute((const void*)k0b::k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * k0::STAGE);
        k0b::k<4><<<nwg, 512, 4 * k0::STAGE>>>(A, B, C, M, N, K, tiles_n, nwg);
      }
    } else if (variant == 5 || variant == 6) {
      tiles_n = (N + 127) / 128; nwg = ((M + 127) / 128) * tiles_n;
      if (variant == 5) {
        hipFuncSetAttribute((const void*)k0a::k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * k0::STAGE);
        k0a::k<0><<<nwg, 512, 2 * k0::STAGE>>>(A, B, C, M, N, K, tiles_n, nwg);
      } else {
        hipFuncSetAttribute((const void*)k0a::k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * k0::STAGE);
        k0a::k<1><<<nwg, 512, 2 * k0::STAGE>>>(A, B, C, M, N, K, tiles_n, nwg);
      }
    } else if (variant == 0) {
      tiles_n = (N + 127) / 128; nwg = ((M + 127) / 128) * tiles_n;
      hipFuncSetAttribute((const void*)k0::k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * k0::STAGE);
      k0::k<<<nwg, 512, 2 * k0::STAGE>>>(A, B, C, M, N, K, tiles_n, nwg);
    } else {
      tiles_n = (N + 127) / 128; nwg = ((M + 255) / 256) * tiles_n;
      const int smem = k1::NSLOT * k1::STAGE;
#define LAUNCH_K1(PH, ST)                                                                                      \
  do {                                                                                                         \
    hipFuncSetAttribute((const void*)k1::k<PH, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);         \
    k1::k<PH, ST><<<nwg, 512, smem>>>(A, B, C, M, N, K, tiles_n, nwg, S);                                      \
  } while (0)
      if (variant == 1) { if (stamps) LAUNCH_K1(1, true); else LAUNCH_K1(1, false); }
      else { if (stamps) LAUNCH_K1(2, true); else LAUNCH_K1(2, false); }
    }
  };
  launch(false);
  hipDeviceSynchronize();
  hipMemcpy(hc.data(), C, nc * 2, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int s = 0; s < 400; ++s) {
    int m = rand() % M, n = rand() % N;
    if (s < 8) { m = (s & 1) ? M - 1 : 0; n = (s & 2) ? N - 1 : 0; }
    uint32_t u = (uint32_t)hc[(size_t)m * N + n] << 16; float got; memcpy(&got, &u, 4);
    float ref = ref_elem(ha, hb, K, m, n);
    maxerr = std::max(maxerr, (double)fabsf(got - ref) / (fabs(ref) + 1.0));
  }
  if ((variant == 3 || variant == 4)) {
    hipMemcpy(hc.data(), Hd, nc * 2, hipMemcpyDeviceToHost);
    maxerr = 0;
    for (int s2 = 0; s2 < 6; ++s2) {
      const int m = s2 == 0 ? 0 : (s2 == 1 ? M - 1 : rand() % M);
      std::vector<double> xr(N);
      double mu = 0;
      for (int n = 0; n < N; ++n) { xr[n] = (double)ref_elem(ha, hb, K, m, n) + hbias[n] + hres[(size_t)m * N + n]; mu += xr[n]; }
      mu /= N;
      double var = 0;
      for (int n = 0; n < N; ++n) var += (xr[n] - mu) * (xr[n] - mu);
      const double rs = 1.0 / sqrt(var / N + 1e-5);
      for (int n = 0; n < N; ++n) {
        uint32_t u = (uint32_t)hc[(size_t)m * N + n] << 16; float got; memcpy(&got, &u, 4);
        maxerr = std::max(maxerr, fabs(got - (xr[n] - mu) * rs));
      }
    }
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) launch(false);
  hipEventRecord(e0);
  const int iters = 50;
  for (int i = 0; i < iters; ++i) launch(false);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / iters;
  printf("variant %d  M=%d N=%d K=%d  wgs=%d  %.2f us  %.1f TF/s  max rel err %.2e %s\n", variant, M, N, K, nwg, us,
         2.0 * M * N * K / us * 1e-6, maxerr, maxerr < 2e-2 ? "OK" : "WRONG");
  if ((variant == 3 || variant == 4)) return 0;
  if (want_stamps && variant >= 1) {
    hipMalloc(&S, (size_t)nwg * 2 * 16 * 8);
    hipMemset(S, 0, (size_t)nwg * 2 * 16 * 8);
    launch(true); launch(true);
    hipDeviceSynchronize();
    std::vector<uint64_t> hs((size_t)nwg * 2 * 16);
    hipMemcpy(hs.data(), S, hs.size() * 8, hipMemcpyDeviceToHost);
    const int nt = K / 64;
    for (int h = 0; h < 2; ++h) {
      double pro = 0, loop = 0, epi = 0, L = 0, I = 0, V = 0, Bq = 0, Cq = 0;
      for (int w = 0; w < nwg; ++w) {
        const uint64_t* o = &hs[((size_t)w * 2 + h) * 16];
        pro += o[1] - o[0]; loop += o[2] - o[1]; epi += o[3] - o[2];
        L += o[4]; I += o[5]; V += o[6]; Bq += o[7]; Cq += o[8];
      }
      const double d = (double)nwg;
      printf("  half %d: prologue %.0f  loop %.0f (%.0f / K-tile)  epilogue %.0f | per K-tile(first phase): reads-issue %.0f  dma-issue %.0f  "
             "waits %.0f  barrier %.0f  compute+barrier %.0f\n", h, pro / d, loop / d, loop / d / nt, epi / d, L / d / nt, I / d / nt,
             V / d / nt, Bq / d / nt, Cq / d / nt);
    }
  }
  return 0;
}
